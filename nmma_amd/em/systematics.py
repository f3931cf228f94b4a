"""Systematic-uncertainty handling of the EM likelihood (``nmma/em/systematics.py``).

``FilterSystematicsHandler`` keeps the reference's selection logic (:212-263, legacy
YAML :298-336) so the same YAML / error-budget inputs pick the same evaluator, and adds
``kernel_spec()``: the flat description the HIP kernel consumes.  ``__call__`` keeps the
reference's per-sample API (a dict of sigma arrays) for diagnostics and plotting; the
likelihood itself never calls it -- sigma_sys is evaluated on the device.
"""
from __future__ import annotations

from ast import literal_eval

import numpy as np

from .utils import set_filter_associated_dict


def _interp_constant(x, nodes, vals):
    return np.interp(x, nodes, vals, left=vals[0], right=vals[-1])


class FilterSystematicsHandler:
    allowed_keys = ["time_range", "time_nodes", "prior", "params", "each", "filters"]

    def __init__(self, filters, systematics_file=None, error_budget=None,
                 light_curve_times=np.linspace(0.1, 14, 10), base_prior_name="em_syserr"):
        self.filters = list(filters)
        if not isinstance(light_curve_times, dict):
            light_curve_times = {f: light_curve_times for f in self.filters}
        self.base_prior_name = base_prior_name
        self.default_t_grid_type = "linear"
        self.light_curve_times = light_curve_times
        self.adjust_error_budget(error_budget)
        self.mode = "budget"
        self.direct_sys_map, self.interpolate_map = {}, {}
        if isinstance(systematics_file, str):
            import yaml
            with open(systematics_file) as fh:
                self.systematics_dict = yaml.safe_load(fh) or {}
        elif isinstance(systematics_file, dict):
            self.systematics_dict = systematics_file
        else:
            self.systematics_dict = {}

    # systematics.py:203-210
    def adjust_error_budget(self, error_budget):
        if error_budget is None:
            error_budget = 1.0
        elif isinstance(error_budget, str):
            error_budget = literal_eval(error_budget)
        self.error_budget_values = set_filter_associated_dict(error_budget, self.filters, 1.0)

    def prior_name(self, key):
        return f"{self.base_prior_name}_{key}" if key else self.base_prior_name

    # systematics.py:131-160
    def get_time_range(self, info):
        num = info.get("time_nodes", None)
        t_range = info.get("time_range", "").split()
        if num is None and t_range:
            num = t_range.pop(-1)
        if num is None:
            return None
        if len(t_range) == 3:
            grid_type, t_start, t_end = t_range
        elif len(t_range) == 2:
            t_start, t_end = t_range
            grid_type = self.default_t_grid_type
            try:
                float(t_start)
            except ValueError:
                grid_type, t_end = t_range
                t_start = self.time_range[0]
        elif len(t_range) == 0:
            t_start, t_end = self.time_range
            grid_type = self.default_t_grid_type
        else:
            raise ValueError("time range specfication invalid")
        if "lin" in grid_type:
            return np.linspace(float(t_start), float(t_end), int(num))
        if "log" in grid_type or "geo" in grid_type:
            return np.geomspace(float(t_start), float(t_end), int(num))
        raise ValueError(f"unknown time grid type {grid_type}")

    def get_name_and_times(self, key, info):
        return self.prior_name(key), self.get_time_range(info)

    # systematics.py:187-192
    def reset(self, model_times, priors):
        self.time_range = (model_times[0], model_times[-1])
        if self.systematics_dict:
            self.setup_systematics_sampling(priors)
        elif self.base_prior_name in priors:
            self.mode = "param"

    # systematics.py:212-263
    def setup_systematics_sampling(self, priors):
        self.direct_sys_map, self.interpolate_map = {}, {}
        self.missing_filters = set(self.filters)
        cleared = False
        for key, info in self.systematics_dict.items():
            if key == "config":
                self.legacy_systematics_setup(self.systematics_dict)
                break
            if key in self.allowed_keys:
                name, tr = self.get_name_and_times("", self.systematics_dict)
                for f in self.filters:
                    self._register(f, tr, name, priors)
                break
            if key in self.filters:
                name, tr = self.get_name_and_times(key, info)
                self._register(key, tr, name, priors)
            elif "filters" in info:
                name, tr = self.get_name_and_times(key, info)
                for f in info["filters"]:
                    self._register(f, tr, name, priors)
            elif "each" in info:
                name, tr = self.get_name_and_times(key, info)
                for f in info["each"]:
                    self._register(f, tr, name.replace(key, f), priors)
            else:
                cleared = True
                name, tr = self.get_name_and_times(key, info)
                for f in sorted(self.missing_filters, key=self.filters.index):
                    self._register(f, tr, name, priors, clean=False)
        assert cleared or len(self.missing_filters) == 0, \
            f"Some filters are missing systematic uncertainty definitions: {self.missing_filters}"
        if not self.interpolate_map:
            self.mode = "param" if len(set(self.direct_sys_map.values())) == 1 else "single"
            if self.mode == "param":
                self.base_prior_name = next(iter(self.direct_sys_map.values()))
        elif not self.direct_sys_map:
            self.mode = "interp"
        else:
            self.mode = "mixed"

    def _register(self, filt, time_range, prior_name, priors, clean=True):
        if clean:
            self.direct_sys_map.pop(filt, None)
            self.interpolate_map.pop(filt, None)
            self.missing_filters.discard(filt)
        if time_range is None:
            assert prior_name in priors, "Required systematics prior missing"
            self.direct_sys_map[filt] = prior_name
        else:
            names = [f"{prior_name}_{i}" for i, _ in enumerate(time_range)]
            for p in names:
                assert p in priors, f"Required systematics prior missing: {p}"
            self.interpolate_map[filt] = (names, np.asarray(time_range, float))

    # systematics.py:298-336
    def legacy_systematics_setup(self, sysdict):
        cfg = sysdict["config"]
        with_time, without = cfg["withTime"], cfg.get("withoutTime", {"value": False})
        if bool(with_time["value"]) == bool(without["value"]):
            raise ValueError("Only one of withTime / withoutTime may be true")
        if not with_time["value"]:
            self.direct_sys_map = {f: self.base_prior_name for f in self.filters}
            self.missing_filters = set()
            return
        groups = {}
        for grp in list(with_time["filters"]):
            if grp is None:
                groups = {f: "all" for f in self.filters}
                self.missing_filters = set()
                break
            if isinstance(grp, list):
                for f in grp:
                    self.missing_filters.discard(f)
                    groups[f] = "___".join(grp)
            else:
                groups[grp] = grp
                self.missing_filters.discard(grp)
        nodes = np.round(np.linspace(*self.time_range, with_time["time_nodes"]), decimals=2)
        self.interpolate_map = {f: ([f"{self.base_prior_name}_{name}_{i}" for i, _ in enumerate(nodes)], nodes)
                                for f, name in groups.items()}

    # ---- what the HIP kernel consumes (engine.EMEngine._systematics_arrays)
    def kernel_spec(self):
        if self.mode == "budget":
            return {"mode": "budget", "values": dict(self.error_budget_values)}
        if self.mode == "param":
            return {"mode": "param", "name": self.base_prior_name}
        return {"mode": "mixed", "names": dict(self.direct_sys_map),
                "nodes": {f: (list(n), np.asarray(t, float)) for f, (n, t) in self.interpolate_map.items()}}

    # ---- reference per-sample API (diagnostics only; systematics.py:51-55, :279-296)
    def __call__(self, parameters):
        t = self.light_curve_times
        if self.mode == "budget":
            return {f: np.full_like(t[f], self.error_budget_values[f]) for f in self.filters}
        if self.mode == "param":
            return {f: np.full_like(t[f], parameters[self.base_prior_name]) for f in self.filters}
        out = {f: np.full_like(t[f], parameters[n]) for f, n in self.direct_sys_map.items()}
        for f, (names, nodes) in self.interpolate_map.items():
            out[f] = _interp_constant(t[f], nodes, np.array([parameters[p] for p in names]))
        return out
