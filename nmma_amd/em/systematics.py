"""Systematic-uncertainty setup of the EM likelihood: YAML / error budget -> the flat table
the HIP kernel reads.

The reference (``nmma/em/systematics.py``) picks one of several evaluator methods per call
and builds sigma_sys arrays on the host for every sample.  Here sigma_sys is evaluated on
the device, so the host side is only a compiler from the user's inputs to ONE table:

    filter -> Entry(prior names, node times)        # node times None: one sampled scalar

It is built in two passes: the document is first lowered to an ordered list of ``Rule``s
(who, which prior stem, which node grid, whether the rule only fills gaps), then the rules
are applied in order.  The accepted documents and the resulting assignment are the
reference's (:131-160 time ranges, :212-263 document forms, :298-336 the legacy
``config:`` form), which ``tests/test_host_logic.py`` checks against the reference's own
handler.  ``kernel_spec()`` is what ``nmma_amd.engine.EMEngine`` consumes; ``__call__``
keeps the reference's per-sample dict-of-arrays API for plots and diagnostics only.
"""
from __future__ import annotations

from ast import literal_eval
from typing import NamedTuple, Optional

import numpy as np

from .utils import set_filter_associated_dict

#: keys that describe ONE entry; a document whose top level holds them is a single global entry
ENTRY_KEYS = ("time_range", "time_nodes", "prior", "params", "each", "filters")


class Rule(NamedTuple):
    filters: tuple            # filters it addresses; () = every filter no rule has claimed yet
    stem: str                 # prior name (scalar) or prefix of <stem>_<i> (node grid)
    nodes: Optional[np.ndarray]
    fills_gaps: bool = False


class Entry(NamedTuple):
    names: tuple
    nodes: Optional[np.ndarray]


def node_grid(info, span, default_spacing="linear"):
    """Node times of one entry, or None when it has none (reference :131-160).

    ``time_nodes: n`` alone spans the model's time range; ``time_range`` is
    "[spacing] [start] end [n]" with the count taken from its last token when
    ``time_nodes`` is absent."""
    count = info.get("time_nodes")
    tokens = str(info.get("time_range", "")).split()
    if count is None:
        if not tokens:
            return None
        count = tokens.pop()
    spacing, (start, stop) = default_spacing, span
    if len(tokens) == 3:
        spacing, start, stop = tokens
    elif len(tokens) == 2:
        try:
            start, stop = float(tokens[0]), tokens[1]
        except ValueError:
            spacing, stop = tokens
    elif tokens:
        raise ValueError(f"time range specification invalid: {info.get('time_range')!r}")
    maker = np.linspace if "lin" in spacing else np.geomspace if ("log" in spacing or "geo" in spacing) else None
    if maker is None:
        raise ValueError(f"unknown time grid type {spacing!r}")
    return maker(float(start), float(stop), int(count))


def lower_document(doc, filters, span, stem):
    """YAML document -> ordered rules (reference :212-263)."""
    def named(key):
        return f"{stem}_{key}" if key else stem

    if "config" in doc:
        return lower_legacy(doc["config"], filters, span, stem)
    rules = []
    for key, info in doc.items():
        if key in ENTRY_KEYS:
            # the document itself is one entry for every filter; the reference stops reading here
            rules.append(Rule(tuple(filters), named(""), node_grid(doc, span)))
            break
        rules.extend(_lower_entry(key, info, filters, span, named))
    return rules


def _lower_entry(key, info, filters, span, named):
    nodes = node_grid(info, span)
    if key in filters:
        return [Rule((key,), named(key), nodes)]
    if "filters" in info:
        return [Rule(tuple(info["filters"]), named(key), nodes)]
    if "each" in info:
        return [Rule((f,), named(key).replace(key, f), nodes) for f in info["each"]]
    return [Rule((), named(key), nodes, fills_gaps=True)]


def lower_legacy(cfg, filters, span, stem):
    """The pre-0.2 ``config: {withTime, withoutTime}`` document (reference :298-336)."""
    timed = cfg["withTime"]
    flat = cfg.get("withoutTime", {"value": False})
    if bool(timed["value"]) == bool(flat["value"]):
        raise ValueError("Only one of withTime / withoutTime may be true")
    if flat["value"]:
        return [Rule(tuple(filters), stem, None)]
    nodes = np.round(np.linspace(span[0], span[1], timed["time_nodes"]), decimals=2)
    rules = []
    for group in timed["filters"]:
        if group is None:
            return [Rule(tuple(filters), f"{stem}_all", nodes)]
        members = tuple(group) if isinstance(group, list) else (group,)
        rules.append(Rule(members, f"{stem}_" + "___".join(members), nodes))
    return rules


def apply_rules(rules, filters, priors):
    """Rules in order -> {filter: Entry}.  A filter may be addressed once (a second addressed rule
    is a KeyError, as in the reference's bookkeeping, :265-269); a gap-filling rule takes every
    filter no ADDRESSED rule has claimed so far (so two of them stack, and -- as in the
    reference's evaluator -- a node grid then wins over a scalar)."""
    table, unclaimed, gap_filled = {}, list(filters), False
    for rule in rules:
        if rule.nodes is None:
            entry = Entry((rule.stem,), None)
        else:
            entry = Entry(tuple(f"{rule.stem}_{i}" for i in range(len(rule.nodes))), np.asarray(rule.nodes, float))
        for name in entry.names:
            if name not in priors:
                raise AssertionError(f"Required systematics prior missing: {name}")
        if rule.fills_gaps:
            gap_filled = True
            for f in unclaimed:
                if f not in table or entry.nodes is not None or table[f].nodes is None:
                    table[f] = entry
        else:
            for f in rule.filters:
                if f not in unclaimed:
                    raise KeyError(f"filter {f!r} is addressed twice (or not observed) in the systematics document")
                unclaimed.remove(f)
                table[f] = entry
    if unclaimed and not gap_filled:
        raise AssertionError(f"Some filters are missing systematic uncertainty definitions: {set(unclaimed)}")
    return table


class FilterSystematicsHandler:
    """Same constructor, ``reset`` and per-sample call as the reference's handler (:163-210)."""

    def __init__(self, filters, systematics_file=None, error_budget=None,
                 light_curve_times=np.linspace(0.1, 14, 10), base_prior_name="em_syserr"):
        self.filters = list(filters)
        self.light_curve_times = (light_curve_times if isinstance(light_curve_times, dict)
                                  else {f: light_curve_times for f in self.filters})
        self.base_prior_name = base_prior_name
        self.adjust_error_budget(error_budget)
        if isinstance(systematics_file, str):
            import yaml
            with open(systematics_file) as fh:
                systematics_file = yaml.safe_load(fh)
        self.systematics_dict = systematics_file if isinstance(systematics_file, dict) else {}
        self.table = None          # None: fixed error budget (no sampled systematic)

    def adjust_error_budget(self, error_budget):
        budget = 1.0 if error_budget is None else error_budget
        if isinstance(budget, str):
            budget = literal_eval(budget)
        self.error_budget_values = set_filter_associated_dict(budget, self.filters, 1.0)

    def reset(self, model_times, priors):
        """Bind to a model's time span and a prior set (reference :187-192): a systematics
        document wins, else a prior named ``base_prior_name`` means one global sampled scalar,
        else the fixed budget."""
        self.time_range = (model_times[0], model_times[-1])
        if self.systematics_dict:
            rules = lower_document(self.systematics_dict, self.filters, self.time_range, self.base_prior_name)
            self.table = apply_rules(rules, self.filters, priors)
        elif self.base_prior_name in priors:
            self.table = {f: Entry((self.base_prior_name,), None) for f in self.filters}
        else:
            self.table = None

    # the reference's two maps, for callers and tests that look at them
    @property
    def direct_sys_map(self):
        return {f: e.names[0] for f, e in (self.table or {}).items() if e.nodes is None}

    @property
    def interpolate_map(self):
        return {f: (list(e.names), e.nodes) for f, e in (self.table or {}).items() if e.nodes is not None}

    def kernel_spec(self):
        """Flat description for ``EMEngine(systematics=...)``."""
        if self.table is None:
            return {"mode": "budget", "values": dict(self.error_budget_values)}
        direct = self.direct_sys_map
        if len(direct) == len(self.table) and len(set(direct.values())) == 1:
            return {"mode": "param", "name": next(iter(direct.values()))}
        return {"mode": "mixed", "names": direct, "nodes": self.interpolate_map}

    def __call__(self, parameters):
        """sigma_sys per filter at the data epochs for ONE sample (diagnostics only)."""
        t = self.light_curve_times
        if self.table is None:
            return {f: np.full_like(t[f], self.error_budget_values[f]) for f in self.filters}
        out = {}
        for f, entry in self.table.items():
            vals = np.array([parameters[n] for n in entry.names], dtype=float)
            out[f] = (np.full_like(t[f], vals[0]) if entry.nodes is None
                      else np.interp(t[f], entry.nodes, vals, left=vals[0], right=vals[-1]))
        return out


# ---------------------------------------------------------------------------------------------------------------
# The legacy systematics file (``config: {withTime, withoutTime}``) as a source of PRIOR STRINGS: validation and the
# ``name = Prior(...)`` lines the reference writes into a prior file for it (nmma/em/systematics.py:340-513; its tests:
# nmma/tests/systematics.py, restated in tests/test_systematics_yaml.py).  The distribution classes are bilby's analytical
# priors (third-party: bilby.core.prior.analytical); with bilby importable the strings are ``repr`` of its objects, as in the
# reference -- without it the same text is composed from the classes' published constructor signatures below (pinned for
# ``Uniform`` by the string the reference's own test expects; the other rows are restated from bilby 2.x, unpinned here).
# ---------------------------------------------------------------------------------------------------------------
class ValidationError(ValueError):
    """``Validation error for '<key>': <message>`` (reference :9-11)."""

    def __init__(self, key, message):
        super().__init__(f"Validation error for '{key}': {message}")


#: sncosmo bandpass names a legacy systematics file may address (reference :342-369)
ALLOWED_FILTERS = ["2massh", "2massj", "2massks", "atlasc", "atlaso", "bessellb", "besselli", "bessellr", "bessellux", "bessellv",
                   "ps1::g", "ps1::i", "ps1::r", "ps1::y", "ps1::z", "sdssu", "uvot::b", "uvot::u", "uvot::uvm2", "uvot::uvw1",
                   "uvot::uvw2", "uvot::v", "uvot::white", "ztfg", "ztfi", "ztfr"]

_TAIL = (("name", None), ("latex_label", None), ("unit", None), ("boundary", None))
_REQ = object()
#: constructor signatures of bilby's analytical priors, in order: (argument, default) with _REQ for a required one
_SIGNATURES = {
    "DeltaFunction": (("peak", _REQ),) + _TAIL[:3],
    "PowerLaw": (("alpha", _REQ), ("minimum", _REQ), ("maximum", _REQ)) + _TAIL,
    "Uniform": (("minimum", _REQ), ("maximum", _REQ)) + _TAIL,
    "LogUniform": (("minimum", _REQ), ("maximum", _REQ)) + _TAIL,
    "SymmetricLogUniform": (("minimum", _REQ), ("maximum", _REQ)) + _TAIL,
    "Cosine": (("minimum", -np.pi / 2), ("maximum", np.pi / 2)) + _TAIL,
    "Sine": (("minimum", 0), ("maximum", np.pi)) + _TAIL,
    "Gaussian": (("mu", _REQ), ("sigma", _REQ)) + _TAIL,
    "TruncatedGaussian": (("mu", _REQ), ("sigma", _REQ), ("minimum", _REQ), ("maximum", _REQ)) + _TAIL,
    "HalfGaussian": (("sigma", _REQ),) + _TAIL,
    "LogNormal": (("mu", _REQ), ("sigma", _REQ)) + _TAIL,
    "Exponential": (("mu", _REQ),) + _TAIL,
    "StudentT": (("df", _REQ), ("mu", 0.0), ("scale", 1.0)) + _TAIL,
    "Beta": (("alpha", _REQ), ("beta", _REQ), ("minimum", 0), ("maximum", 1)) + _TAIL,
    "Logistic": (("mu", _REQ), ("scale", _REQ)) + _TAIL,
    "Cauchy": (("alpha", _REQ), ("beta", _REQ)) + _TAIL,
    "Gamma": (("k", _REQ), ("theta", 1.0)) + _TAIL,
    "ChiSquared": (("nu", _REQ),) + _TAIL,
    "FermiDirac": (("sigma", _REQ), ("mu", None), ("r", None)) + _TAIL[:3],
    "Categorical": (("ncategories", _REQ),) + _TAIL,
    "Triangular": (("mode", _REQ), ("minimum", _REQ), ("maximum", _REQ)) + _TAIL,
}
for _alias, _of in (("Normal", "Gaussian"), ("TruncatedNormal", "TruncatedGaussian"), ("HalfNormal", "HalfGaussian"),
                    ("LogGaussian", "LogNormal"), ("Lorentzian", "Cauchy")):
    _SIGNATURES[_alias] = _SIGNATURES[_of]


def _spec_class(dist_name, signature):
    """Stand-in for one bilby prior class: keeps the constructor arguments and prints them as bilby's ``Prior.__repr__`` does
    (every constructor argument in signature order; ``latex_label`` falls back to the name)."""

    class _Spec:
        def __init__(self, *args, **kwargs):
            values = dict(zip((k for k, _ in signature), args))
            values.update(kwargs)
            unknown = set(values) - {k for k, _ in signature}
            missing = [k for k, d in signature if d is _REQ and k not in values]
            if unknown or missing:
                raise TypeError(f"{dist_name}: unexpected arguments {sorted(unknown)}, missing {missing}")
            self._values = {k: values.get(k, d) for k, d in signature}
            if self._values.get("latex_label") is None:
                self._values["latex_label"] = self._values.get("name")

        def __repr__(self):
            return dist_name + "(" + ", ".join(f"{k}={v!r}" for k, v in self._values.items()) + ")"

    _Spec.__name__ = _Spec.__qualname__ = dist_name
    return _Spec


def _allowed_distributions():
    try:  # pragma: no cover - bilby is not installed in the build image
        import inspect
        from bilby.core import prior as bprior
        classes = dict(inspect.getmembers(bprior.analytical, inspect.isclass))

        def positional(cls):
            sig = inspect.signature(cls.__init__)
            return [p.name for p in sig.parameters.values() if p.name != "self" and p.default is inspect.Parameter.empty]
        return classes, {k: positional(v) for k, v in classes.items()}
    except Exception:
        classes = {k: _spec_class(k, sig) for k, sig in _SIGNATURES.items()}
        return classes, {k: [a for a, d in sig if d is _REQ] for k, sig in _SIGNATURES.items()}


#: distribution name -> prior class (bilby's when importable), and the constructor arguments each one requires (reference :371-390)
ALLOWED_DISTRIBUTIONS, DISTRIBUTION_PARAMETERS = _allowed_distributions()


def load_yaml(file_path):
    """core/utils.py:46-47: environment variables in the text are expanded before parsing."""
    import os
    from pathlib import Path
    import yaml
    return yaml.safe_load(os.path.expandvars(Path(file_path).read_text()))


def validate_only_one_true(yaml_dict):
    """Exactly one entry of ``config`` is switched on, and every entry says so with a boolean (reference :393-401)."""
    entries = yaml_dict["config"]
    for key, values in entries.items():
        if not isinstance(values.get("value") if isinstance(values, dict) else None, bool):
            raise ValidationError(key, "'value' key must be present and be a boolean")
    switched_on = sum(1 for values in entries.values() if values["value"])
    if switched_on > 1:
        raise ValidationError("config", "Only one configuration key can be set to True at a time")
    if switched_on == 0:
        raise ValidationError("config", "At least one configuration key must be set to True")


def validate_filters(filter_groups):
    """Every filter of a ``withTime`` entry is a known bandpass and is addressed once: not twice inside a group, not in two
    groups; ``None`` stands for "all filters" (reference :404-439)."""
    allowed = ", ".join(str(f) for f in ALLOWED_FILTERS)
    claimed = set()

    def known(filt):
        if filt not in ALLOWED_FILTERS:
            raise ValidationError("filters", f"Invalid filter value '{filt}'. Allowed values are {allowed}")

    def free(filt):
        if filt in claimed:
            raise ValidationError("filters", f"Duplicate filter value '{filt}'. A filter can only be used in one group.")

    for group in filter_groups:
        if isinstance(group, list):
            inside = set()
            for filt in group:
                known(filt)
                if filt in inside:
                    raise ValidationError("filters", f"Duplicate filter value '{filt}' within the same group.")
                free(filt)
                inside.add(filt)
                claimed.add(filt)
        else:
            if group is not None:
                known(group)
            free(group)
            claimed.add(group)


def validate_distribution(distribution):
    """The entry names a known distribution and carries its required constructor arguments (reference :442-456)."""
    dist_type = distribution.get("type")
    if dist_type not in ALLOWED_DISTRIBUTIONS:
        raise ValidationError("distribution type", f"Invalid distribution '{dist_type}'. Allowed values are "
                                                   f"{', '.join(str(f) for f in ALLOWED_DISTRIBUTIONS)}")
    missing = set(DISTRIBUTION_PARAMETERS[dist_type]) - set(distribution.keys())
    if missing:
        raise ValidationError("distribution", f"Missing required parameters for {dist_type} distribution: {', '.join(missing)}")


def create_prior_string(name, distribution):
    """``name = Prior(...)``: the prior-file line of one systematics parameter (reference :459-472).  Only the distribution's
    required arguments are passed on; anything else in the entry (beyond the entry's own keys) is reported and ignored."""
    import warnings
    dist_type = distribution["type"]
    required = DISTRIBUTION_PARAMETERS[dist_type]
    given = {k: v for k, v in distribution.items() if k not in ("type", "value", "time_nodes", "filters")}
    extra = set(given) - set(required)
    if extra:
        warnings.warn(f"Distribution parameters {extra} are not used by {dist_type} distribution and will be ignored")
    args = {k: given[k] for k in required if k in given}
    return f"{name} = {ALLOWED_DISTRIBUTIONS[dist_type](**args, name=name)!r}"


def handle_withTime(values):
    """One prior line per (filter group, time node): ``em_syserr_<group>_<n>``, a group's filters joined by ``___``, ``all`` for
    ``None`` (reference :475-492; the names ``lower_legacy`` above evaluates)."""
    validate_distribution(values)
    groups = values.get("filters", [])
    validate_filters(groups)
    lines = []
    for group in groups:
        label = "___".join(group) if isinstance(group, list) else ("all" if group is None else group)
        lines += [create_prior_string(f"em_syserr_{label}_{n}", dict(values)) for n in range(values["time_nodes"])]
    return lines


def handle_withoutTime(values):
    """One prior line: ``em_syserr`` (reference :495-497)."""
    validate_distribution(values)
    return [create_prior_string("em_syserr", values)]


config_handlers = {"withTime": handle_withTime, "withoutTime": handle_withoutTime}


def get_prior_strings(yaml_dict):
    """The prior lines of the entry that is switched on (reference :504-510)."""
    validate_only_one_true(yaml_dict)
    lines = []
    for key, values in yaml_dict["config"].items():
        if values["value"] and key in config_handlers:
            lines.extend(config_handlers[key](values))
    return lines


def main(yaml_file_path):
    """File -> prior lines (reference :512-513)."""
    return get_prior_strings(load_yaml(yaml_file_path))
