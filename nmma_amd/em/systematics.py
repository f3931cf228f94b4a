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
