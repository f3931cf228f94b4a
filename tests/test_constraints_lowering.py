"""Constraint priors lowered to a device program (nmma_amd/core/constraints.py): the tracer runs the likelihood's OWN conversion
functions on symbolic columns; here the resulting program -- through its numpy interpreter ``evaluate_program``, the arithmetic of
the kernel ``con_row_ok`` -- is compared with ``evaluate_constraints(parameter_conversion(columns))`` evaluated by numpy on real
columns, i.e. with what the reference does per sample (nmma/core/base.py:51-82).  CPU only; the kernel itself is compared with the
interpreter in tests/test_gpu_constraints.py."""
import numpy as np
import pytest

from nmma_amd import _lib as L
from nmma_amd.core import conversion as cv
from nmma_amd.core.base import Constraint, ConstraintSet
from nmma_amd.core.constraints import Sym, evaluate_program, trace_constraints


def _host_mask(constraints, names, fixed, conversions, theta):
    sample = {n: theta[:, i] for i, n in enumerate(names)}
    for k, v in fixed.items():
        sample.setdefault(k, np.full(len(theta), v))
    for conv in conversions:
        out = conv(sample)
        sample = out[0] if isinstance(out, tuple) else out
    with np.errstate(all="ignore"):
        return np.asarray(ConstraintSet(constraints).mask(sample), dtype=bool)


def _check(constraints, names, fixed, conversions, theta, expect_some=True):
    prog = trace_constraints(constraints, names, fixed, conversions)
    assert prog is not None and len(prog) >= 2 * len(constraints)
    got = evaluate_program(prog, theta)
    want = _host_mask(constraints, names, fixed, conversions, theta)
    assert np.array_equal(got, want)
    if expect_some:
        assert 0 < want.sum() < len(want)
    return prog


def test_em_conversion_chain_knTheta_and_log10_alias():
    """KNtheta from inclination_EM (core/conversion.py:119-126), a log10_ alias of a linear mass (em/model.py:272-286), a sampled
    column constrained directly and a FIXED parameter's constraint (a constant program)."""
    from nmma_amd.em.model import LightCurveModelContainer
    m = object.__new__(LightCurveModelContainer)
    m.model_parameters = ["log10_mej_dyn", "log10_mej_wind", "KNphi", "KNtheta"]
    names = ["luminosity_distance", "KNphi", "inclination_EM", "timeshift", "mej_dyn", "log10_mej_wind"]
    rng = np.random.default_rng(1)
    theta = np.column_stack([rng.uniform(1, 200, 500), rng.uniform(15, 75, 500), np.arccos(rng.uniform(0, 1, 500)), rng.uniform(-2, 0.1, 500),
                             10 ** rng.uniform(-3, -1, 500), rng.uniform(-3, -0.5, 500)])
    cons = {"KNtheta": Constraint(10.0, 60.0, "KNtheta"), "log10_mej_dyn": Constraint(-2.8, -1.2, "log10_mej_dyn"),
            "timeshift": Constraint(-1.5, 0.0, "timeshift"), "Ebv": Constraint(-1.0, 1.0, "Ebv")}
    prog = _check(cons, names, {"Ebv": 0.0}, [m.parameter_conversion], theta)
    assert any(op == L.CON_LOG10 for op, _, _ in prog) and any(op == L.CON_PUSH_COL and col == 2 for op, col, _ in prog)
    # NaN in a constrained column fails the check, like Constraint.prob
    theta[3, 3] = np.nan
    assert not evaluate_program(prog, theta)[3]


def test_theta_jn_folding_and_cos_theta_jn():
    from nmma_amd.em.model import LightCurveModelContainer
    m = object.__new__(LightCurveModelContainer)
    m.model_parameters = ["KNtheta"]
    rng = np.random.default_rng(2)
    for name, col in (("theta_jn", np.arccos(rng.uniform(-1, 1, 400))), ("cos_theta_jn", rng.uniform(-1, 1, 400))):
        _check({"KNtheta": Constraint(20.0, 70.0)}, [name], {}, [m.parameter_conversion], col[:, None])


def test_gw_component_masses_from_chirp_mass_and_mass_ratio():
    """priors/GWBNS.prior: mass_1 / mass_2 are Constraint priors on quantities bilby derives from (chirp_mass, mass_ratio);
    lambda_1 / lambda_2 constrained where they are sampled.  The redshift the source-frame conversion root-finds is opaque and
    does not stand in the way -- unless it is what is constrained."""
    names = ["chirp_mass", "mass_ratio", "lambda_1", "lambda_2", "luminosity_distance", "chi_1"]
    rng = np.random.default_rng(3)
    theta = np.column_stack([rng.uniform(1.1, 1.3, 600), rng.uniform(0.3, 1.0, 600), rng.uniform(-100, 3000, 600), rng.uniform(-100, 3000, 600),
                             rng.uniform(10, 100, 600), rng.uniform(-0.05, 0.05, 600)])
    cons = {"mass_1": Constraint(1.001398, 1.9, "mass_1"), "mass_2": Constraint(1.1, 4.31, "mass_2"),
            "lambda_1": Constraint(0.0, np.inf, "lambda_1"), "lambda_2": Constraint(0.0, np.inf, "lambda_2")}
    prog = trace_constraints(cons, names, {}, [cv.bns_source_frame])
    assert prog is not None
    got = evaluate_program(prog, theta)
    q, mc = theta[:, 1], theta[:, 0]
    m1 = mc * (1 + q) ** 1.2 / q ** 0.6 / (1 + q)
    want = (m1 > 1.001398) & (m1 < 1.9) & (m1 * q > 1.1) & (m1 * q < 4.31) & (theta[:, 2] > 0) & (theta[:, 3] > 0)
    assert np.array_equal(got, want) and 0 < want.sum() < len(want)
    assert trace_constraints({"mass_1_source": Constraint(1.0, 2.0)}, names, {}, [cv.bns_source_frame]) is None      # needs z(d_L)
    # component masses sampled directly, mass ratio constrained (priors/injec_alspin.prior)
    names2 = ["mass_1", "mass_2"]
    th2 = np.column_stack([rng.uniform(1.0, 3.0, 300), rng.uniform(0.2, 2.0, 300)])
    _check({"mass_ratio": Constraint(0.125, 1.0)}, names2, {}, [cv.generate_mass_parameters], th2)


def test_supernova_grid_derived_fractions():
    """priors/AnBa2022.prior: mni_c, mrp_c (core/conversion.py:184-192)."""
    names = ["log10_mtot", "log10_mni", "log10_mrp", "xmix"]
    rng = np.random.default_rng(4)
    theta = np.column_stack([rng.uniform(0.3, 1.0, 500), rng.uniform(-2.0, 0.2, 500), rng.uniform(-2.5, -0.5, 500), rng.uniform(0.1, 0.9, 500)])
    _check({"mni_c": Constraint(0, 0.5), "mrp_c": Constraint(0, 2.0)}, names, {}, [cv.convert_mtot_mni], theta)


def test_untraceable_sets_are_refused_not_approximated():
    names = ["a", "b"]
    def branchy(p):
        p["c"] = p["a"] if p["a"] > 0 else p["b"]
        return p
    def table_lookup(p):
        p["c"] = np.interp(p["a"], [0.0, 1.0], [0.0, 2.0])
        return p
    def exotic_ufunc(p):
        p["c"] = np.arctan2(p["a"], p["b"])
        return p
    for conv in (branchy, table_lookup, exotic_ufunc):
        assert trace_constraints({"c": Constraint(0, 1)}, names, {}, [conv]) is None
        # ... but a constraint that does not depend on the untraceable quantity still lowers when the conversion survives the trace
    assert trace_constraints({"a": Constraint(0, 1)}, names, {}, [exotic_ufunc]) is not None
    assert trace_constraints({"missing": Constraint(0, 1)}, names, {}, []) is None
    assert trace_constraints({}, names, {}, []) == []
    deep = Sym.column(0)
    for _ in range(12):
        deep = 1.0 / (1.0 + deep)              # right-nested: two stack slots per level
    assert trace_constraints({"d": Constraint(0, 1)}, names, {}, [lambda p: dict(p, d=deep)]) is None      # deeper than NMMA_CON_MAX_STACK
    big = Sym.column(0)
    for _ in range(40):
        big = big * big + 1.0                  # 2^40 operations written out
    assert trace_constraints({"d": Constraint(0, 1)}, names, {}, [lambda p: dict(p, d=big)]) is None


def test_joint_likelihood_chain_is_traced_batched():
    """MultiMessengerLikelihood: the constraint program is traced through the messengers' conversions in turn (or the
    MultimessengerConversion chain, without its scalar unwrapping)."""
    from nmma_amd.core.conversion import MultimessengerConversion
    chain = MultimessengerConversion.from_dict({"gw": cv.bns_source_frame, "custom": lambda p: dict(p, msum=p["mass_1"] + p["mass_2"])})
    conv = lambda p: chain.convert_to_multimessenger_parameters(p, batched=True)
    names = ["chirp_mass", "mass_ratio"]
    rng = np.random.default_rng(5)
    theta = np.column_stack([rng.uniform(1.0, 1.5, 300), rng.uniform(0.3, 1.0, 300)])
    prog = trace_constraints({"msum": Constraint(2.5, 3.2)}, names, {}, [conv])
    assert prog is not None
    q, mc = theta[:, 1], theta[:, 0]
    tot = mc * (1 + q) ** 1.2 / q ** 0.6
    assert np.array_equal(evaluate_program(prog, theta), (tot > 2.5) & (tot < 3.2))
    # a batch of ONE row keeps its columns (round-3 advisor finding on _scalar)
    out = chain.convert_to_multimessenger_parameters({"chirp_mass": np.array([1.2]), "mass_ratio": np.array([0.8])}, batched=True)
    assert np.shape(out["msum"]) == (1,)
    out = chain.convert_to_multimessenger_parameters({"chirp_mass": np.array([1.2]), "mass_ratio": np.array([0.8])})
    assert np.ndim(out["msum"]) == 0


def test_device_prior_table_takes_exact_class_names_only():
    """Round-3 advisor finding: TruncatedNormal / LogNormal / SymmetricLogUniform must not inherit the Gaussian's or the
    log-uniform's transform through a suffix match."""
    from math import erf, sqrt
    from nmma_amd.sampler import device_prior_table

    def cls(name, **kw):
        return type(name, (), {})().__class__, kw

    def make(name, **kw):
        obj = type(name, (), {})()
        for k, v in kw.items():
            setattr(obj, k, v)
        return obj

    t = device_prior_table({"x": make("TruncatedNormal", mu=0.9, sigma=0.3, minimum=0.0, maximum=10.0)}, ["x"])     # priors/Sr2023.prior
    assert t is not None and t[0].kind == L.PRIOR_TRUNC_GAUSSIAN and (t[0].a, t[0].b) == (0.9, 0.3)
    e_lo, e_hi = erf((0.0 - 0.9) / (sqrt(2) * 0.3)), erf((10.0 - 0.9) / (sqrt(2) * 0.3))
    assert t[0].c == pytest.approx(e_lo, rel=1e-15) and t[0].alpha == pytest.approx((e_hi - e_lo) / 2, rel=1e-15)
    assert device_prior_table({"x": make("TruncatedGaussianPrior", mu=0.0, sigma=1.0, minimum=-1.0, maximum=1.0)}, ["x"])[0].kind == L.PRIOR_TRUNC_GAUSSIAN
    assert device_prior_table({"x": make("LogNormal", mu=0.1, sigma=0.5)}, ["x"])[0].kind == L.PRIOR_LOGNORMAL
    assert device_prior_table({"x": make("LogGaussian", mu=0.1, sigma=0.5)}, ["x"])[0].kind == L.PRIOR_LOGNORMAL
    assert device_prior_table({"x": make("HalfNormal", sigma=0.5)}, ["x"])[0].kind == L.PRIOR_HALF_GAUSSIAN
    assert device_prior_table({"x": make("Gaussian", mu=0.1, sigma=0.5)}, ["x"])[0].kind == L.PRIOR_GAUSSIAN
    assert device_prior_table({"x": make("UniformPrior", minimum=0.0, maximum=2.0)}, ["x"])[0].kind == L.PRIOR_UNIFORM
    # no device formula: the host transform has to be used
    for name, kw in (("SymmetricLogUniform", dict(minimum=0.1, maximum=2.0)), ("MyUniform", dict(minimum=0.0, maximum=1.0)),
                     ("Gaussian", dict(mu=0.0, sigma=1.0, minimum=-1.0, maximum=1.0)), ("Beta", dict(alpha=1.0, beta=2.0, minimum=0, maximum=1)),
                     ("Interped", dict(minimum=0.0, maximum=1.0)), ("ConditionalUniform", dict(minimum=0.0, maximum=1.0)),
                     ("TruncatedNormal", dict(mu=0.0, sigma=1.0))):
        assert device_prior_table({"x": make(name, **kw)}, ["x"]) is None, name
