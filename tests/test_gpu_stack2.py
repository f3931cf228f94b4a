"""The combined model of two transients in ONE launch (``nmma_em_loglike_stack2``: ``em_logl<.., 7>`` takes the second transient's
curves as an operand and forms the flux sum on the two nodes every datum interpolates between) -- against the golden log L of
the reference's own CombinedLightCurveModelContainer (nmma/em/model.py:1411-1510), against the materialising path
(``em_fused<MODE_LC_ABS>`` -> ``em_lc_loglike``), with gaps / edges / failed rows in the second transient's curves, and the
conditions under which a handle has no one-launch form."""
import numpy as np
import pytest

from nmma_amd import synthetic as syn
from tests import cases, cases_combined
from tests.helpers import SimplePrior, rel_err

pytestmark = pytest.mark.gpu
FLOOR = -1.7976931348623157e308
# One-launch form vs materialising path.  NOT bit-identical by construction: the materialising path reconstructs all nodes on the
# fp64 matrix cores and divides by the SVD-grid spacing (em_fused); the lean task reconstructs the bracket rows as FMA chains and
# multiplies by the reciprocal spacing (DEVIATIONS.md).  Node magnitudes agree to a few ulp; log L to ~1e-13 relative.
FUSED_VS_MATERIALISED_RTOL = 1e-10


def _engines(case, stack_operands=1, sample_times=None):
    from nmma_amd.engine import EMEngine
    st = case["sample_times"] if sample_times is None else sample_times
    one = EMEngine(case["svd"], case["filters"], case["model_parameters"], case["names"], sample_times=st,
                   cosmo_grid=case["cosmo_grid"], data=case["data"], observed_filters=case["filters"], stack_operands=stack_operands)
    kn = EMEngine(case["svd"], case["filters"], case["model_parameters"], case["names"], sample_times=st, cosmo_grid=case["cosmo_grid"])
    tail = EMEngine(None, case["filters"], [], case["names"], sample_times=st, cosmo_grid=case["cosmo_grid"],
                    data=case["data"], observed_filters=case["filters"], model_kind="external")
    return one, kn, tail


def _theta(seed, B):
    rng = np.random.default_rng(seed)
    return np.concatenate([syn.draw_theta(seed + 1, B, cases_combined.NAMES[:6])[1], rng.uniform(-17.5, -14.0, (B, 1)),
                           rng.uniform(0.8, 1.6, (B, 1))], axis=1)


def _plugin(case, generate_lightcurve=None):
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel, SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    kn = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["filters"], model_parameters=case["model_parameters"],
                            sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    grb = ExternalLightCurveModel("PLGRB", case["filters"], case["sample_times"], model_parameters=["grb_mag0", "grb_slope"],
                                  generate_lightcurve=generate_lightcurve)
    comb = CombinedLightCurveModelContainer([kn, grb], cosmo_grid=case["cosmo_grid"])
    times, mags, sigmas = case["data"]
    sys_ref = case.get("systematics_ref") or dict(error_budget=1.0, systematics_file=None)
    handler = FilterSystematicsHandler(case["filters"], systematics_file=sys_ref["systematics_file"], error_budget=sys_ref["error_budget"],
                                       light_curve_times=times)
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    return EMTransientLikelihood(comb, (times, mags, sigmas, 0.0), handler, priors, filters=case["filters"])


def test_golden_combined_goes_through_the_one_launch_form():
    """The reference's combined model (golden ``combined``: its own container on the imported source) through the plugin: the
    likelihood picks the one-launch engine (two sub-models on one grid), never builds the likelihood-from-curves engine, and
    matches the golden log L at 1e-6; the materialising path on the same inputs agrees with it to 1e-10."""
    import torch
    case = cases_combined.case_combined()
    gold = cases.load_golden("combined")
    _, grb_oracle = cases_combined.oracle_likelihood(case)
    lik = _plugin(case)
    st = case["sample_times"]
    ext = np.stack([np.stack([grb_oracle.abs_lightcurves(dict(zip(case["names"], row)), st)[f] for f in case["filters"]])
                    for row in case["theta"]])
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    sub = lik.sub_model
    assert sub._engine2 is not None and sub._engine is None and not sub._stack2_off        # the one-launch engine, and only it
    err = rel_err(got, gold["logl"])
    print(f"combined through em_logl<.., 7>: max rel err vs the reference {err.max():.3e}")
    assert np.array_equal(got == FLOOR, gold["logl"] == FLOOR) and err.max() <= 1e-6
    sub._stack2_off = True                                                                 # force the materialising path
    mat = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    assert sub._engine is not None
    assert rel_err(got, mat).max() <= FUSED_VS_MATERIALISED_RTOL
    # a sub-model without a light curve for some rows (model.py:1423-1426): floor, the other rows untouched
    sub._stack2_off = False
    ok = np.ones(len(ext), dtype=bool)
    ok[[3, 17]] = False
    got2 = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": (torch.as_tensor(ext), ok)})
    assert np.all(got2[~ok] == FLOOR) and np.array_equal(got2[ok], got[ok])


def test_combined_model_is_a_drop_in_with_a_host_generator():
    """Config 3 as a user of the reference runs it: the second transient is a HOST model with the reference's signature
    ``generate_lightcurve(sample_times, parameters) -> {filter: mag}`` (their GRBLightCurveModel; here the power-law stand-in the
    goldens were made with).  No ``external_lc`` anywhere: ``log_likelihood(parameters)``, ``log_likelihood_batch(theta)`` and
    ``GPUPool.map`` call the generator once per row and evaluate each batch in one launch -- golden log L at 1e-6; a row for which
    the generator returns an empty dict is floored (model.py:1423-1426)."""
    from nmma_amd.pool import GPUPool
    case = cases_combined.case_combined_syserr()
    gold = cases.load_golden("combined_syserr")
    _, grb_oracle = cases_combined.oracle_likelihood(case)
    calls = []

    def generate(sample_times, parameters):
        calls.append(1)
        if parameters["grb_mag0"] < -17.6:            # (outside the case's prior box: see the failure check below)
            return {}
        return grb_oracle.abs_lightcurves(parameters, np.asarray(sample_times))

    lik = _plugin(case, generate)
    names, theta = case["names"], case["theta"]
    got = lik.log_likelihood_batch(theta, names)
    assert len(calls) == len(theta) and lik.sub_model._engine2 is not None and lik.sub_model._engine is None
    assert rel_err(got, gold["logl"]).max() <= 1e-6
    one = lik.log_likelihood(dict(zip(names, (float(v) for v in theta[5]))))
    assert abs(one - gold["logl"][5]) <= 1e-6 * abs(gold["logl"][5])
    pool = GPUPool(lik, queue_size=16, names=names)
    mapped = np.array(pool.map(pool.log_likelihood, [row for row in theta[:16]]))
    assert np.array_equal(mapped, got[:16]) and pool.n_batches == 1
    failing = theta[:4].copy()
    failing[2, names.index("grb_mag0")] = -17.7
    out = lik.log_likelihood_batch(failing, names)
    assert out[2] == FLOOR and np.array_equal(out[[0, 1, 3]], got[[0, 1, 3]])
    assert lik.log_likelihood(dict(zip(names, (float(v) for v in failing[2])))) == FLOOR          # (core/base.py:77-82)


@pytest.mark.parametrize("name", ["combined_syserr", "combined_loggrid"])
def test_golden_combinations_with_extras_and_holes(name):
    """Reference-made goldens of the shared-grid combination with a SAMPLED systematic (the task's sigma_tot per datum) and on the CLI's
    log-spaced grid (em_logl<.., 8>: bracket search), the second transient's curves with an interior hole in 18 of 40 rows (the
    reference fills it: autocomplete_data, model.py:1440-1448 -- here those rows take the re-evaluation launch): through the plugin,
    which picks the one-launch engine, at 1e-6 against the reference; the materialising path agrees to 1e-10."""
    import torch
    case = getattr(cases_combined, "case_" + name)()
    gold = cases.load_golden(name)
    _, grb_oracle = cases_combined.oracle_likelihood(case)
    lik = _plugin(case)
    st = case["sample_times"]
    ext = np.stack([np.stack([grb_oracle.abs_lightcurves(dict(zip(case["names"], row)), st)[f] for f in case["filters"]])
                    for row in case["theta"]])
    holes = np.isnan(ext).any(axis=(1, 2))
    assert 10 <= holes.sum() <= 30
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    sub = lik.sub_model
    assert sub._engine2 is not None and sub._engine is None, getattr(sub._engine2, "stack2_reason", None)
    err = rel_err(got, gold["logl"])
    print(f"{name} through the one-launch form: max rel err vs the reference {err.max():.3e} (rows with holes: {err[holes].max():.3e})")
    assert np.array_equal(got == FLOOR, gold["logl"] == FLOOR) and err.max() <= 1e-6
    sub._stack2_off = True
    mat = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    assert rel_err(got, mat).max() <= FUSED_VS_MATERIALISED_RTOL
    # re-evaluated rows carry the materialising kernels' own bits (a row is re-evaluated when one of its data brackets a hole node:
    # always for combined_syserr's hole at 4.6-6.1 days; on the log grid the hole spans 0.8-1.1 days, which few epochs reach)
    same = got[holes] == mat[holes]
    print(f"   {int(same.sum())} of {int(holes.sum())} rows with a hole were re-evaluated")
    if name == "combined_syserr":
        assert same.all()
    sub._stack2_off = False


@pytest.mark.parametrize("B", [1, 5, 16, 17, 333, 4100, 20011])      # (20011: more tiles than workgroups in the re-evaluation launch)
def test_one_launch_against_materialised_with_gaps_and_edges(B):
    """Second-transient curves with everything the operand contract allows: finite, +inf at the first / last nodes (the edge of its
    time range: handled in the task), +inf over the first three nodes, NaN / inf holes in the middle (gaps autocomplete_data fills),
    a curve without any finite node.  Rows that met an interior non-finite node are re-evaluated by the materialising kernels in
    the same call: the SAME bits as the materialising path; every other row agrees to 1e-10."""
    import torch
    case = cases_combined.case_combined()
    one, kn, tail = _engines(case)
    M, NS = len(case["filters"]), len(case["sample_times"])
    rng = np.random.default_rng(700 + B)
    theta = _theta(70 + B, B)
    th = torch.as_tensor(theta, device="cuda:0")
    lc2 = rng.uniform(-17.0, -12.0, (B, M, NS))
    kind = rng.integers(0, 6, B)                       # 0, 1: all finite; 2: first node inf; 3: first three; 4: holes; 5: last node + a dark curve
    lc2[kind == 2, :, 0] = np.inf
    lc2[kind == 3, :, :3] = np.inf
    holes = (rng.uniform(size=lc2.shape) < 0.04) & (kind == 4)[:, None, None]
    lc2[holes] = rng.choice([np.inf, np.nan, -np.inf], size=int(holes.sum()))
    lc2[kind == 5, :, NS - 1] = np.inf
    lc2[kind == 5, 1, :] = np.inf
    lc2_t = torch.as_tensor(lc2, device="cuda:0")
    got = one.loglike_stack2(th, lc2_t)
    assert got is not None
    one.check()
    want = tail.loglike_lc_sets(th, [kn.model_lightcurves(th), lc2_t])
    got, want = got.cpu().numpy(), want.cpu().numpy()
    assert np.array_equal(got == FLOOR, want == FLOOR)
    fin = want > FLOOR
    assert rel_err(got[fin], want[fin]).max() <= FUSED_VS_MATERIALISED_RTOL if fin.any() else True
    # the rows the kernel flagged carry the materialising kernels' own values (which rows those are follows from the data: a
    # non-finite node strictly inside the grid that a datum's bracket touches -- rows of kind 3 and 4 are the candidates)
    cand = np.isin(kind, (3, 4))
    exact = got == want
    print(f"B={B}: {int((~exact).sum())} rows differ in the last bits, {int(cand.sum())} candidates for re-evaluation, "
          f"max rel {rel_err(got[fin], want[fin]).max() if fin.any() else 0:.2e}")
    if B >= 333:
        assert exact[kind == 4].mean() > 0.5           # most rows with holes met one and were re-evaluated: bit-identical
    # a sub-model failure flag floors the row whatever its curves hold
    bad = torch.zeros(B, dtype=torch.bool, device="cuda:0")
    bad[B // 2] = True
    flagged = one.loglike_stack2(th, lc2_t, bad).cpu().numpy()
    assert flagged[B // 2] == FLOOR
    keep = np.arange(B) != B // 2
    assert np.array_equal(flagged[keep], got[keep])
    for e in (one, kn, tail):
        e.close()


def test_one_launch_properties_and_batch_independence():
    """Determinism, permutation equivariance and independence of the batch size / tile geometry (16-sample tiles up to 4096 rows,
    32-sample tiles beyond) -- also for rows that are re-evaluated."""
    import torch
    case = cases_combined.case_combined()
    one, kn, tail = _engines(case)
    M, NS = len(case["filters"]), len(case["sample_times"])
    B = 5000
    rng = np.random.default_rng(9)
    theta = _theta(91, B)
    lc2 = rng.uniform(-17.0, -12.0, (B, M, NS))
    lc2[::7, :, 0] = np.inf
    lc2[3::11, 2, 10] = np.nan

    def fn(rows):
        return one.loglike_stack2(torch.as_tensor(theta[rows], device="cuda:0"), torch.as_tensor(lc2[rows], device="cuda:0")).cpu().numpy()
    rows = np.arange(B)
    a = fn(rows)
    assert np.array_equal(a, fn(rows))
    perm = rng.permutation(B)
    assert np.array_equal(fn(perm), a[perm])
    for n in (1, 100, 777, 4096):
        assert np.array_equal(fn(rows[:n]), a[:n]), n
    for e in (one, kn, tail):
        e.close()


def test_gap_free_promise_skips_the_re_evaluation_and_fails_loudly_when_broken():
    """``gap_free=True`` (NMMA_STACK2_GAP_FREE): same values for curves that keep the promise (finite, or non-finite at the first /
    last node only); a curve with an interior gap poisons the engine -- the next call raises and names the cause."""
    import torch
    from nmma_amd import _lib as L
    case = cases_combined.case_combined()
    one, kn, tail = _engines(case)
    M, NS = len(case["filters"]), len(case["sample_times"])
    rng = np.random.default_rng(3)
    B = 700
    th = torch.as_tensor(_theta(31, B), device="cuda:0")
    lc2 = rng.uniform(-17.0, -12.0, (B, M, NS))
    lc2[::3, :, 0] = np.inf
    lc2[1::5, 4, NS - 1] = np.nan
    lc2_t = torch.as_tensor(lc2, device="cuda:0")
    safe = one.loglike_stack2(th, lc2_t)
    fast = one.loglike_stack2(th, lc2_t, gap_free=True)
    one.check()
    assert torch.equal(safe, fast)
    lc2[17, :, 4:30] = np.nan                               # gaps strictly inside the grid, where the data are: the promise is broken
    one.loglike_stack2(th, torch.as_tensor(lc2, device="cuda:0"), gap_free=True)
    with pytest.raises(L.NMMAHipError, match="NMMA_STACK2_GAP_FREE"):
        one.check()
    with pytest.raises(L.NMMAHipError, match="NMMA_STACK2_GAP_FREE"):
        one.loglike_stack2(th, lc2_t)
    for e in (kn, tail):
        e.close()
    one.close()


def test_handles_without_a_one_launch_form_say_so():
    """``loglike_stack2`` returns None (the library: 2, nothing launched) for a handle not created for it, for sample_times that
    reach beyond the surrogate's grid (the flux sum is then finite where the kilonova is not: the task's window test would be
    wrong); the plugin then takes the materialising path and says so once."""
    import torch
    case = cases_combined.case_combined()
    M = len(case["filters"])
    th = torch.as_tensor(_theta(5, 8), device="cuda:0")
    one, kn, tail = _engines(case, stack_operands=0)
    assert one.loglike_stack2(th, torch.zeros((8, M, len(case["sample_times"])), dtype=torch.float64, device="cuda:0")) is None
    for e in (one, kn, tail):
        e.close()
    tt_end = float(np.max(next(iter(case["svd"].values()))["tt"]))
    beyond = np.arange(0.1, tt_end + 3.0, 0.5)
    one, kn, tail = _engines(case, sample_times=beyond)
    assert one.loglike_stack2(th, torch.zeros((8, M, len(beyond)), dtype=torch.float64, device="cuda:0")) is None
    for e in (one, kn, tail):
        e.close()


def test_one_launch_on_a_log_spaced_grid():
    """Unequally spaced sample_times (the CLI's log-spaced grid): ``em_logl<.., 8>`` -- the combined flavour with the lean task's
    bracket search for such grids -- against the materialising path, with edges and gaps in the second transient's curves."""
    import torch
    case = cases_combined.case_combined()
    M = len(case["filters"])
    logt = np.geomspace(0.1, 20.0, 40)
    one, kn, tail = _engines(case, sample_times=logt)
    rng = np.random.default_rng(88)
    for B in (7, 500, 8192):
        theta = _theta(880 + B, B)
        th = torch.as_tensor(theta, device="cuda:0")
        lc2 = rng.uniform(-17.0, -12.0, (B, M, len(logt)))
        lc2[::5, :, 0] = np.inf
        lc2[3::9, 2, 11] = np.nan
        lc2_t = torch.as_tensor(lc2, device="cuda:0")
        got = one.loglike_stack2(th, lc2_t)
        assert got is not None
        want = tail.loglike_lc_sets(th, [kn.model_lightcurves(th), lc2_t]).cpu().numpy()
        got = got.cpu().numpy()
        assert np.array_equal(got == FLOOR, want == FLOOR)
        fin = want > FLOOR
        assert fin.mean() > 0.5 and rel_err(got[fin], want[fin]).max() <= FUSED_VS_MATERIALISED_RTOL
    for e in (one, kn, tail):
        e.close()


MIX_CASES = ["c2_default", "c2_dt05", "grid_subset", "syserr_param", "fast_np6", "ncoeff7", "fast_many_filters", "fast_single_filter",
             "fixed_distance", "conversions", "conversions_cos", "real_nets", "bulla_svd", "log_grid", "nonuniform_tt", "extinction_linear",
             "hubble_sampled", "edges", "unobserved_filter_overflow", "c2_dt05_limit", "averaging", "c4_shape", "syserr_time_nodes",
             "syserr_nodes_masked"]


@pytest.mark.parametrize("name", MIX_CASES)
def test_one_launch_flavour_carries_every_feature_of_the_lean_task(name):
    """The flavour with the operand (FASTM 7 / 8) under the features of the single-model cases -- sampled systematics, extinction,
    six-parameter surrogates, two-stage grids, unequally spaced grids, conversions, a fixed distance, real networks, samples pushed
    off the model window.  With a second transient 60 mag fainter than anything the surrogate emits the flux sum IS the kilonova's
    magnitude (min(kn, m2) - g(|kn - m2|), g < 1e-20), so log L must be the single-model likelihood of the same handle's plain
    flavour -- which the golden vectors of these cases pin to the reference.  Cases whose handle has no one-launch form (the general
    task: limits, averaged bands, other than 10 coefficients) must say so (status 2 -> None, with the reason); config 4's shape
    takes the row-form lean task under the operand (the plain handle: the dense task -- agreement to 1e-12)."""
    import torch
    from nmma_amd.engine import EMEngine
    case = (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    plain = EMEngine.from_case(case)
    one = EMEngine.from_case(case, stack_operands=1)
    th = torch.as_tensor(np.ascontiguousarray(case["theta"]), device="cuda:0")
    B, M, NS = th.shape[0], len(case["model_filters"]), one.n_sample_times
    faint = torch.full((B, M, NS), 45.0, dtype=torch.float64, device="cuda:0")
    want = plain.loglike(th).cpu().numpy()
    got = one.loglike_stack2(th, faint)
    # no one-launch form: the general task (averaged bands; surrogates with other than 10 coefficients -- ncoeff7;
    # a model filter nobody observed)
    # (finite detection limits alone no longer are: since round 6 the combined-model flavours carry the truncation mass -- the plain
    #  flavours of the same handle do not, so its plain entry points refuse it and the reference value comes from `plain`)
    no_form = {"averaging": "general task", "ncoeff7": "general task", "unobserved_filter_overflow": "general task"}
    if name in no_form:
        assert got is None, name
        assert no_form[name] in one.stack2_reason, (name, one.stack2_reason)
        # ... and the handle still evaluates the single model
        assert np.array_equal(one.loglike(th).cpu().numpy(), want)
    else:
        assert got is not None, (name, getattr(one, "stack2_reason", None))
        one.check()
        got = got.cpu().numpy()
        assert np.array_equal(got == FLOOR, want == FLOOR), name
        fin = want > FLOOR
        err = rel_err(got[fin], want[fin]).max() if fin.any() else 0.0
        print(f"{name}: one-launch flavour with a dark second transient vs the plain flavour: max rel {err:.2e} over {int(fin.sum())} rows")
        assert err <= 1e-12, name
        if name in ("c2_dt05_limit", "bulla_svd", "syserr_time_nodes"):
            from nmma_amd import _lib as L
            with pytest.raises(L.NMMAHipError, match="finite detection limits or time-node systematics"):
                one.loglike(th)
        # the promise form takes the same rows through the kernel alone
        again = one.loglike_stack2(th, faint, gap_free=True).cpu().numpy()
        assert np.array_equal(again, got)
        one.check()
        # a hole in every third row's dark curves: autocomplete_data fills it from the (equally dark) neighbours, so nothing changes
        # -- but those rows now come from the re-evaluation kernel (the surrogate's curves + likelihood from curves, under the same
        # features: sampled systematics, extinction, two-stage / unequally spaced grids, six-parameter surrogates)
        holes = faint.clone()
        holes[::3, :, 1:NS - 1:4] = float("nan")
        redo = one.loglike_stack2(th, holes).cpu().numpy()
        one.check()
        assert np.array_equal(redo == FLOOR, want == FLOOR), name
        err_r = rel_err(redo[fin], want[fin]).max() if fin.any() else 0.0
        print(f"   re-evaluated rows: max rel {err_r:.2e}; {int((redo != got).sum())} of {B} rows changed bits")
        assert err_r <= FUSED_VS_MATERIALISED_RTOL, name
        keep = np.ones(B, dtype=bool); keep[::3] = False
        assert np.array_equal(redo[keep], got[keep])
    plain.close(); one.close()


UNION_MIX_CASES = [n for n in MIX_CASES if n not in ("averaging", "ncoeff7", "unobserved_filter_overflow")]


@pytest.mark.parametrize("name", UNION_MIX_CASES)
def test_union_grid_flavour_carries_every_feature_of_the_lean_task(name):
    """The same on a UNION grid (engine argument ``base_times``): the handle's grid = the case's own sample_times + the midpoints of
    every third interval + two nodes before / after them.  A lerp through extra nodes that lie ON the surrogate's piecewise-linear
    curve changes nothing, and a second transient that is dark inside the surrogate's window and absent outside it (+inf, as ``regrid``
    leaves it: ``completed=True``) contributes nothing -- so log L must again be the plain flavour's, to rounding; rows whose photometry
    leaves the surrogate's window meet nodes NEITHER sub-model covers and get their floor from the re-evaluation launch."""
    import torch
    from nmma_amd.engine import EMEngine
    case = (cases.CASES.get(name) or cases.SHAPE_CASES[name])()
    plain = EMEngine.from_case(case)
    s1 = np.asarray(plain.sample_times, float)
    mids = 0.5 * (s1[:-1] + s1[1:])[::3]
    extra = np.array([s1[0] - 0.013, s1[-1] + 0.4, s1[-1] + 1.7])
    union = np.array(sorted(set(s1.tolist()) | set(mids.tolist()) | set(extra[extra > 0].tolist())))
    one = EMEngine.from_case(dict(case, sample_times=union), stack_operands=1, base_times=s1)
    th = torch.as_tensor(np.ascontiguousarray(case["theta"]), device="cuda:0")
    B, M, NS = th.shape[0], len(case["model_filters"]), one.n_sample_times
    assert NS == len(union) and NS > len(s1)
    want = plain.loglike(th).cpu().numpy()
    # the surrogate's own curves on the union grid say where its window is (per filter: sample_times may reach beyond the SVD grid)
    kn_u = one.model_lightcurves(th[:1])[0]
    faint = torch.where(torch.isfinite(kn_u), torch.full_like(kn_u, 45.0), torch.full_like(kn_u, float("inf"))).expand(B, M, NS).contiguous()
    got = one.loglike_stack2(th, faint, completed=True)
    assert got is not None, (name, getattr(one, "stack2_reason", None))
    one.check()
    got = got.cpu().numpy()
    assert np.array_equal(got == FLOOR, want == FLOOR), name
    fin = want > FLOOR
    err = rel_err(got[fin], want[fin]).max() if fin.any() else 0.0
    print(f"{name}: union-grid flavour ({len(s1)} -> {NS} nodes) with a dark second transient vs the plain flavour: max rel {err:.2e} over {int(fin.sum())} rows")
    assert err <= 1e-11, name
    # a second transient that is finite EVERYWHERE on the union grid but 60 mag fainter: outside the surrogate's window the flux sum is
    # that transient alone -- a finite, absurdly faint curve: rows that were floored for leaving the window now have a (very negative) value
    allf = torch.full((B, M, NS), 45.0, dtype=torch.float64, device="cuda:0")
    got2 = one.loglike_stack2(th, allf, completed=True).cpu().numpy()
    one.check()
    assert np.array_equal(got2[fin], got[fin])
    with pytest.raises(Exception, match="base_times"):
        one.loglike(th)
    plain.close(); one.close()


def test_union_filter_lists_keep_the_materialising_path():
    """A combination with filters the surrogate does not list (golden ``combined_union``: a band only the second sub-model provides, an
    averaged band) has no one-launch plan; own time grids alone do (``test_own_time_grids_go_through_the_one_launch_form``)."""
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel, SVDLightCurveModel
    case = cases_combined.case_combined_union()
    kn = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["filters"], model_parameters=case["model_parameters"],
                            sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    grb = ExternalLightCurveModel("PLGRB", case["grb_filters"], case["grb_times"])
    comb = CombinedLightCurveModelContainer([kn, grb], cosmo_grid=case["cosmo_grid"])
    assert comb.stack2_plan() is None
    same = CombinedLightCurveModelContainer([kn, ExternalLightCurveModel("PLGRB", case["filters"], case["sample_times"])],
                                            cosmo_grid=case["cosmo_grid"])
    assert same.stack2_plan() is not None and same.stack2_plan()[0] is kn


# ---------------------------------------------------------------------------------------------------------------------------------
# Sub-models on their OWN time grids (model.py:1372-1374, :1440-1448): the one-launch form on the union grid (round 6; engine
# argument ``base_times``) -- golden ``combined_owngrids`` written by the reference's own container.
# ---------------------------------------------------------------------------------------------------------------------------------
def _owngrids_plugin(case):
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel, SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    kn = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=case["filters"], model_parameters=case["model_parameters"],
                            sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    grb = ExternalLightCurveModel("PLGRB", case["grb_filters"], case["grb_times"])
    comb = CombinedLightCurveModelContainer([kn, grb], cosmo_grid=case["cosmo_grid"])
    times, mags, sigmas = case["data"]
    handler = FilterSystematicsHandler(case["observed_filters"], error_budget=1.0, light_curve_times=times)
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    lik = EMTransientLikelihood(comb, (times, mags, sigmas, 0.0), handler, priors, filters=case["observed_filters"])
    _, grb_oracle = cases_combined.oracle_likelihood_owngrids(case)
    ext = np.stack([np.stack([grb_oracle.abs_lightcurves(dict(zip(case["names"], row)), case["grb_times"])[f] for f in case["grb_filters"]])
                    for row in case["theta"]])
    return lik, comb, kn, ext


def test_own_time_grids_go_through_the_one_launch_form():
    """Golden ``combined_owngrids`` (the reference's container: kilonova on 0.1 .. 14.1 d, second transient on a log grid 0.25 .. 30 d
    that lacks one filter and has interior holes for half of the rows; photometry beyond the kilonova's last node, and for early
    time shifts beyond every node of the filter only the kilonova has) through the plugin: the likelihood picks the one-launch engine
    on the UNION grid -- the surrogate's move there is in the kernel's basis rows, the operand went through ``regrid`` -- matches the
    golden log L at 1e-6 with the same floor pattern (the floored rows meet a node NEITHER sub-model covers: the re-evaluation launch
    decides them), and agrees with the materialising path (regrid of both sets + ``loglike_lc_sets``) to 1e-10."""
    import torch
    case = cases_combined.case_combined_owngrids()
    gold = cases.load_golden("combined_owngrids")["logl"]
    lik, comb, kn, ext = _owngrids_plugin(case)
    assert comb.stack2_plan() is not None and comb.stack2_plan()[0] is kn
    base, plan2 = comb.stack2_union()
    assert np.array_equal(base, case["sample_times"]) and plan2 is not None and plan2[comb.filters.index("2massh")] == []
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    sub = lik.sub_model
    assert sub._engine2 is not None and not sub._stack2_off and sub._engine is None
    assert sub._engine2.base_times is not None and sub._engine2.n_sample_times == len(comb.model_times)
    floor = gold == FLOOR
    assert floor.sum() >= 5 and (~floor).sum() >= 5
    assert np.array_equal(got == FLOOR, floor)
    err = rel_err(got[~floor], gold[~floor])
    print(f"combined_owngrids, one launch on the union grid: max rel err {err.max():.3e}")
    assert err.max() <= 1e-6
    # the materialising path on the same inputs
    sub._stack2_off = True
    mat = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    assert sub._engine is not None
    assert np.array_equal(mat == FLOOR, floor)
    assert rel_err(got[~floor], mat[~floor]).max() <= FUSED_VS_MATERIALISED_RTOL
    # single-sample reference API; a sub-model that reports "no light curve" for two rows
    sub._stack2_off = False
    ok = np.ones(len(ext), dtype=bool)
    ok[[1, 30]] = False
    got2 = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": (torch.as_tensor(ext), ok)})
    assert np.all(got2[~ok] == FLOOR) and np.array_equal(got2[ok], got[ok])


def test_union_grid_handle_curves_and_refusals():
    """A handle created with ``base_times``: ``model_lightcurves`` gives the surrogate's curves on the UNION grid (+inf outside its own
    nodes) -- the kilonova engine's curves through ``regrid`` to 1e-13 -- and the entry points that would take the surrogate alone as
    the likelihood's model refuse it; a union-grid engine on a grid that IS the surrogate's reproduces the plain one-launch form."""
    import torch
    from nmma_amd import _lib as L
    from nmma_amd.engine import EMEngine
    case = cases_combined.case_combined_owngrids()
    union = np.array(sorted(set(case["sample_times"].tolist()) | set(case["grb_times"].tolist())))
    th = torch.as_tensor(case["theta"], device="cuda:0")
    one = EMEngine(case["svd"], case["filters"], case["model_parameters"], case["names"], sample_times=union, base_times=case["sample_times"],
                   cosmo_grid=case["cosmo_grid"], data=case["data"], observed_filters=case["filters"], stack_operands=1)
    kn = EMEngine(case["svd"], case["filters"], case["model_parameters"], case["names"], sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"])
    own = kn.model_lightcurves(th)
    want = one.regrid(own, case["sample_times"], [[k] for k in range(len(case["filters"]))]).cpu().numpy()
    got = one.model_lightcurves(th).cpu().numpy()
    fin = np.isfinite(want)
    inside = (union >= case["sample_times"][0]) & (union <= case["sample_times"][-1])
    assert np.array_equal(np.isfinite(got), fin) and np.array_equal(fin, np.broadcast_to(inside, fin.shape))
    assert np.all(got[~fin] == np.inf)
    assert np.abs(got[fin] - want[fin]).max() <= 1e-12
    # the surrogate's own nodes carry the own-grid values themselves
    at = np.searchsorted(union, case["sample_times"])
    assert np.abs(got[:, :, at] - own.cpu().numpy()).max() <= 1e-12
    for call in (lambda: one.loglike(th), lambda: one.lightcurves(th), lambda: one.loglike_parts(th)):
        with pytest.raises(L.NMMAHipError, match="base_times"):
            call()
    # base_times == sample_times: the union map is the identity -- the plain one-launch form's values to a few ulp
    c3 = cases_combined.case_combined()
    same = EMEngine(c3["svd"], c3["filters"], c3["model_parameters"], c3["names"], sample_times=c3["sample_times"], base_times=c3["sample_times"],
                    cosmo_grid=c3["cosmo_grid"], data=c3["data"], observed_filters=c3["filters"], stack_operands=1)
    plain = EMEngine(c3["svd"], c3["filters"], c3["model_parameters"], c3["names"], sample_times=c3["sample_times"],
                     cosmo_grid=c3["cosmo_grid"], data=c3["data"], observed_filters=c3["filters"], stack_operands=1)
    _, grb_oracle = cases_combined.oracle_likelihood(c3)
    lc2 = torch.as_tensor(np.stack([np.stack([grb_oracle.abs_lightcurves(dict(zip(c3["names"], row)), c3["sample_times"])[f] for f in c3["filters"]])
                                    for row in c3["theta"]]), device="cuda:0")
    th3 = torch.as_tensor(c3["theta"], device="cuda:0")
    a, b = same.loglike_stack2(th3, lc2).cpu().numpy(), plain.loglike_stack2(th3, lc2).cpu().numpy()
    assert np.array_equal(a == FLOOR, b == FLOOR) and rel_err(a[a != FLOOR], b[b != FLOOR]).max() <= 1e-12
    for e in (one, kn, same, plain):
        e.close()


def test_union_grid_one_launch_is_batch_size_independent_at_config3_size():
    """8192 rows (32-sample tiles) against 16-sample tiles and ragged sub-batches of the same rows: the same bits; determinism."""
    import torch
    from nmma_amd.engine import EMEngine
    case = cases_combined.case_combined_owngrids()
    union = np.array(sorted(set(case["sample_times"].tolist()) | set(case["grb_times"].tolist())))
    one = EMEngine(case["svd"], case["filters"], case["model_parameters"], case["names"], sample_times=union, base_times=case["sample_times"],
                   cosmo_grid=case["cosmo_grid"], data=case["data"], observed_filters=case["filters"], stack_operands=1)
    B = 8192
    theta = _theta(77, B)
    theta[:, 3] = np.random.default_rng(5).uniform(-1.0, 0.1, B)       # (time shifts that keep most rows off the floor)
    _, grb_oracle = cases_combined.oracle_likelihood_owngrids(case)
    t = case["grb_times"]
    with np.errstate(divide="ignore"):
        base = theta[:, 6:7] + 2.5 * theta[:, 7:8] * np.log10(t)[None, :]
    ext = np.stack([np.where(t >= 0.3, base + 0.15 * k, np.inf) for k in range(len(case["grb_filters"]))], axis=1)
    th = torch.as_tensor(theta, device="cuda:0")
    plan = [[case["grb_filters"].index(f)] if f in case["grb_filters"] else [] for f in case["filters"]]
    lc2 = one.regrid(torch.as_tensor(ext, device="cuda:0"), t, plan)
    full = one.loglike_stack2(th, lc2, completed=True).cpu().numpy()
    again = one.loglike_stack2(th, lc2, completed=True).cpu().numpy()
    assert np.array_equal(full, again)
    assert (full != FLOOR).sum() > B // 2
    for lo, hi in ((0, 4096), (4096, 4096 + 1001), (8000, 8192)):
        part = one.loglike_stack2(th[lo:hi], lc2[lo:hi], completed=True).cpu().numpy()
        assert np.array_equal(part, full[lo:hi]), (lo, hi)
    one.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_own_time_grids_random_grids_against_the_oracle(seed):
    """Random pairs of grids through the plugin's one-launch form against the reference-pinned oracle's combined model: the kilonova's
    grid equally or unequally spaced, inside the SVD grid or reaching beyond either end of it (nodes without a value), the second
    transient's grid coarser or finer, sharing nodes with the kilonova's or not, starting before / after it, with holes for half of
    the rows -- log L at 1e-6 with the identical floor pattern."""
    import torch
    from oracle import nmma_oracle as orc
    rng = np.random.default_rng(8000 + seed)
    case = cases_combined.case_combined_owngrids(seed=9600 + seed, batch=24)
    kind = seed % 3
    if kind == 0:      # equally spaced, inside the SVD grid (0 .. 21 d)
        s1 = np.arange(0.2, 0.2 + 0.4 * rng.integers(30, 45), 0.4)
    elif kind == 1:    # unequally spaced, reaching beyond the SVD grid's end
        s1 = np.sort(np.unique(np.round(rng.uniform(0.05, 23.0, 40), 3)))
    else:              # log-spaced
        s1 = np.geomspace(0.08, 19.0, 33)
    n2 = int(rng.integers(12, 60))
    s2 = np.sort(np.unique(np.round(np.concatenate([rng.uniform(0.1, 28.0, n2), rng.choice(s1, 5, replace=False)]), 3)))
    case["sample_times"], case["grb_times"] = s1, s2
    case["grb_hole"] = (len(s2) // 3, len(s2) // 3 + 2, 1.2)
    olik, grb_oracle = cases_combined.oracle_likelihood_owngrids(case)
    want = orc.log_likelihood_batch(olik, case["names"], case["theta"])
    lik, comb, kn, ext = _owngrids_plugin(case)
    assert comb.stack2_plan() is not None
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    sub = lik.sub_model
    assert sub._engine2 is not None and not sub._stack2_off and sub._engine is None, getattr(sub._engine2, "stack2_reason", None)
    floor = want == FLOOR
    assert np.array_equal(got == FLOOR, floor), (seed, np.nonzero((got == FLOOR) != floor)[0])
    err = rel_err(got[~floor], want[~floor]).max() if (~floor).any() else 0.0
    print(f"seed {seed}: kilonova grid {len(s1)} nodes ({s1[0]:.2f} .. {s1[-1]:.2f}), second grid {len(s2)}, union {len(comb.model_times)}; "
          f"{int(floor.sum())} of {len(want)} rows floored; max rel err {err:.2e}")
    assert err <= 1e-6


# ---------------------------------------------------------------------------------------------------------------------------------
# Null filters: bands the surrogate LISTS without having a network for them (lightcurve_generation.py:168-169, "radio and X-ray filters
# when using with GRB data") -- golden ``combined_nullfilters`` written by the reference's own container.
# ---------------------------------------------------------------------------------------------------------------------------------
def _nullfilters_plugin(case, kn_times=None):
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel, SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    allf = case["all_filters"]
    kn = SVDLightCurveModel(case["model"], svd_mag_model=case["svd"], filters=allf, model_parameters=case["model_parameters"],
                            sample_times=case["sample_times"] if kn_times is None else kn_times, cosmo_grid=case["cosmo_grid"])
    grb = ExternalLightCurveModel("PLGRB", allf, case["sample_times"])
    comb = CombinedLightCurveModelContainer([kn, grb], cosmo_grid=case["cosmo_grid"])
    times, mags, sigmas = case["data"]
    handler = FilterSystematicsHandler(allf, error_budget=1.0, light_curve_times=times)
    priors = {n: SimplePrior(0.0, 1.0) for n in case["names"]}
    lik = EMTransientLikelihood(comb, (times, mags, sigmas, 0.0), handler, priors, filters=allf)
    _, grb_oracle = cases_combined.oracle_likelihood_nullfilters(case)
    ext = np.stack([np.stack([grb_oracle.abs_lightcurves(dict(zip(case["names"], row)), case["sample_times"])[f] for f in allf])
                    for row in case["theta"]])
    return lik, comb, kn, ext


def test_null_filters_go_through_the_one_launch_form():
    """Golden ``combined_nullfilters`` (the reference's container on the drivers' shared grid; 9 optical filters + a radio and an X-ray band
    the kilonova lists but has no network for; holes in the afterglow's curves for the steeper half of the slope prior; an upper limit in
    the X-ray band) through the plugin: ONE launch -- the null filters are model filters of the engine whose rows give +inf, so the
    band's flux sum is the afterglow alone -- log L at 1e-6, and the materialising path on the same inputs agrees to 1e-10."""
    import torch
    case = cases_combined.case_combined_nullfilters()
    gold = cases.load_golden("combined_nullfilters")["logl"]
    lik, comb, kn, ext = _nullfilters_plugin(case)
    assert comb.stack2_plan() is not None and comb.stack2_plan()[0] is kn and comb.stack2_union() == (None, None)
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    sub = lik.sub_model
    assert sub._engine2 is not None and not sub._stack2_off and sub._engine is None, getattr(sub._engine2, "stack2_reason", None)
    assert sub._engine2.null_filters == cases_combined.NULL_FILTERS and len(sub._engine2.model_filters) == 11
    assert not np.any(got == FLOOR) and not np.any(gold == FLOOR)
    err = rel_err(got, gold)
    print(f"combined_nullfilters, one launch: max rel err {err.max():.3e}")
    assert err.max() <= 1e-6
    sub._stack2_off = True
    mat = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    assert sub._engine is not None and rel_err(mat, gold).max() <= 1e-6
    assert rel_err(got, mat).max() <= FUSED_VS_MATERIALISED_RTOL
    # the engine's curve outputs: +inf on every node of a null filter; the plain likelihood entry points refuse the handle
    from nmma_amd import _lib as L
    th = torch.as_tensor(case["theta"], device="cuda:0")
    lc = sub._engine2.model_lightcurves(th).cpu().numpy()
    assert np.all(lc[:, 9:, :] == np.inf) and np.all(np.isfinite(lc[:, :9, :]))
    with pytest.raises(L.NMMAHipError, match="null_filters|base_times"):
        sub._engine2.loglike(th)
    # a band where the AFTERGLOW has no value either (a row whose X-ray curve is gone): nothing covers the photometry -> floor
    ext2 = ext.copy()
    ext2[5, 10, :] = np.inf
    got2 = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext2)})
    sub._stack2_off = False
    got3 = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext2)})
    assert got2[5] == FLOOR and got3[5] == FLOOR
    keep = np.arange(len(got)) != 5
    assert np.array_equal(got3[keep], got[keep])


def test_null_filters_on_own_time_grids():
    """Both extensions together: null filters AND the kilonova on its own (coarser) grid -- against the oracle's combined model."""
    import torch
    from oracle import nmma_oracle as orc
    case = cases_combined.case_combined_nullfilters(batch=24)
    kn_times = np.arange(0.1, 18.0, 0.8)
    allf = case["all_filters"]
    okn = orc.OracleSVDModel(case["model_parameters"], case["svd"], filters=allf, sample_times=kn_times, cosmo_grid=case["cosmo_grid"])
    ogrb = orc.OraclePowerLawModel(allf, case["sample_times"], cosmo_grid=case["cosmo_grid"], hole=case.get("grb_hole"))
    olik = orc.OracleLikelihood(orc.OracleCombinedModel([okn, ogrb]), case["data"], dict(mode="budget", values={f: 1.0 for f in allf}), allf,
                                detection_limit=np.inf, known_filters=allf, use_scipy=True)
    want = orc.log_likelihood_batch(olik, case["names"], case["theta"])
    lik, comb, kn, ext = _nullfilters_plugin(case, kn_times=kn_times)
    base, plan2 = comb.stack2_union()
    assert np.array_equal(base, kn_times) and plan2 is not None
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": torch.as_tensor(ext)})
    sub = lik.sub_model
    assert sub._engine2 is not None and not sub._stack2_off and sub._engine is None, getattr(sub._engine2, "stack2_reason", None)
    floor = want == FLOOR
    assert np.array_equal(got == FLOOR, floor)
    err = rel_err(got[~floor], want[~floor]).max()
    print(f"null filters on own grids: {int(floor.sum())} rows floored, max rel err {err:.2e}")
    assert err <= 1e-6


def test_finite_detection_limits_go_through_the_one_launch_form():
    """Golden ``combined_limit`` (the reference's container under per-filter detection limits, 0.3 mag above each filter's faintest
    detection; holes in the afterglow's curves for half of the rows): since round 6 the combined-model flavours of the lean task carry the
    truncated Gaussian's mass (log Phi from the table behind the flux-sum table), so the plugin stays on ONE launch -- log L at 1e-6,
    the materialising path (general term in ``em_lc_loglike``) within 1e-10; a limit BELOW a detection floors every sample in both."""
    import torch
    case = cases_combined.case_combined_limit()
    gold = cases.load_golden("combined_limit")["logl"]
    _, grb_oracle = cases_combined.oracle_likelihood(case)
    ext = torch.as_tensor(np.stack([np.stack([grb_oracle.abs_lightcurves(dict(zip(case["names"], row)), case["sample_times"])[f] for f in case["filters"]])
                                    for row in case["theta"]]))

    def plugin(limits):
        lik = _plugin(case)
        lik.sub_model.set_detection_limit(limits)
        return lik
    lik = plugin(case["detection_limit"])
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": ext})
    sub = lik.sub_model
    assert sub._engine2 is not None and not sub._stack2_off and sub._engine is None, getattr(sub._engine2, "stack2_reason", None)
    assert not np.any(gold == FLOOR) and not np.any(got == FLOOR)
    err = rel_err(got, gold)
    print(f"combined_limit, one launch: max rel err {err.max():.3e}")
    assert err.max() <= 1e-6
    sub._stack2_off = True
    mat = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": ext})
    assert sub._engine is not None and rel_err(mat, gold).max() <= 1e-6 and rel_err(got, mat).max() <= FUSED_VS_MATERIALISED_RTOL
    # the limits matter: without them the values differ
    free = plugin(np.inf).log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": ext})
    assert np.abs(free - got).max() > 1e-3
    # one filter's limit below its faintest detection: -inf for every sample
    bad = dict(case["detection_limit"])
    f0 = case["filters"][2]
    bad[f0] = float(np.max(case["data"][1][f0][np.isfinite(case["data"][2][f0])]) - 0.2)
    lik_bad = plugin(bad)
    assert np.all(lik_bad.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": ext}) == FLOOR)
    assert lik_bad.sub_model._engine2 is not None and not lik_bad.sub_model._stack2_off


def test_time_node_systematics_go_through_the_one_launch_form():
    """Golden ``combined_nodes`` (the reference's container with systematics at time nodes for two filter groups and one sampled
    parameter for the rest; non-finite sampled node values in a few rows -- one leaves 2massj a single finite node, which turns its
    detections into upper limits; holes in the afterglow's curves): since round 6 the combined-model flavours interpolate the nodes
    themselves -- ONE launch, log L at 1e-6, the materialising path within 1e-10."""
    import torch
    case = cases_combined.case_combined_nodes()
    gold = cases.load_golden("combined_nodes")["logl"]
    _, grb_oracle = cases_combined.oracle_likelihood(case)
    ext = torch.as_tensor(np.stack([np.stack([grb_oracle.abs_lightcurves(dict(zip(case["names"], row)), case["sample_times"])[f] for f in case["filters"]])
                                    for row in case["theta"]]))
    lik = _plugin(case)
    got = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": ext})
    sub = lik.sub_model
    assert sub._engine2 is not None and not sub._stack2_off and sub._engine is None, getattr(sub._engine2, "stack2_reason", None)
    floor = gold == FLOOR
    assert np.array_equal(got == FLOOR, floor)
    err = rel_err(got[~floor], gold[~floor])
    print(f"combined_nodes, one launch: max rel err {err.max():.3e} ({int(floor.sum())} rows floored)")
    assert err.max() <= 1e-6
    sub._stack2_off = True
    mat = lik.log_likelihood_batch(case["theta"], case["names"], external_lc={"PLGRB": ext})
    assert sub._engine is not None and np.array_equal(mat == FLOOR, floor)
    assert rel_err(got[~floor], mat[~floor]).max() <= FUSED_VS_MATERIALISED_RTOL


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_one_launch_form_random_feature_mixes_against_the_materialising_path(seed):
    """Random mixes of everything the one-launch form carries since round 6 -- shared or own time grids, zero to two null filters, constant /
    sampled / time-node systematics, finite detection limits on some filters, extra upper limits, holes and failed rows in the second
    transient's curves -- through the plugin twice: ONE launch against the materialising path (``em_fused`` + ``regrid`` +
    ``em_lc_loglike``, itself pinned by the goldens): identical floor patterns, log L within 1e-10."""
    import torch
    from nmma_amd.em.em_likelihood import EMTransientLikelihood
    from nmma_amd.em.model import CombinedLightCurveModelContainer, ExternalLightCurveModel, SVDLightCurveModel
    from nmma_amd.em.systematics import FilterSystematicsHandler
    rng = np.random.default_rng(31000 + seed)
    base = cases_combined.case_combined(seed=9100 + seed, batch=32)
    F = list(base["filters"])
    null = list(rng.choice(cases_combined.NULL_FILTERS, size=int(rng.integers(0, 3)), replace=False))
    allf = F + null
    times, mags, sigmas = ({k: np.array(v, float) for k, v in d.items()} for d in base["data"])
    for k, f in enumerate(null):
        n = int(rng.integers(4, 12))
        t = np.sort(rng.uniform(0.6, 13.0, n))
        times[f], sigmas[f] = t, rng.uniform(0.05, 0.2, n)
        mags[f] = -16.0 + 3.0 * np.log10(t) + 33.0 + sigmas[f] * rng.standard_normal(n)
    for f in rng.choice(allf, size=3, replace=False):           # a few more upper limits
        i = int(rng.integers(0, len(times[f])))
        sigmas[f][i] = np.inf
    st = base["sample_times"]
    own = bool(rng.integers(0, 2))
    kn_times = np.arange(0.1, 18.0, 0.7) if own else st
    grb_times = np.geomspace(0.25, 30.0, int(rng.integers(20, 45))) if own else st
    names = list(cases_combined.NAMES)
    theta = base["theta"].copy()
    sys_mode = ["budget", "param", "nodes"][seed % 3]
    if sys_mode == "budget":
        sys_file, budget = None, 1.0
    elif sys_mode == "param":
        sys_file, budget = None, None
        names, theta = names + ["em_syserr"], np.concatenate([theta, rng.uniform(0.1, 1.5, (len(theta), 1))], axis=1)
    else:
        g1 = F[:2]
        extra = ["em_syserr_rest"] + [f"em_syserr_blue_{i}" for i in range(3)]
        sys_file, budget = {"blue": {"filters": g1, "time_nodes": 3, "time_range": "lin 0.0 21.0"}, "rest": {"prior": "unused"}}, None
        names, theta = names + extra, np.concatenate([theta, rng.uniform(0.1, 1.5, (len(theta), len(extra)))], axis=1)
        theta[3, names.index("em_syserr_blue_1")] = np.nan
    limits = np.inf
    if rng.integers(0, 2):
        limits = {f: (float(np.max(mags[f][np.isfinite(sigmas[f])]) + 0.3) if rng.integers(0, 2) else np.inf) for f in allf}
    kn = SVDLightCurveModel(base["model"], svd_mag_model=base["svd"], filters=allf, model_parameters=base["model_parameters"],
                            sample_times=kn_times, cosmo_grid=base["cosmo_grid"])
    grb = ExternalLightCurveModel("PLGRB", allf, grb_times)
    comb = CombinedLightCurveModelContainer([kn, grb], cosmo_grid=base["cosmo_grid"])
    handler = FilterSystematicsHandler(allf, systematics_file=sys_file, error_budget=budget, light_curve_times=times)
    lik = EMTransientLikelihood(comb, (times, mags, sigmas, 0.0), handler, {n: SimplePrior(0.0, 1.0) for n in names}, filters=allf,
                                detection_limit=limits)
    from oracle import nmma_oracle as orc
    helper = orc.OraclePowerLawModel(allf, grb_times, hole=(len(grb_times) // 3, len(grb_times) // 3 + 2, 1.2))
    ext = np.stack([np.stack([helper.abs_lightcurves(dict(zip(names, row)), grb_times)[f] for f in allf]) for row in theta])
    ok = np.ones(len(theta), dtype=bool)
    ok[int(rng.integers(0, len(theta)))] = False
    ext_lc = {"PLGRB": (torch.as_tensor(ext), ok)}
    got = lik.log_likelihood_batch(theta, names, external_lc=ext_lc)
    sub = lik.sub_model
    assert sub._engine2 is not None and not sub._stack2_off and sub._engine is None, getattr(sub._engine2, "stack2_reason", None)
    sub._stack2_off = True
    mat = lik.log_likelihood_batch(theta, names, external_lc=ext_lc)
    assert sub._engine is not None
    floor = mat == FLOOR
    assert np.array_equal(got == FLOOR, floor), (seed, np.nonzero((got == FLOOR) != floor)[0])
    assert floor[~ok].all()
    err = rel_err(got[~floor], mat[~floor]).max() if (~floor).any() else 0.0
    print(f"seed {seed}: {'own' if own else 'shared'} grids, null {null}, systematics {sys_mode}, limits {'yes' if isinstance(limits, dict) else 'no'}: "
          f"{int(floor.sum())} of {len(theta)} rows floored, one launch vs materialising {err:.2e}")
    assert err <= FUSED_VS_MATERIALISED_RTOL


def test_null_filters_and_limits_are_batch_size_independent():
    """5000 rows of the null-filter case under finite limits (16- and 32-sample tiles, the null items ahead of the record stream): a row's
    value does not depend on the batch it arrives in, nor on the call -- the same bits from ragged sub-batches and from a second call."""
    import torch
    from nmma_amd.engine import EMEngine
    case = cases_combined.case_combined_nullfilters()
    A = case["all_filters"]
    limits = {f: float(np.max(case["data"][1][f][np.isfinite(case["data"][2][f])]) + 0.3) for f in A}
    one = EMEngine(case["svd"], A, case["model_parameters"], case["names"], sample_times=case["sample_times"], cosmo_grid=case["cosmo_grid"],
                   data=case["data"], observed_filters=A, detection_limit=limits, stack_operands=1, null_filters=cases_combined.NULL_FILTERS)
    B = 5000
    theta = _theta(91, B)
    st = case["sample_times"]
    with np.errstate(divide="ignore"):
        base = theta[:, 6:7] + 2.5 * theta[:, 7:8] * np.log10(st)[None, :]
    ext = torch.as_tensor(np.stack([np.where(st >= 0.3, base + 0.15 * k, np.inf) for k in range(len(A))], axis=1), device="cuda:0")
    th = torch.as_tensor(theta, device="cuda:0")
    full = one.loglike_stack2(th, ext).cpu().numpy()
    assert (full != FLOOR).mean() > 0.5 and np.array_equal(full, one.loglike_stack2(th, ext).cpu().numpy())
    for lo, hi in ((0, 4096), (4096, 5000), (17, 1040), (4999, 5000)):
        assert np.array_equal(one.loglike_stack2(th[lo:hi], ext[lo:hi]).cpu().numpy(), full[lo:hi]), (lo, hi)
    one.close()
