"""Stub-import harness that lets the *reference's own source* run in this container.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``nmma_amd/``) may import
this module.  It is used by ``tools/make_golden.py`` (golden-vector generation)
and by ``tests/test_oracle_vs_reference.py`` (skipped when ``/root/reference`` is
absent, i.e. on the GPU box).

The reference (``/root/reference/nmma``) is pure Python but imports a long list of
third-party packages that are absent here (bilby, astropy, sncosmo, keras, h5py,
healpy, dust_extinction, ...).  None of them does arithmetic on the hot path we
restate, so we fabricate them:

* a ``sys.meta_path`` finder returns ``MagicMock`` modules for every absent
  top-level package (and all of its submodules);
* ``bilby.core.likelihood.Likelihood`` and ``bilby.core.prior.{Prior, Constraint,
  PriorDict, ...}`` get tiny *real* stand-ins because the reference subclasses /
  isinstance-checks them (``nmma/core/base.py:11-13``, ``:56-64``, ``:133``).

With that in place ``nmma.em.{em_likelihood, model, lightcurve_generation, utils,
systematics}`` and ``nmma.core.{base, conversion}`` import unmodified.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("NMMA_REFERENCE_ROOT", "/root/reference")

_STUB_TOPLEVEL = (
    "bilby", "bilby_pipe", "astropy", "sncosmo", "keras", "tensorflow", "h5py",
    "healpy", "dust_extinction", "dustmaps", "pymultinest", "dynesty", "mpi4py",
    "schwimmbad", "afterglowpy", "lalsimulation", "lal", "jax", "fiesta",
    "configargparse", "toml", "m4opt", "wrapt_timeout_decorator", "ligo", "arviz",
    "corner", "seaborn", "tqdm_joblib", "nflows", "gwpy", "requests_toolbelt",
    "extinction", "synphot", "regions", "astroplan", "ultranest", "PyMultiNest",
    "tornado", "sqlalchemy", "penquins", "numba", "matplotlib", "scienceplots",
    "pyfftw", "dill_stub_never", "flowMC", "jaxlib", "optax", "flax", "ipdb",
)


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__path__ = []          # behave like a package so submodules resolve
        m.__loader__ = self
        return m

    def exec_module(self, module):
        return None


class _StubFinder(importlib.abc.MetaPathFinder):
    def __init__(self, names):
        self.names = set(names)

    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in self.names:
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


# ---------------------------------------------------------------------------
# Minimal real stand-ins for the bilby classes the reference subclasses.
# ---------------------------------------------------------------------------
class _Likelihood:
    """bilby.core.likelihood.Likelihood stand-in (bilby >= 2.7 signature)."""

    def __init__(self, parameters=None):
        self.parameters = parameters if parameters is not None else {}
        self._meta_data = None

    def log_likelihood(self, parameters=None):
        return float("nan")

    def noise_log_likelihood(self):
        return float("nan")

    def log_likelihood_ratio(self, parameters=None):
        return self.log_likelihood(parameters) - self.noise_log_likelihood()

    @property
    def meta_data(self):
        return self._meta_data

    @meta_data.setter
    def meta_data(self, v):
        self._meta_data = v


class _Prior:
    def __init__(self, name=None, minimum=None, maximum=None, **kw):
        self.name, self.minimum, self.maximum = name, minimum, maximum


class _Uniform(_Prior):
    pass


class _Constraint(_Prior):
    def prob(self, val):
        # bilby.core.prior.Constraint.prob: elementwise, works on arrays
        return (val > self.minimum) & (val < self.maximum)


class _PriorDict(dict):
    pass


def install():
    """Install the stubs and put the reference on ``sys.path``. Idempotent."""
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "nmma")):
        raise ImportError(f"reference tree not found at {REFERENCE_ROOT}")
    absent = []
    for name in _STUB_TOPLEVEL:
        if name in sys.modules:
            continue
        try:
            found = importlib.util.find_spec(name) is not None
        except (ImportError, ValueError):
            found = False
        if not found:
            absent.append(name)
    # matplotlib is present on some images; only stub when absent
    sys.meta_path.insert(0, _StubFinder(absent))

    import importlib as _il
    lk = _il.import_module("bilby.core.likelihood")
    lk.Likelihood = _Likelihood
    lk.JointLikelihood = _Likelihood
    pr = _il.import_module("bilby.core.prior")
    for nm, cls in dict(Prior=_Prior, Constraint=_Constraint, PriorDict=_PriorDict,
                        ConditionalPriorDict=_PriorDict, Uniform=_Uniform,
                        Interped=_Prior, MultivariateGaussianDist=_Prior,
                        MultivariateGaussian=_Prior, DeltaFunction=_Prior).items():
        setattr(pr, nm, cls)
    core = _il.import_module("bilby.core")
    core.prior = pr
    core.likelihood = lk
    # wrapt_timeout_decorator.timeout must be a transparent decorator factory
    wt = _il.import_module("wrapt_timeout_decorator")
    wt.timeout = lambda *a, **k: (lambda f: f)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # `nmma/em/__init__.py` eagerly imports CLI/training modules we do not need
    # and which drag in even more packages; pre-seed lightweight package shells so
    # only the hot-path modules are executed.
    for pkg in ("nmma", "nmma.em", "nmma.core"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *pkg.split("."))]
            sys.modules[pkg] = m
    install._done = True


def reference_modules():
    """Import and return the reference's hot-path modules (unmodified source)."""
    install()
    import importlib as _il
    import numpy as np

    # nmma.core.constants needs astropy numbers; inject CODATA-2018 values
    # (astropy>=4 definitions) before anything imports it.
    const = types.ModuleType("nmma.core.constants")
    const.msun_cgs = 1.988409870698051e33
    const.c_cgs = 2.99792458e10
    const.c_SI = 2.99792458e8
    const.c_kms = const.c_SI / 1000.0
    const.h = 6.62607015e-27
    const.kb = 1.380649e-16
    pc_cgs = 3.085677581491367e18
    const.Mpc = pc_cgs * 1e6
    const.D = 10 * pc_cgs
    const.sigSB = 5.6703744191844314e-05
    const.arad = 4 * const.sigSB / const.c_cgs
    const.eV_per_h_SI = 1.602176634e-19 / 6.62607015e-34
    const.seconds_a_day = 24 * 3600
    const.geom_msun_km = 1.476625038050125
    const.msun_to_ergs = const.msun_cgs * const.c_cgs ** 2
    const.msun_s = 4.925490947641267e-06
    const.get_cosmology = lambda: None
    const.set_cosmology = lambda *a, **k: None
    sys.modules["nmma.core.constants"] = const
    sys.modules["nmma.core"].constants = const

    mods = {}
    for name in ("nmma.core.conversion", "nmma.core.base", "nmma.em.utils",
                 "nmma.em.lightcurve_generation", "nmma.em.systematics",
                 "nmma.em.model", "nmma.em.em_likelihood"):
        mods[name.split(".")[-1]] = _il.import_module(name)
    return types.SimpleNamespace(**mods)
