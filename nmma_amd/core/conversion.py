"""Per-sample conversions on the EM path (host-side restatements used at setup time and
by the single-sample convenience API; the batched path does them on the device).

Mirrors ``nmma/core/conversion.py``: :30-34 distance modulus, :49-55 z(d_L) grid,
:57-64 get_redshift, :66-103 cosmology_to_distance, :119-126 observation_angle_conversion,
:184-192 convert_mtot_mni.
"""
from __future__ import annotations

import numpy as np

#: Planck18 flat LCDM (astropy.cosmology.Planck18 parameters: H0, Om0, Tcmb0, Neff, m_nu)
PLANCK18 = dict(H0=67.66, Om0=0.30966, Tcmb0=2.7255, Neff=3.046, m_nu=(0.0, 0.0, 0.06))


def distance_modulus_nmma(d_lum=1e-5):
    return 5.0 * (5 + np.log10(d_lum))


def observation_angle_conversion(parameters):
    theta_jn = parameters.get("theta_jn", np.arccos(parameters.get("cos_theta_jn", 1.0)))
    theta_jn = np.minimum(theta_jn, np.pi - theta_jn)
    if "KNtheta" not in parameters:
        parameters["KNtheta"] = parameters.get("inclination_EM", theta_jn) * 180.0 / np.pi
    if "inclination_EM" not in parameters:
        parameters["inclination_EM"] = parameters["KNtheta"] / 180.0 * np.pi
    return parameters


class FlatLambdaCDM:
    """Native flat LCDM luminosity distance (photons + massless/massive neutrinos as in
    astropy's FlatLambdaCDM: massive species via the Komatsu et al. fitting function).
    Replaces the astropy dependency of ``get_cosmo_grids``; parity vs astropy is
    UNPINNED (astropy is absent from the build image) -- expected agreement ~1e-6."""

    def __init__(self, H0=67.66, Om0=0.30966, Tcmb0=2.7255, Neff=3.046, m_nu=(0.0, 0.0, 0.06)):
        self.H0, self.Om0, self.Tcmb0 = H0, Om0, Tcmb0
        h = H0 / 100.0
        # photon density: 4 sigma T^4 / c^3 / rho_crit
        a_rad = 7.565733250280007e-15            # erg cm^-3 K^-4
        rho_crit = 1.878341616e-29 * h ** 2      # g cm^-3
        self.Ogamma0 = a_rad * Tcmb0 ** 4 / (2.99792458e10 ** 2) / rho_crit
        self.Neff, self.m_nu = Neff, np.asarray(m_nu, dtype=float)
        self.Tnu0 = 0.7137658555036082 * Tcmb0
        self.nu_y = self.m_nu / (8.617333262e-5 * self.Tnu0)
        self.neff_per_nu = Neff / len(self.m_nu)
        self.Onu0 = self.Ogamma0 * self._nu_rel(0.0)
        self.Ode0 = 1.0 - Om0 - self.Ogamma0 - self.Onu0

    def clone(self, **changes):
        """A copy with some of H0 / Om0 / Tcmb0 / Neff / m_nu replaced (astropy's ``clone``)."""
        args = dict(H0=self.H0, Om0=self.Om0, Tcmb0=self.Tcmb0, Neff=self.Neff, m_nu=tuple(self.m_nu))
        args.update(changes)
        return FlatLambdaCDM(**args)

    def _nu_rel(self, z):
        prefac = 0.22710731766       # 7/8 (4/11)^(4/3)
        p, invp, k = 1.83, 0.54644808743, 0.3173
        z = np.asarray(z, dtype=float)
        curr = self.nu_y[:, None] / (1.0 + np.atleast_1d(z))[None, :]
        rel = (1.0 + (k * curr) ** p) ** invp
        out = prefac * self.neff_per_nu * rel.sum(axis=0)
        return out if z.ndim else float(out[0])

    def inv_efunc(self, z):
        zp1 = 1.0 + np.asarray(z, dtype=float)
        org = self.Ogamma0 * (1.0 + self._nu_rel(z))
        return 1.0 / np.sqrt(zp1 ** 3 * (org * zp1 + self.Om0) + self.Ode0)

    def luminosity_distance(self, z):
        """Mpc; composite Simpson on 2049 nodes per redshift."""
        z = np.atleast_1d(np.asarray(z, dtype=float))
        out = np.empty_like(z)
        for i, zi in enumerate(z):
            x = np.linspace(0.0, zi, 2049)
            y = self.inv_efunc(x)
            hstep = x[1] - x[0] if zi > 0 else 0.0
            integ = hstep / 3.0 * (y[0] + y[-1] + 4 * y[1:-1:2].sum() + 2 * y[2:-1:2].sum())
            out[i] = (1.0 + zi) * 299792.458 / self.H0 * integ
        return out

    def z_at_luminosity_distance(self, d):
        lo, hi = 0.0, 1.0
        while self.luminosity_distance(hi)[0] < d:
            hi *= 2.0
        for _ in range(100):
            mid = 0.5 * (lo + hi)
            if self.luminosity_distance(mid)[0] < d:
                lo = mid
            else:
                hi = mid
        return 0.5 * (lo + hi)


def native_cosmology(cosmology=None):
    """Map whatever a prior carries as ``cosmology`` onto the native :class:`FlatLambdaCDM`:
    None / a name -> Planck18 (nmma/core/constants.py:43); an object with the native interface is
    used as is; an astropy-like flat cosmology is rebuilt from its H0 / Om0 / Tcmb0 / Neff / m_nu."""
    if cosmology is None or isinstance(cosmology, str):
        return FlatLambdaCDM(**PLANCK18)
    if hasattr(cosmology, "z_at_luminosity_distance"):
        return cosmology

    def val(x):
        return getattr(x, "value", x)
    try:
        kw = dict(H0=float(val(cosmology.H0)), Om0=float(val(cosmology.Om0)))
        for key in ("Tcmb0", "Neff"):
            if hasattr(cosmology, key):
                kw[key] = float(val(getattr(cosmology, key)))
        if hasattr(cosmology, "m_nu"):
            kw["m_nu"] = tuple(np.atleast_1d(val(cosmology.m_nu)).astype(float))
        return FlatLambdaCDM(**kw)
    except (AttributeError, TypeError, ValueError) as exc:
        raise ValueError(f"cannot map cosmology {cosmology!r} onto the native FlatLambdaCDM") from exc


def redshift_at_distance(d_lum, cosmo_grid=None, cosmology=None):
    """z of ONE fixed luminosity distance: through the caller's grid when it covers the distance
    (what the reference's redshift_from_dlum does), else the native root-find (get_redshift)."""
    d_lum = float(d_lum)
    if cosmo_grid is not None and cosmo_grid[0][0] <= d_lum <= cosmo_grid[0][-1]:
        return float(np.interp(d_lum, cosmo_grid[0], cosmo_grid[1]))
    return float(native_cosmology(cosmology).z_at_luminosity_distance(d_lum))


def get_cosmo_grids(distance_min, distance_max, cosmology=None, n=50):
    """(dist_grid, z_grid) with the layout of nmma/core/conversion.py:49-55."""
    cosmology = native_cosmology(cosmology)
    zmin = cosmology.z_at_luminosity_distance(distance_min)
    zmax = cosmology.z_at_luminosity_distance(distance_max)
    z_grid = np.geomspace(zmin, zmax, n)
    return cosmology.luminosity_distance(z_grid), z_grid


def _symbolic(*values):
    """Is any of the values a traced column (nmma_amd.core.constraints lowers Constraint priors by running these conversions on
    symbolic columns)?  A root-find or a table look-up has no device expression: it then returns an opaque symbol."""
    from .constraints import is_symbolic
    return any(is_symbolic(v) for v in values)


def luminosity_distance_to_redshift(distance, cosmology=None):
    """z(d_L [Mpc]) by root-finding on the native cosmology; scalar or array."""
    if _symbolic(distance):
        from .constraints import opaque_like
        return opaque_like(distance)
    cosmo = native_cosmology(cosmology)
    d = np.asarray(distance, dtype=float)
    if d.ndim == 0:
        return float(cosmo.z_at_luminosity_distance(float(d)))
    return np.array([cosmo.z_at_luminosity_distance(float(x)) for x in d.ravel()]).reshape(d.shape)


def cosmology_to_distance(parameters, cosmology=None):
    """Fill in ``redshift`` from ``luminosity_distance`` (or the reverse) under a cosmology whose H0 / Om0
    are overridden by ``Hubble_constant`` / ``Omega_matter`` when those are in ``parameters``
    (nmma/core/conversion.py:66-103).  Values may be scalars or equal-length arrays (one cosmology per row)."""
    base = native_cosmology(cosmology)
    over = {}
    if "Hubble_constant" in parameters:
        over["H0"] = parameters["Hubble_constant"]
    if "Omega_matter" in parameters:
        over["Om0"] = parameters["Omega_matter"]
    if "luminosity_distance" not in parameters and "redshift" not in parameters:
        raise KeyError("Either redshift or luminosity_distance must be in parameters")
    if _symbolic(parameters.get("luminosity_distance"), parameters.get("redshift"), *over.values()):
        from .constraints import opaque_like
        parameters["redshift" if "luminosity_distance" in parameters else "luminosity_distance"] = opaque_like(None)
        return parameters

    def variant(**kw):
        return base.clone(**{k: float(v) for k, v in kw.items()})

    def one(cosmo, row):
        if "luminosity_distance" in parameters:
            return cosmo.z_at_luminosity_distance(float(row["luminosity_distance"]))
        return float(cosmo.luminosity_distance(float(row["redshift"]))[0])

    target = "redshift" if "luminosity_distance" in parameters else "luminosity_distance"
    given = "luminosity_distance" if target == "redshift" else "redshift"
    per_row = [k for k, v in over.items() if np.ndim(v) > 0]
    if not per_row:
        cosmo = variant(**over)
        vals = np.asarray(parameters[given], dtype=float)
        if vals.ndim == 0:
            parameters[target] = one(cosmo, {given: vals})
        else:
            parameters[target] = np.array([one(cosmo, {given: v}) for v in vals])
        return parameters
    n = len(np.atleast_1d(over[per_row[0]]))
    rows = np.broadcast_to(np.asarray(parameters[given], dtype=float), (n,))
    parameters[target] = np.array([
        one(variant(**{k: (np.broadcast_to(v, (n,))[i]) for k, v in over.items()}), {given: rows[i]})
        for i in range(n)])
    return parameters


def convert_mtot_mni(params):
    """Derived quantities of the AnBa2022 supernova grids (nmma/core/conversion.py:184-192): linear masses
    from their log10 aliases, the nickel fraction ``mni_c`` and the mixing margin ``mrp_c`` that
    the priors constrain.  Works on scalars and on columns."""
    for par in ("mni", "mtot", "mrp"):
        if par not in params:
            params[par] = np.power(10.0, params[f"log10_{par}"])
    params["mni_c"] = params["mni"] / params["mtot"]
    params["mrp_c"] = params["xmix"] * (params["mtot"] - params["mni"]) - params["mrp"]
    return params


# ---------------------------------------------------------------------------------------------------------------------
# compact-binary conversions of the GW messenger (reference core/conversion.py:104-173; bilby/gw/conversion.py)
# All of them work on floats and on columns (numpy arrays) alike.
# ---------------------------------------------------------------------------------------------------------------------
def tidal_deformabilities_and_mass_ratio_to_eff_tidal_deformabilities(lambda1, lambda2, q):
    """(lambda_tilde, delta_lambda_tilde) from the component deformabilities (reference :164-173)."""
    eta = q / np.power(1. + q, 2.)
    eta2, eta3 = eta * eta, eta * eta * eta
    root14eta = np.sqrt(1. - 4 * eta)
    lam_t = (8. / 13.) * ((1. + 7 * eta - 31 * eta2) * (lambda1 + lambda2)
                          + root14eta * (1. + 9 * eta - 11. * eta2) * (lambda1 - lambda2))
    dlam_t = 0.5 * (root14eta * (1. - 13272. * eta / 1319. + 8944. * eta2 / 1319.) * (lambda1 + lambda2)
                    + (1. - 15910. * eta / 1319. + 32850. * eta2 / 1319. + 3380. * eta3 / 1319.) * (lambda1 - lambda2))
    return lam_t, dlam_t


def generate_mass_parameters(p):
    """Whatever two mass parameters are present -> mass_1, mass_2, total_mass, chirp_mass, mass_ratio,
    symmetric_mass_ratio (bilby: generate_mass_parameters / generate_component_masses)."""
    if "mass_1" not in p or "mass_2" not in p:
        if "chirp_mass" in p and "mass_ratio" in p:
            q = p["mass_ratio"]
            total = p["chirp_mass"] * (1 + q) ** 1.2 / q ** 0.6
            p["mass_1"] = total / (1 + q)
            p["mass_2"] = p["mass_1"] * q
        elif "chirp_mass" in p and "symmetric_mass_ratio" in p:
            eta = np.minimum(p["symmetric_mass_ratio"], 0.25)
            q = (1 - 2 * eta - np.sqrt(np.maximum(1 - 4 * eta, 0.0))) / (2 * eta)
            total = p["chirp_mass"] / eta ** 0.6
            p["mass_1"] = total / (1 + q)
            p["mass_2"] = p["mass_1"] * q
        elif "total_mass" in p and "mass_ratio" in p:
            p["mass_1"] = p["total_mass"] / (1 + p["mass_ratio"])
            p["mass_2"] = p["mass_1"] * p["mass_ratio"]
        elif "mass_1" in p and "mass_ratio" in p:
            p["mass_2"] = p["mass_1"] * p["mass_ratio"]
        elif "mass_2" in p and "mass_ratio" in p:
            p["mass_1"] = p["mass_2"] / p["mass_ratio"]
        else:
            return p
    m1, m2 = p["mass_1"], p["mass_2"]
    p.setdefault("total_mass", m1 + m2)
    p.setdefault("mass_ratio", m2 / m1)
    p.setdefault("symmetric_mass_ratio", m1 * m2 / (m1 + m2) ** 2)
    p.setdefault("chirp_mass", (m1 * m2) ** 0.6 / (m1 + m2) ** 0.2)
    return p


def _aligned_spin_conversion(p):
    """chi_i -> (a_i, cos_tilt_i) and cos_theta_jn -> theta_jn, as convert_to_lal_binary_black_hole_parameters does for
    the aligned-spin parametrisation."""
    for idx in ("1", "2"):
        key = f"chi_{idx}"
        if key in p:
            p.setdefault(f"a_{idx}", np.abs(p[key]))
            p.setdefault(f"cos_tilt_{idx}", np.sign(p[key]))
    if "cos_theta_jn" in p and "theta_jn" not in p:
        p["theta_jn"] = np.arccos(p["cos_theta_jn"])
    return p


def source_frame_masses(p, cosmology=None):
    """Reference :104-117: mass parameters, the redshift of the luminosity distance when none is given, source-frame masses."""
    p = generate_mass_parameters(p)
    if "redshift" not in p and "luminosity_distance" in p:
        p["redshift"] = luminosity_distance_to_redshift(p["luminosity_distance"], cosmology)
    if "redshift" in p and "mass_1" in p:
        z = p["redshift"]
        wrap = (lambda x: x) if _symbolic(z, p["mass_1"], p["mass_2"]) else np.array
        p.setdefault("mass_1_source", wrap(p["mass_1"] / (1 + z)))
        p.setdefault("mass_2_source", wrap(p["mass_2"] / (1 + z)))
    return p


def bbh_source_frame(params):
    """Reference :131-134 (bilby's convert_to_lal_binary_black_hole_parameters for aligned spins, then source_frame_masses)."""
    return source_frame_masses(_aligned_spin_conversion(dict(params)))


def bns_source_frame(params):
    """Reference :136-139: the black-hole conversion plus the tidal parameters -- (lambda_tilde, delta_lambda_tilde) are
    turned into component deformabilities when those are what is sampled (bilby:
    lambda_tilde_delta_lambda_tilde_to_lambda_1_lambda_2)."""
    p = source_frame_masses(_aligned_spin_conversion(dict(params)))
    if "lambda_1" not in p and "lambda_tilde" in p and "mass_1" in p:
        eta = p["symmetric_mass_ratio"]
        lt, dlt = p["lambda_tilde"], p.get("delta_lambda_tilde", 0.0)
        root = np.sqrt(np.maximum(1 - 4 * eta, 0.0))
        a = (8 / 13) * (1 + 7 * eta - 31 * eta ** 2)
        b = (8 / 13) * root * (1 + 9 * eta - 11 * eta ** 2)
        c = 0.5 * root * (1 - 13272 / 1319 * eta + 8944 / 1319 * eta ** 2)
        d = 0.5 * (1 - 15910 / 1319 * eta + 32850 / 1319 * eta ** 2 + 3380 / 1319 * eta ** 3)
        # [lt, dlt] = [[a, b], [c, d]] [lambda_1 + lambda_2, lambda_1 - lambda_2]
        det = a * d - b * c
        s, dl = (d * lt - b * dlt) / det, (-c * lt + a * dlt) / det
        p["lambda_1"], p["lambda_2"] = 0.5 * (s + dl), 0.5 * (s - dl)
    return p


class MultimessengerConversion:
    """``nmma/core/conversion.py:768-824``: the ordered chain of parameter conversions a joint likelihood applies -- cosmology
    (distance from redshift / Hubble constant), GW source frame, EOS, kilonova ejecta fits, EM, custom -- each a callable
    ``parameters -> parameters``.  The container, its ordering and ``convert_to_multimessenger_parameters`` are the reference's;
    the EOS converter and ``KilonovaEjectaFitting`` (NR fit formulae over EOS tables, SURVEY section 2: out of scope) are not
    built here, so ``'ejecta'`` needs the callable itself (``{'ejecta': my_fit}``) instead of ``True``.  Works on scalars (one
    sample) and on arrays (the batched constraint check of ``MultiMessengerLikelihood.log_likelihood_batch``)."""

    def __init__(self, *conversions):
        self._conversions = conversions

    @classmethod
    def from_args(cls, args):
        raise NotImplementedError("from_args not yet implemented")          # (as in the reference, :773-776)

    @classmethod
    def from_dict(cls, instruction_dict):
        conversions = []
        # NOTE: Order matters!!!  (:780)
        if "cosmo" in instruction_dict:
            cosmo = native_cosmology(instruction_dict["cosmo"])
            conversions.append(lambda p, _c=cosmo: cosmology_to_distance(p, cosmology=_c))
        if "gw" in instruction_dict:
            conversions.append(instruction_dict["gw"])
        if "eos" in instruction_dict:
            conversions.append(instruction_dict["eos"])
        if "ejecta" in instruction_dict:
            fit = instruction_dict["ejecta"]
            if not callable(fit):
                raise NotImplementedError("KilonovaEjectaFitting (NR ejecta fits over an EOS) is not part of this build: "
                                          "pass the fitting callable itself under 'ejecta'")
            conversions.append(fit)
        if "em" in instruction_dict:
            conversions.append(instruction_dict["em"])
        if "custom" in instruction_dict:
            conversions.append(instruction_dict["custom"])
        return cls(*conversions)

    @classmethod
    def basic_cbc(cls, eos_conversion, em_conversion, ejecta_fit=None):
        if ejecta_fit is None:
            raise NotImplementedError("KilonovaEjectaFitting is not part of this build: pass ejecta_fit")
        return cls(bbh_source_frame, eos_conversion, ejecta_fit, em_conversion)

    @staticmethod
    def _scalar(v):
        """``val_to_scalar`` (reference :19-27): single-value quantities become Python scalars."""
        if _symbolic(v) or isinstance(v, (str, bytes)):
            return v
        a = np.asarray(v)
        return a.item() if a.size == 1 and a.ndim <= 1 and a.dtype != object else v

    def convert_to_multimessenger_parameters(self, parameters, add_new_keys=False, batched=False):
        """``batched=True`` (the constraint check of a BATCH of samples, columns in and columns out): nothing is collapsed to a
        scalar -- a batch of one row keeps its shape-(1,) columns, which the messengers' converters index."""
        original_keys = list(parameters.keys())
        scalar = (lambda v: v) if batched else self._scalar
        converted = {k: scalar(v) for k, v in parameters.items()}
        converted = self.core_conversion(converted)
        converted = {k: scalar(v) for k, v in converted.items()}
        if add_new_keys:
            return converted, [k for k in converted if k not in original_keys]
        return converted

    def core_conversion(self, parameters):
        for conv in self._conversions:
            out = conv(parameters)
            parameters = out[0] if isinstance(out, tuple) else out       # (bilby-style converters return (parameters, added_keys))
        return parameters

    def identity_conversion(self, parameters):
        return parameters
