"""CPU restatement of the gravitational-wave leg of BASELINE config 5: frequency-domain
IMRPhenomD_NRTidalv2 waveform -> detector response -> noise-weighted inner products -> log-likelihood ratio.

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``tools/`` and ``bench``-side CPU baselines, never by ``nmma_amd/``.

PARITY UNPINNED -- read this before trusting a number.  ``nmma/gw/gw_likelihood.py:97-247`` holds no arithmetic: it hands
``interferometers``, ``waveform_generator`` and the marginalisation flags (:164, :174-178) to
``bilby.gw.likelihood.GravitationalWaveTransient`` (:185-203), whose waveform comes from ``lalsimulation``
(``bilby.gw.source.lal_binary_neutron_star`` -> ``SimInspiralChooseFDWaveform``, approximant ``IMRPhenomD_NRTidalv2``).
Neither bilby (>= 2.7.1) nor lalsimulation is in the build image and the reference ships no golden vector for this leg
(``tests/joint_analysis_pipeline.py:75-82`` has the GW arguments commented out).  What follows therefore restates the
PUBLISHED algorithms from memory of the papers and the public LALSuite / bilby sources:

* IMRPhenomD: Husa et al. PRD 93 044006 (arXiv:1508.07250), Khan et al. PRD 93 044007 (arXiv:1508.07253) -- ansaetze of
  section V-VII, Table V fit coefficients, as implemented in ``LALSimIMRPhenomD_internals.c``.
* final state: the ``FinalSpin0815`` / ``EradRational0815`` fits of the same papers.
* ring-down frequency and damping: LAL interpolates a 1003-point table of the l = m = 2, n = 0 Kerr quasi-normal mode
  (absent here); this file uses the published fit of Berti, Cardoso & Will PRD 73 064030 Table VIII
  (F = f1 + f2 (1-j)^f3, Q = q1 + q2 (1-j)^q3), accurate to a few 1e-3 -- a STATED DEVIATION, irrelevant below ~2 kHz for
  neutron-star masses.
* TaylorF2 aligned-spin phasing to 3.5PN incl. spin-induced quadrupole terms: ``LALSimInspiralPNCoefficients.c``
  (Arun et al. 2009, Mishra et al. 2016, Krishnendu et al. 2017).
* NRTidalv2: Dietrich et al. PRD 100 044003 (arXiv:1905.06011) eqs. (17)-(24), ``LALSimNRTunedTides.c``; universal
  relations of Yagi & Yunes (2017) for the spin-induced quadrupole / octupole.  The tidal AMPLITUDE term is taken in the
  normalisation of eq. (24) of the paper (strain units); LAL's own scaling of that term could not be checked here.
* detector response and likelihood: ``bilby/gw/detector/interferometer.py`` (get_detector_response), ``geometry.py``,
  ``bilby_cython`` (antenna pattern, time delay, GMST), ``bilby/gw/likelihood/base.py`` (calculate_snrs,
  log_likelihood_ratio, phase marginalisation ``ln I0(|<d|h>|) - <h|h>/2``), ``bilby/gw/utils.py``
  (noise_weighted_inner_product).

The Table V digits and the PN expressions were typed from memory; a wrong digit would make this oracle and the HIP kernel
(which holds its own copy of the table in ``nmma_amd/csrc/gw_math.h``) disagree only if the two copies differ -- i.e. the
GPU tests pin HIP-vs-oracle consistency and NOT physical correctness against lalsimulation.  Only three pieces are pinned
against installed software: ``ln I0`` (scipy.special.ive), the inner products (extended-precision ``math.fsum`` /
longdouble evaluation in the tests) and the reference wrapper's control flow (``tests/test_gw_wrapper_reference.py``).
"""
from __future__ import annotations

import math

import numpy as np

# ------------------------------------------------------------------------------------------------------------------
# constants (LALConstants.h values)
# ------------------------------------------------------------------------------------------------------------------
PI = math.pi
GAMMA = 0.5772156649015329
MTSUN_SI = 4.925490947641266978e-06      # G M_sun / c^3 [s]
MRSUN_SI = 1.476625038050124729e+03      # G M_sun / c^2 [m]
PC_SI = 3.085677581491367e16
C_SI = 299792458.0

F_CUT = 0.2              # Mf above which IMRPhenomD returns zero
AMP_FJOIN_INS = 0.014    # Mf: inspiral -> intermediate amplitude
PHI_FJOIN_INS = 0.018    # Mf: inspiral -> intermediate phase

# Khan et al. 2016, Table V.  Row = coefficient, columns = the 11 numbers of
#   c0 + c1 eta + (chi-1) (c2 + c3 eta + c4 eta^2) + (chi-1)^2 (c5 + c6 eta + c7 eta^2) + (chi-1)^3 (c8 + c9 eta + c10 eta^2)
# with chi = chi_PN.
PHENOMD_TABLE = {
    "rho1": [3931.8979897196696, -17395.758706812805, 3132.375545898835, 343965.86092361377, -1.2162565819981997e6,
             -70698.00600428853, 1.383907177859705e6, -3.9662761890979446e6, -60017.52423652596, 803515.1181825735,
             -2.091710365941658e6],
    "rho2": [-40105.47653771657, 112253.0169706701, 23561.696065836168, -3.476180699403351e6, 1.137593670849482e7,
             754313.1127166454, -1.308476044625268e7, 3.6444584853928134e7, 596226.612472288, -7.4277901143564405e6,
             1.8928977514040343e7],
    "rho3": [83208.35471266537, -191237.7264145924, -210916.2454782992, 8.71797508352568e6, -2.6914942420669552e7,
             -1.9889806527362722e6, 3.0888029960154563e7, -8.390870279256162e7, -1.4535031953446497e6,
             1.7063528990822166e7, -4.2748659731120914e7],
    "v2": [0.8149838730507785, 2.5747553517454658, 1.1610198035496786, -2.3627771785551537, 6.771038707057573,
           0.7570782938606834, -2.7256896890432474, 7.1140380397149965, 0.1766934149293479, -0.7978690983168183,
           2.1162391502005153],
    "gamma1": [0.006927402739328343, 0.03020474290328911, 0.006308024337706171, -0.12074130661131138,
               0.26271598905781324, 0.0034151773647198794, -0.10779338611188374, 0.27098966966891747,
               0.0007374185938559283, -0.02749621038376281, 0.0733150789135702],
    "gamma2": [1.010344404799477, 0.0008993122007234548, 0.283949116804459, -4.049752962958005, 13.207828172665366,
               0.10396278486805426, -7.025059158961947, 24.784892370130475, 0.03093202475605892, -2.6924023896851663,
               9.609374464684983],
    "gamma3": [1.3081615607036106, -0.005537729694807678, -0.06782917938621007, -0.6689834970767117, 3.403147966134083,
               -0.05296577374411866, -0.9923793203111362, 4.820681208409587, -0.006134139870393713,
               -0.38429253308696365, 1.7561754421985984],
    "sigma1": [2096.551999295543, 1463.7493168261553, 1312.5493286098522, 18307.330017082117, -43534.1440746107,
               -833.2889543511114, 32047.31997183187, -108609.45037520859, 452.25136398112204, 8353.439546391714,
               -44531.3250037322],
    "sigma2": [-10114.056472621156, -44631.01109458185, -6541.308761668722, -266959.23419307504, 686328.3229317984,
               3405.6372187679685, -437507.7208209015, 1.6318171307344697e6, -7462.648563007646, -114585.25177153319,
               674402.4689098676],
    "sigma3": [22933.658273436497, 230960.00814979506, 14961.083974183695, 1.1940181342318142e6, -3.1042239693052764e6,
               -3038.166617199259, 1.8720322849093592e6, -7.309145012085539e6, 42738.22871475411, 467502.018616601,
               -3.064853498512499e6],
    "sigma4": [-14621.71522218357, -377812.8579387104, -9608.682631509726, -1.7108925257214056e6, 4.332924601416521e6,
               -22366.683262266528, -2.5019716386377467e6, 1.0274495902259542e7, -85360.30079034246,
               -570025.3441737515, 4.396844346849777e6],
    "beta1": [97.89747327985583, -42.659730877489224, 153.48421037904913, -1417.0620760768954, 2752.8614143665027,
              138.7406469558649, -1433.6585075135881, 2857.7418952430758, 41.025109467376126, -423.680737974639,
              850.3594335657173],
    "beta2": [-3.282701958759534, -9.051384468245866, -12.415449742258042, 55.4716447709787, -106.05109938966335,
              -11.953044553690658, 76.80704618365418, -155.33172948098394, -3.4129261592393263, 25.572377569952536,
              -54.408036707740465],
    "beta3": [-0.000025156429818799565, 0.000019750256942201327, -0.000018370671469295915, 0.000021886317041311973,
              0.00008250240316860033, 7.157371250566708e-6, -0.000055780000112270685, 0.00019142082884072178,
              5.447166261464217e-6, -0.00003220610095021982, 0.00007974016714984341],
    "alpha1": [43.31514709695348, 638.6332679188081, -32.85768747216059, 2415.8938269370315, -5766.875169379177,
               -61.85459307173841, 2953.967762459948, -8986.29057591497, -21.571435779762044, 981.2158224673428,
               -3239.5664895930286],
    "alpha2": [-0.07020209449091723, -0.16269798450687084, -0.1872514685185499, 1.138313650449945, -2.8334196304430046,
               -0.17137955686840617, 1.7197549338119527, -4.539717148261272, -0.049983437357548705, 0.6062072055948309,
               -1.682769616644546],
    "alpha3": [9.5988072383479, -397.05438595557433, 16.202126189517813, -1574.8286986717037, 3600.3410843831093,
               27.092429659075467, -1786.482357315139, 5152.919378666511, 11.175710130033895, -577.7999423177481,
               1808.730762932043],
    "alpha4": [-0.02989487384493607, 1.4022106448583738, -0.07356049468633846, 0.8337006542278661, 0.2240008282397391,
               -0.055202870001177226, 0.5667186343606578, 0.7186931973380503, -0.015507437354325743,
               0.15750322779277187, 0.21076815715176228],
    "alpha5": [0.9974408278363099, -0.007884449714907203, -0.059046901195591035, 1.3958712396764088, -4.516631601676276,
               -0.05585343136869692, 1.7516580039343603, -5.990208965347804, -0.017945336522161195, 0.5965097794825992,
               -2.0608879367971804],
}
PHENOMD_ORDER = ["rho1", "rho2", "rho3", "v2", "gamma1", "gamma2", "gamma3", "sigma1", "sigma2", "sigma3", "sigma4",
                 "beta1", "beta2", "beta3", "alpha1", "alpha2", "alpha3", "alpha4", "alpha5"]


def table_fit(name, eta, chi_pn):
    """Khan et al. 2016 eq. (31): the bi-polynomial fit of one phenomenological coefficient."""
    c = PHENOMD_TABLE[name]
    xi = chi_pn - 1.0
    return (c[0] + c[1] * eta
            + xi * (c[2] + c[3] * eta + c[4] * eta * eta)
            + xi * xi * (c[5] + c[6] * eta + c[7] * eta * eta)
            + xi * xi * xi * (c[8] + c[9] * eta + c[10] * eta * eta))


# ------------------------------------------------------------------------------------------------------------------
# final state (LALSimIMRPhenomD_internals.c: FinalSpin0815, EradRational0815, fring, fdamp)
# ------------------------------------------------------------------------------------------------------------------
def final_spin_0815(eta, chi1, chi2):
    seta = math.sqrt(max(1.0 - 4.0 * eta, 0.0))
    m1, m2 = 0.5 * (1.0 + seta), 0.5 * (1.0 - seta)
    s = m1 * m1 * chi1 + m2 * m2 * chi2
    eta2, eta3 = eta * eta, eta ** 3
    eta4 = eta2 * eta2
    s2, s3, s4 = s * s, s ** 3, s ** 4
    return (3.4641016151377544 * eta - 4.399247300629289 * eta2 + 9.397292189321194 * eta3 - 13.180949901606242 * eta4
            + (1 - 0.0850917821418767 * eta - 5.837029316602263 * eta2) * s
            + (0.1014665242971878 * eta - 2.0967746996832157 * eta2) * s2
            + (-1.3546806617824356 * eta + 4.108962025369336 * eta2) * s3
            + (-0.8676969352555539 * eta + 2.064046835273906 * eta2) * s4)


def erad_rational_0815(eta, chi1, chi2):
    seta = math.sqrt(max(1.0 - 4.0 * eta, 0.0))
    m1, m2 = 0.5 * (1.0 + seta), 0.5 * (1.0 - seta)
    s = (m1 * m1 * chi1 + m2 * m2 * chi2) / (m1 * m1 + m2 * m2)
    eta2, eta3 = eta * eta, eta ** 3
    return ((0.055974469826360077 * eta + 0.5809510763115132 * eta2 - 0.9606726679372312 * eta3
             + 3.352411249771192 * eta3 * eta)
            * (1. + (-0.0030302335878845507 - 2.0066110851351073 * eta + 7.7050567802399215 * eta2) * s)
            / (1. + (-0.6714403054720589 - 1.4756929437702908 * eta + 7.304676214885011 * eta2) * s))


def qnm_220(final_spin):
    """(M_final * f_ring, M_final * f_damp) of the l = m = 2, n = 0 Kerr mode: Berti, Cardoso & Will 2006, Table VIII."""
    j = min(max(final_spin, -0.999), 0.999)
    fr = (1.5251 - 1.1568 * (1.0 - j) ** 0.1292) / (2.0 * PI)
    q = 0.7000 + 1.4187 * (1.0 - j) ** (-0.4990)
    return fr, fr / (2.0 * q)


# ------------------------------------------------------------------------------------------------------------------
# universal relations used by NRTidalv2 (LALSimUniversalRelations.c)
# ------------------------------------------------------------------------------------------------------------------
def quadrupole_from_lambda(lam):
    """Spin-induced quadrupole parameter from the tidal deformability (Yagi & Yunes 2017, eq. 15; LAL's low-lambda branch)."""
    if lam < 1.0:
        return 1.0 + lam * (0.427688866723244 + lam * (-0.324336526985068 + lam * 0.1107439432180572))
    ll = math.log(lam)
    return math.exp(0.1940 + 0.09163 * ll + 0.04812 * ll * ll - 4.283e-3 * ll ** 3 + 1.245e-4 * ll ** 4)


def octupole_from_quadrupole(qm):
    """Spin-induced octupole from the quadrupole (Yagi & Yunes 2017; LAL: ...OctupoleVSSpinInducedQuadrupole)."""
    lq = math.log(qm)
    return math.exp(0.003131 + 2.071 * lq - 0.7152 * lq * lq + 0.2458 * lq ** 3 - 0.03309 * lq ** 4)


# ------------------------------------------------------------------------------------------------------------------
# per-source setup
# ------------------------------------------------------------------------------------------------------------------
class PhenomDNRTidalv2:
    """All frequency-independent quantities of one source (masses in solar masses, detector frame; aligned spins)."""

    def __init__(self, mass_1, mass_2, chi_1, chi_2, lambda_1, lambda_2, tidal=True):
        """``tidal=False``: plain IMRPhenomD (no tidal terms, black-hole multipoles, no taper)."""
        self.tidal = tidal
        if mass_1 < mass_2:            # LAL swaps so that body 1 is the heavier one
            mass_1, mass_2, chi_1, chi_2, lambda_1, lambda_2 = mass_2, mass_1, chi_2, chi_1, lambda_2, lambda_1
        self.m1, self.m2, self.chi1, self.chi2, self.lam1, self.lam2 = mass_1, mass_2, chi_1, chi_2, lambda_1, lambda_2
        M = mass_1 + mass_2
        self.M = M
        self.M_sec = M * MTSUN_SI
        eta = mass_1 * mass_2 / (M * M)
        self.eta = eta = min(eta, 0.25)
        seta = math.sqrt(max(1.0 - 4.0 * eta, 0.0))
        chi_s, chi_a = 0.5 * (chi_1 + chi_2), 0.5 * (chi_1 - chi_2)
        self.chi_pn = chi_s * (1.0 - eta * 76.0 / 113.0) + seta * chi_a
        fit = {k: table_fit(k, eta, self.chi_pn) for k in PHENOMD_ORDER}
        self.fit = fit
        # ---- final state
        fs = final_spin_0815(eta, chi_1, chi_2)
        erad = erad_rational_0815(eta, chi_1, chi_2)
        fr, fd = qnm_220(fs)
        self.fRD, self.fDM = fr / (1.0 - erad), fd / (1.0 - erad)
        # ---- spin-induced multipoles (NRTidalv2 passes quadparam - 1 to the PN phasing)
        if tidal:
            self.qm1, self.qm2 = quadrupole_from_lambda(lambda_1), quadrupole_from_lambda(lambda_2)
            self.oct1, self.oct2 = octupole_from_quadrupole(self.qm1), octupole_from_quadrupole(self.qm2)
        else:
            self.qm1 = self.qm2 = self.oct1 = self.oct2 = 1.0
        self._setup_amplitude(seta)
        self._setup_phase()
        self._setup_tides()

    # ---- amplitude (Khan et al. section VI; LAL init_amp_ins_prefactors, AmpIntColFitCoeff, ComputeDeltasFromCollocation)
    def _setup_amplitude(self, seta):
        eta, chi1, chi2 = self.eta, self.chi1, self.chi2
        eta2, eta3 = eta * eta, eta ** 3
        chi12, chi22 = chi1 * chi1, chi2 * chi2
        sp1 = 1.0 + seta
        pi2 = PI * PI
        self.amp0 = math.sqrt(2.0 * eta / 3.0) * PI ** (-1.0 / 6.0)
        a = {}
        a[2] = ((-969 + 1804 * eta) * PI ** (2.0 / 3.0)) / 672.0
        a[3] = ((chi1 * (81 * sp1 - 44 * eta) + chi2 * (81 - 81 * seta - 44 * eta)) * PI) / 48.0
        a[4] = ((-27312085.0 - 10287648 * chi22 - 10287648 * chi12 * sp1 + 10287648 * chi22 * seta
                 + 24 * (-1975055 + 857304 * chi12 - 994896 * chi1 * chi2 + 857304 * chi22) * eta
                 + 35371056 * eta2) * PI ** (4.0 / 3.0)) / 8.128512e6
        a[5] = (PI ** (5.0 / 3.0) * (chi2 * (-285197 * (-1 + seta) + 4 * (-91902 + 1579 * seta) * eta - 35632 * eta2)
                                     + chi1 * (285197 * sp1 - 4 * (91902 + 1579 * seta) * eta - 35632 * eta2)
                                     + 42840 * (-1.0 + 4 * eta) * PI)) / 32256.0
        a[6] = -(pi2 * (-336 * (-3248849057.0 + 2943675504 * chi12 - 3339284256 * chi1 * chi2 + 2943675504 * chi22) * eta2
                        - 324322727232 * eta3
                        - 7 * (-177520268561 + 107414046432 * chi22 + 107414046432 * chi12 * sp1
                               - 107414046432 * chi22 * seta
                               + 11087290368 * (chi1 + chi2 + chi1 * seta - chi2 * seta) * PI)
                        + 12 * eta * (-545384828789 - 176491177632 * chi1 * chi2 + 202603761360 * chi22
                                      + 77616 * chi12 * (2610335 + 995766 * seta) - 77287373856 * chi22 * seta
                                      + 5841690624 * (chi1 + chi2) * PI + 21384760320 * pi2))) / 6.0085960704e10
        a[7], a[8], a[9] = self.fit["rho1"], self.fit["rho2"], self.fit["rho3"]
        self.amp_ins = a                       # bracket(f) = 1 + sum_k a_k f^(k/3)
        g1, g2, g3 = self.fit["gamma1"], self.fit["gamma2"], self.fit["gamma3"]
        fRD, fDM = self.fRD, self.fDM
        if g2 <= 1.0:
            self.fmax = abs(fRD + fDM * (-1.0 + math.sqrt(1.0 - g2 * g2)) * g3 / g2)
        else:
            self.fmax = abs(fRD + fDM * (-1.0) * g3 / g2)
        # collocation: values at f1, f2 = (f1 + f3)/2, f3 and slopes at f1, f3
        f1, f3 = AMP_FJOIN_INS, self.fmax
        self.amp_f1, self.amp_f3 = f1, f3
        v1 = float(self.amp_ins_bracket(np.array([f1]))[0])
        d1 = float(self.amp_ins_bracket_derivative(np.array([f1]))[0])
        v3 = float(self.amp_mrd_bracket(np.array([f3]))[0])
        d2 = float(self.amp_mrd_bracket_derivative(np.array([f3]))[0])
        v2 = self.fit["v2"]
        # quartic through the five conditions, solved in the scaled variable u = (f - f1)/(f3 - f1) (the monomial system of
        # LAL's ComputeDeltasFromCollocation is the same polynomial, worse conditioned)
        L = f3 - f1
        A = np.array([[1, 0, 0, 0, 0],
                      [1, 0.5, 0.25, 0.125, 0.0625],
                      [1, 1, 1, 1, 1],
                      [0, 1, 0, 0, 0],
                      [0, 1, 2, 3, 4]], dtype=float)
        self.amp_int_poly = np.linalg.solve(A, np.array([v1, v2, v3, d1 * L, d2 * L]))

    def amp_ins_bracket(self, f):
        f13 = np.cbrt(f)
        return 1.0 + sum(self.amp_ins[k] * f13 ** k for k in range(2, 10))

    def amp_ins_bracket_derivative(self, f):
        f13 = np.cbrt(f)
        return sum(self.amp_ins[k] * (k / 3.0) * f13 ** k for k in range(2, 10)) / f

    def amp_mrd_bracket(self, f):
        g1, g2, g3 = self.fit["gamma1"], self.fit["gamma2"], self.fit["gamma3"]
        w = g3 * self.fDM
        x = f - self.fRD
        return np.exp(-x * g2 / w) * (w * g1) / (x * x + w * w)

    def amp_mrd_bracket_derivative(self, f):
        g2, g3 = self.fit["gamma2"], self.fit["gamma3"]
        w = g3 * self.fDM
        x = f - self.fRD
        return self.amp_mrd_bracket(f) * (-g2 / w - 2.0 * x / (x * x + w * w))

    def amplitude(self, f):
        """IMRPhenDAmplitude: dimensionless amplitude at geometric frequency f = M f_Hz (array)."""
        f = np.asarray(f, float)
        pre = self.amp0 * f ** (-7.0 / 6.0)
        u = (f - self.amp_f1) / (self.amp_f3 - self.amp_f1)
        p = self.amp_int_poly
        inter = p[0] + u * (p[1] + u * (p[2] + u * (p[3] + u * p[4])))
        out = np.where(f < self.amp_f1, self.amp_ins_bracket(f), np.where(f < self.amp_f3, inter, self.amp_mrd_bracket(f)))
        return pre * out

    # ---- phase (Khan et al. section VII; LAL init_phi_ins_prefactors, ComputeIMRPhenDPhaseConnectionCoefficients)
    def _setup_phase(self):
        eta, chi1, chi2 = self.eta, self.chi1, self.chi2
        M = self.M
        m1M, m2M = self.m1 / M, self.m2 / M
        d = (self.m1 - self.m2) / M
        pfaN = 3.0 / (128.0 * eta)
        v = np.zeros(8)
        vl = np.zeros(8)
        v[0] = 1.0
        v[2] = 5.0 * (74.3 / 8.4 + 11.0 * eta) / 9.0
        v[3] = -16.0 * PI
        v[4] = 5.0 * (3058.673 / 7.056 + 5429.0 / 7.0 * eta + 617.0 * eta * eta) / 72.0
        v[5] = 5.0 / 9.0 * (772.9 / 8.4 - 13.0 * eta) * PI
        vl[5] = 5.0 / 3.0 * (772.9 / 8.4 - 13.0 * eta) * PI
        v[6] = (11583.231236531 / 4.694215680 - 640.0 / 3.0 * PI * PI - 684.8 / 2.1 * GAMMA
                + eta * (-15737.765635 / 3.048192 + 225.5 / 1.2 * PI * PI)
                + eta * eta * 76.055 / 1.728 - eta ** 3 * 127.825 / 1.296 - 684.8 / 2.1 * math.log(4.0))
        vl[6] = -684.8 / 2.1
        v[7] = PI * (770.96675 / 2.54016 + 378.515 / 1.512 * eta - 740.45 / 7.56 * eta * eta)
        # aligned spins (XLALSimInspiralPNPhasing_F2 with S_i . L = chi_i)
        SL = m1M * m1M * chi1 + m2M * m2M * chi2
        dSigmaL = d * (m2M * chi2 - m1M * chi1)

        def ss3(q1, q2):
            out = (326.75 / 1.12 + 557.5 / 1.8 * eta) * eta * chi1 * chi2
            out += ((4703.5 / 8.4 + 2935.0 / 6.0 * m1M - 120.0 * m1M * m1M) * q1
                    + (-4108.25 / 6.72 - 108.5 / 1.2 * m1M + 125.5 / 3.6 * m1M * m1M)) * m1M * m1M * chi1 * chi1
            out += ((4703.5 / 8.4 + 2935.0 / 6.0 * m2M - 120.0 * m2M * m2M) * q2
                    + (-4108.25 / 6.72 - 108.5 / 1.2 * m2M + 125.5 / 3.6 * m2M * m2M)) * m2M * m2M * chi2 * chi2
            return out

        q1, q2 = self.qm1, self.qm2
        pn_sigma = eta * (721.0 / 48.0 * chi1 * chi2 - 247.0 / 48.0 * chi1 * chi2)
        pn_sigma += (720.0 * q1 - 1.0) / 96.0 * m1M * m1M * chi1 * chi1
        pn_sigma += (720.0 * q2 - 1.0) / 96.0 * m2M * m2M * chi2 * chi2
        pn_sigma -= (240.0 * q1 - 7.0) / 96.0 * m1M * m1M * chi1 * chi1
        pn_sigma -= (240.0 * q2 - 7.0) / 96.0 * m2M * m2M * chi2 * chi2
        pn_gamma = (554345.0 / 1134.0 + 110.0 * eta / 9.0) * SL + (13915.0 / 84.0 - 10.0 * eta / 3.0) * dSigmaL
        v[7] += ((-8980424995.0 / 762048.0 + 6586595.0 * eta / 756.0 - 305.0 * eta * eta / 36.0) * SL
                 - (170978035.0 / 48384.0 - 2876425.0 * eta / 672.0 - 4735.0 * eta * eta / 144.0) * dSigmaL)
        # IMRPhenomD was tuned without the 3PN spin-spin term: LAL subtracts the binary-black-hole value (q = 1); what is
        # left is the part proportional to the neutron stars' excess quadrupoles
        v[6] += PI * (3760.0 * SL + 1490.0 * dSigmaL) / 3.0 + ss3(q1, q2) - ss3(1.0, 1.0)
        v[5] += -pn_gamma
        vl[5] += -3.0 * pn_gamma
        v[4] += -10.0 * pn_sigma
        v[3] += 188.0 * SL / 3.0 + 25.0 * dSigmaL
        self.pn_v, self.pn_vl = pfaN * v, pfaN * vl
        fit = self.fit
        self.sig = [fit["sigma1"], 0.75 * fit["sigma2"], 0.6 * fit["sigma3"], 0.5 * fit["sigma4"]]
        # 3.5PN spin-squared / spin-cubed terms of NRTidalv2 (XLALSimInspiralGetHOSpinTerms), phase = 3/(128 eta) v^2 (SS + SSS)
        XA, XB = m1M, m2M
        XA2, XB2 = XA * XA, XB * XB
        c1s, c2s = chi1 * chi1, chi2 * chi2
        ss35 = -400.0 * PI * (q1 - 1.0) * c1s * XA2 - 400.0 * PI * (q2 - 1.0) * c2s * XB2
        sss35 = (10.0 * ((XA2 + 308.0 / 3.0 * XA) * chi1 + (XB2 - 89.0 / 3.0 * XB) * chi2) * (q1 - 1.0) * XA2 * c1s
                 + 10.0 * ((XB2 + 308.0 / 3.0 * XB) * chi2 + (XA2 - 89.0 / 3.0 * XA) * chi1) * (q2 - 1.0) * XB2 * c2s
                 - 440.0 * (self.oct1 - 1.0) * XA * XA2 * c1s * chi1 - 440.0 * (self.oct2 - 1.0) * XB * XB2 * c2s * chi2)
        self.ho_spin = pfaN * (ss35 + sss35)
        # connection coefficients (C1 continuity of the three phase pieces)
        eta_inv = 1.0 / eta
        fi, fm = PHI_FJOIN_INS, 0.5 * self.fRD
        self.phi_f1, self.phi_f2 = fi, fm
        self.C2int = self.dphi_ins(fi) - self.dphi_int(fi)
        self.C1int = self.phi_ins(fi) - self.phi_int(fi) - self.C2int * fi
        phi_int_m = self.phi_int(fm) + self.C1int + self.C2int * fm
        dphi_int_m = self.C2int + self.dphi_int(fm)
        self.C2mrd = dphi_int_m - self.dphi_mrd(fm)
        self.C1mrd = phi_int_m - self.phi_mrd(fm) - self.C2mrd * fm
        self.t0 = self.dphi_mrd(self.fmax)
        del eta_inv

    def phi_ins(self, f):
        v = (PI * f) ** (1.0 / 3.0)
        lv = np.log(v)
        out = -PI / 4.0
        for k in range(8):
            out = out + (self.pn_v[k] + self.pn_vl[k] * lv) * v ** (k - 5)
        s = self.sig
        f13 = np.cbrt(f)
        return out + (s[0] * f + s[1] * f * f13 + s[2] * f * f13 * f13 + s[3] * f * f) / self.eta

    def dphi_ins(self, f):
        v = (PI * f) ** (1.0 / 3.0)
        lv = np.log(v)
        out = 0.0
        for k in range(8):
            out = out + ((k - 5) * (self.pn_v[k] + self.pn_vl[k] * lv) + self.pn_vl[k]) * v ** (k - 5)
        out = out / (3.0 * f)
        fit = self.fit
        f13 = np.cbrt(f)
        return out + (fit["sigma1"] + fit["sigma2"] * f13 + fit["sigma3"] * f13 * f13 + fit["sigma4"] * f) / self.eta

    def phi_int(self, f):
        b1, b2, b3 = self.fit["beta1"], self.fit["beta2"], self.fit["beta3"]
        return (b1 * f - b3 / (3.0 * f ** 3) + b2 * np.log(f)) / self.eta

    def dphi_int(self, f):
        b1, b2, b3 = self.fit["beta1"], self.fit["beta2"], self.fit["beta3"]
        return (b1 + b3 / f ** 4 + b2 / f) / self.eta

    def phi_mrd(self, f):
        a = [self.fit[f"alpha{i}"] for i in range(1, 6)]
        return (a[0] * f - a[1] / f + 4.0 / 3.0 * a[2] * f ** 0.75
                + a[3] * np.arctan((f - a[4] * self.fRD) / self.fDM)) / self.eta

    def dphi_mrd(self, f):
        a = [self.fit[f"alpha{i}"] for i in range(1, 6)]
        y = (f - a[4] * self.fRD) / self.fDM
        return (a[0] + a[1] / (f * f) + a[2] * f ** (-0.25) + a[3] / (self.fDM * (1.0 + y * y))) / self.eta

    def phase(self, f):
        """IMRPhenDPhase at geometric frequency f (array)."""
        f = np.asarray(f, float)
        return np.where(f < self.phi_f1, self.phi_ins(f),
                        np.where(f < self.phi_f2, self.phi_int(f) + self.C1int + self.C2int * f,
                                 self.phi_mrd(f) + self.C1mrd + self.C2mrd * f))

    # ---- NRTidalv2 (Dietrich et al. 2019; LALSimNRTunedTides.c)
    def _setup_tides(self):
        XA, XB = self.m1 / self.M, self.m2 / self.M
        term1 = (1.0 + 12.0 * XB / XA) * XA ** 5 * self.lam1
        term2 = (1.0 + 12.0 * XA / XB) * XB ** 5 * self.lam2
        self.kappa2T = 3.0 / 13.0 * (term1 + term2)
        k = self.kappa2T
        q = self.m1 / self.m2
        num = 1.0 + 3.35411203e-2 * k + 4.31460284e-5 * k * k
        den = 1.0 + 7.54224145e-2 * k + 2.23626859e-4 * k * k
        self.f_merger_hz = 0.3586 / math.sqrt(q) * num / den / self.M_sec / (2.0 * PI)
        self.tidal_phase_pref = -k * 2.4375 / (XA * XB)

    def tidal_phase(self, f_hz):
        x = (PI * self.M_sec * f_hz) ** (2.0 / 3.0)
        xh = np.sqrt(x)
        num = 1.0 + x * (-12.615214237993088 + xh * 19.0537346970349 + x * (-21.166863146081035 + xh * 90.55082156324926
                                                                          + x * -60.25357801943598))
        den = 1.0 + x * (-15.11120782773667 + xh * 22.195327350624694 + x * 8.064109635305156)
        return self.tidal_phase_pref * x * x * xh * num / den

    def tidal_amplitude_bracket(self, f_hz):
        """Dietrich et al. 2019 eq. (24) divided by the leading-order point-particle amplitude amp0 f^(-7/6):
        -9 kappa x^5 (1 + 449/108 x + 22672/9 x^2.89) / (1 + 13477.8 x^4)."""
        x = (PI * self.M_sec * f_hz) ** (2.0 / 3.0)
        poly = (1.0 + 4.157407407407407 * x + 2519.111111111111 * x ** 2.89) / (1.0 + 13477.8073677 * x ** 4)
        return -9.0 * self.kappa2T * x ** 5 * poly

    def planck_taper(self, f_hz):
        """1 - PlanckTaper(f, f_merger, 1.2 f_merger)."""
        t1, t2 = self.f_merger_hz, 1.2 * self.f_merger_hz
        f = np.asarray(f_hz, float)
        out = np.ones_like(f)
        mid = (f > t1) & (f < t2)
        with np.errstate(over="ignore"):
            out[mid] = 1.0 - 1.0 / (np.exp((t2 - t1) / (f[mid] - t1) + (t2 - t1) / (f[mid] - t2)) + 1.0)
        out[f >= t2] = 0.0
        return out

    # ---- the waveform
    def h22(self, f_hz, distance_mpc, phase, f_ref):
        """Complex strain factor h(f) such that h_plus = (1 + cos^2 iota)/2 h and h_cross = -i cos(iota) h
        (SimInspiralChooseFDWaveform's polarisation step); zero where f <= 0 or M f > F_CUT."""
        f_hz = np.asarray(f_hz, float)
        out = np.zeros(f_hz.shape, dtype=complex)
        ok = (f_hz > 0) & (f_hz * self.M_sec <= F_CUT)
        fh = f_hz[ok]
        Mf = fh * self.M_sec
        amp0 = (2.0 * math.sqrt(5.0 / (64.0 * PI)) * self.M * MRSUN_SI * self.M * MTSUN_SI / (distance_mpc * 1e6 * PC_SI))
        Mf_ref = f_ref * self.M_sec
        phi_ref = float(self.phase(np.array([Mf_ref]))[0])
        phi = self.phase(Mf) - self.t0 * (Mf - Mf_ref) - (2.0 * phase + phi_ref)
        x = (PI * Mf) ** (2.0 / 3.0)
        phi = phi + self.ho_spin * x
        amp = self.amplitude(Mf)
        if self.tidal:
            phi = phi + self.tidal_phase(fh)
            amp = (amp + self.amp0 * Mf ** (-7.0 / 6.0) * self.tidal_amplitude_bracket(fh)) * self.planck_taper(fh)
        out[ok] = amp0 * amp * np.exp(-1j * phi)
        return out


# ------------------------------------------------------------------------------------------------------------------
# bilby's source-model conversions (bilby/gw/conversion.py, bilby/gw/source.py:lal_binary_neutron_star)
# ------------------------------------------------------------------------------------------------------------------
def component_masses(chirp_mass, mass_ratio):
    total = chirp_mass * (1.0 + mass_ratio) ** 1.2 / mass_ratio ** 0.6
    m1 = total / (1.0 + mass_ratio)
    return m1, m1 * mass_ratio


def polarizations(params, frequency_array, f_ref, f_min, f_max=np.inf, tidal=True):
    """(h_plus, h_cross) on ``frequency_array`` for a bilby-style parameter dict with aligned spins chi_1, chi_2."""
    if "mass_1" in params:
        m1, m2 = params["mass_1"], params["mass_2"]
    else:
        m1, m2 = component_masses(params["chirp_mass"], params["mass_ratio"])
    src = PhenomDNRTidalv2(m1, m2, params.get("chi_1", 0.0), params.get("chi_2", 0.0),
                           params.get("lambda_1", 0.0), params.get("lambda_2", 0.0), tidal=tidal)
    h = src.h22(frequency_array, params["luminosity_distance"], params.get("phase", 0.0), f_ref)
    band = (frequency_array >= f_min) & (frequency_array <= f_max)     # source.py: frequency_bounds
    h = h * band
    ci = math.cos(params["theta_jn"])
    return 0.5 * (1.0 + ci * ci) * h, -1j * ci * h


# ------------------------------------------------------------------------------------------------------------------
# detectors (bilby/gw/detector/geometry.py, detectors/*.interferometer, bilby_cython.geometry / time)
# ------------------------------------------------------------------------------------------------------------------
DETECTORS = {
    # name: latitude [deg], longitude [deg], elevation [m], x-arm azimuth [deg], y-arm azimuth [deg], x tilt, y tilt [rad]
    "H1": (46 + 27. / 60 + 18.528 / 3600, -(119 + 24. / 60 + 27.5657 / 3600), 142.554, 125.9994, 215.9994, -6.195e-4, 1.25e-5),
    "L1": (30 + 33. / 60 + 46.4196 / 3600, -(90 + 46. / 60 + 27.2654 / 3600), -6.574, 197.7165, 287.7165, -3.121e-4, -6.107e-4),
    "V1": (43 + 37. / 60 + 53.0921 / 3600, 10 + 30. / 60 + 16.1887 / 3600, 51.884, 70.5674, 160.5674, 0.0, 0.0),
}
_LEAP_GPS = [46828800, 78364801, 109900802, 173059203, 252028804, 315187205, 346723206, 393984007, 425520008, 457056009,
             504489610, 551750411, 599184012, 820108813, 914803214, 1025136015, 1119744016, 1167264017]


def detector_geometry(name):
    """(vertex[3] in metres, detector_tensor[3, 3]) from the site description."""
    lat, lon, elev, xaz, yaz, xt, yt = DETECTORS[name]
    lat, lon, xaz, yaz = (math.radians(v) for v in (lat, lon, xaz, yaz))
    a, b = 6378137.0, 6356752.314
    radius = a * a / math.sqrt(a * a * math.cos(lat) ** 2 + b * b * math.sin(lat) ** 2)
    vertex = np.array([(radius + elev) * math.cos(lat) * math.cos(lon), (radius + elev) * math.cos(lat) * math.sin(lon),
                       ((b / a) ** 2 * radius + elev) * math.sin(lat)])

    def arm(tilt, az):
        e_long = np.array([-math.sin(lon), math.cos(lon), 0.0])
        e_lat = np.array([-math.sin(lat) * math.cos(lon), -math.sin(lat) * math.sin(lon), math.cos(lat)])
        e_h = np.array([math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)])
        return math.cos(tilt) * math.cos(az) * e_long + math.cos(tilt) * math.sin(az) * e_lat + math.sin(tilt) * e_h

    x, y = arm(xt, xaz), arm(yt, yaz)
    return vertex, 0.5 * (np.outer(x, x) - np.outer(y, y))


def greenwich_mean_sidereal_time(gps):
    """bilby_cython.time.greenwich_mean_sidereal_time (LAL's XLALGreenwichMeanSiderealTime)."""
    leaps = sum(1 for g in _LEAP_GPS if gps >= g)
    julian_day = 2444244.5 + (math.floor(gps) - leaps) / 86400.0      # GPS epoch 1980-01-06 00:00:00 UTC
    t_hi = (julian_day - 2451545.0) / 36525.0
    t_lo = (gps % 1.0) / (36525.0 * 86400.0)
    t = t_hi + t_lo
    s = (-6.2e-6 * t + 0.093104) * t * t + 67310.54841
    s += 8640184.812866 * t_lo
    s += 3155760000.0 * t_lo
    s += 8640184.812866 * t_hi
    s += 3155760000.0 * t_hi
    return s * PI / 43200.0


def antenna_response(detector_tensor, ra, dec, gps, psi):
    gmst = math.fmod(greenwich_mean_sidereal_time(gps), 2.0 * PI)
    phi, theta = ra - gmst, PI / 2.0 - dec
    u = np.array([math.cos(phi) * math.cos(theta), math.cos(theta) * math.sin(phi), -math.sin(theta)])
    v = np.array([-math.sin(phi), math.cos(phi), 0.0])
    m = -u * math.sin(psi) - v * math.cos(psi)
    n = -u * math.cos(psi) + v * math.sin(psi)
    plus = np.outer(m, m) - np.outer(n, n)
    cross = np.outer(m, n) + np.outer(n, m)
    return float(np.sum(detector_tensor * plus)), float(np.sum(detector_tensor * cross))


def time_delay_from_geocenter(vertex, ra, dec, gps):
    gmst = math.fmod(greenwich_mean_sidereal_time(gps), 2.0 * PI)
    phi, theta = ra - gmst, PI / 2.0 - dec
    omega = np.array([math.sin(theta) * math.cos(phi), math.sin(theta) * math.sin(phi), math.cos(theta)])
    return float(np.dot(omega, -vertex) / C_SI)


# ------------------------------------------------------------------------------------------------------------------
# likelihood (bilby/gw/likelihood/base.py, bilby/gw/detector/interferometer.py, bilby/gw/utils.py)
# ------------------------------------------------------------------------------------------------------------------
def detector_strain(params, name, frequency_array, start_time, f_ref, f_min, f_max=np.inf, tidal=True):
    """Interferometer.get_detector_response without a calibration model."""
    hp, hc = polarizations(params, frequency_array, f_ref, f_min, f_max, tidal)
    vertex, tensor = detector_geometry(name)
    fp, fc = antenna_response(tensor, params["ra"], params["dec"], params["geocent_time"], params["psi"])
    signal = fp * hp + fc * hc
    dt = params["geocent_time"] - start_time + time_delay_from_geocenter(vertex, params["ra"], params["dec"], params["geocent_time"])
    return signal * np.exp(-1j * 2.0 * PI * dt * frequency_array)


def ln_i0(x):
    from scipy.special import ive
    return math.log(ive(0, x)) + abs(x)


def log_likelihood_ratio(params, ifos, f_ref, f_min_waveform, phase_marginalization=False, tidal=True, f_max_waveform=np.inf,
                         distance_marginalization=None, time_marginalization=None):
    """``ifos``: list of dicts with name, frequency_array, data, psd, mask, start_time, duration.
    ``distance_marginalization`` = (grid, ln(prior x step)): bilby's distance-marginalised likelihood
    (bilby/gw/likelihood/base.py: ``distance_marginalized_likelihood`` with the sum of ``_create_lookup_table`` evaluated
    directly instead of tabulated over (d_inner_h, h_inner_h) and interpolated)."""
    d_inner_h, opt = 0.0 + 0.0j, 0.0
    tc_array = 0.0
    for ifo in ifos:
        m = ifo["mask"]
        fa = ifo["frequency_array"]
        h = detector_strain(params, ifo["name"], fa, ifo["start_time"], f_ref, f_min_waveform, f_max_waveform, tidal)
        if time_marginalization is not None:
            # bilby calculate_snrs: d_inner_h_array = 4 / duration * fft(signal[0:-1] * conj(data)[0:-1] / psd[0:-1]); the data are
            # zero outside the detector's band (frequency_domain_strain * frequency_mask)
            prod = np.where(m, h * np.conj(ifo["data"]) / ifo["psd"], 0.0)[:-1]
            tc_array = tc_array + 4.0 / ifo["duration"] * np.fft.fft(prod)
        d_inner_h += 4.0 / ifo["duration"] * np.sum(np.conj(ifo["data"][m]) * h[m] / ifo["psd"][m])
        opt += (4.0 / ifo["duration"] * np.sum(np.conj(h[m]) * h[m] / ifo["psd"][m])).real
    if time_marginalization is not None:
        # bilby time_marginalized_likelihood: logsumexp(log_l_tc_array, b=time_prior_array), params["geocent_time"] = start_time
        from scipy.special import logsumexp
        logw = np.asarray(time_marginalization, dtype=np.float64)
        keep = np.isfinite(logw)
        if distance_marginalization is not None:
            # bilby: log_l_tc_array = distance_marginalized_likelihood(d_inner_h_tc_array, h_inner_h) -- the distance sum per time shift
            grid, dlogw = (np.asarray(a, dtype=np.float64) for a in distance_marginalization)
            dkeep = np.isfinite(dlogw)
            scale = float(params["luminosity_distance"]) / grid[dkeep]
            x = np.empty(int(keep.sum()))
            for n, v in enumerate(tc_array[keep]):
                if phase_marginalization:
                    from scipy.special import ive
                    arg = abs(v) * scale
                    xd = np.log(ive(0, arg)) + arg - opt * scale ** 2 / 2.0
                else:
                    xd = v.real * scale - opt * scale ** 2 / 2.0
                x[n] = logsumexp(xd + dlogw[dkeep])
        elif phase_marginalization:
            x = np.array([ln_i0(abs(v)) for v in tc_array[keep]]) - opt / 2.0
        else:
            x = tc_array[keep].real - opt / 2.0
        return float(logsumexp(x + logw[keep]))
    if distance_marginalization is not None:
        from scipy.special import logsumexp
        grid, logw = (np.asarray(a, dtype=np.float64) for a in distance_marginalization)
        keep = np.isfinite(logw)
        scale = float(params["luminosity_distance"]) / grid[keep]      # <d|h> ~ 1/d, <h|h> ~ 1/d^2
        if phase_marginalization:
            x = np.array([ln_i0(abs(d_inner_h) * s) for s in scale]) - opt * scale ** 2 / 2.0
        else:
            x = d_inner_h.real * scale - opt * scale ** 2 / 2.0
        return float(logsumexp(x + logw[keep]))
    if phase_marginalization:
        return ln_i0(abs(d_inner_h)) - opt / 2.0
    return d_inner_h.real - opt / 2.0


def noise_log_likelihood(ifos):
    out = 0.0
    for ifo in ifos:
        m = ifo["mask"]
        out -= (4.0 / ifo["duration"] * np.sum(np.abs(ifo["data"][m]) ** 2 / ifo["psd"][m])) / 2.0
    return float(out)
