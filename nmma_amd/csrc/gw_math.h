// gw_math.h -- scalar fp64 building blocks of the gravitational-wave leg (BASELINE config 5):
// frequency-domain IMRPhenomD_NRTidalv2 strain, detector projection and the per-sample set-up that folds every
// frequency-independent quantity into one record (GwSource).  Shared by the HIP kernels (gw_kernels.hip) and by
// tests/hostcheck (host build of this very source, checked on the CPU against oracle/gw_waveform_oracle.py).
//
// What this replaces in the reference: nmma/gw/gw_likelihood.py:97-247 hands the work to
// bilby.gw.likelihood.GravitationalWaveTransient (:185-203), which calls lalsimulation's IMRPhenomD_NRTidalv2 through
// bilby.gw.source.lal_binary_neutron_star and projects it with Interferometer.get_detector_response.  Both are third-party
// and absent from the build image: the formulas below restate the published algorithms (Khan et al. 2016 = arXiv:1508.07253
// section V-VII + Table V; Dietrich et al. 2019 = arXiv:1905.06011 eqs. 17-24; LALSimInspiralPNCoefficients.c;
// bilby_cython geometry / time) -- PARITY AGAINST LALSIMULATION IS UNPINNED, see oracle/gw_waveform_oracle.py.
//
// Formulation for the GPU: with f in Hz, f13 = cbrt(f), every power of the PN expansion parameter v = (pi M f)^(1/3) is
// a per-sample constant times a power of f13, so the per-(bin, sample) work is polynomial arithmetic on a small per-bin basis
// (f, f13, 1/f13, ln f13, f^(-7/6)) that is tabulated once; phases are carried in units of pi so the final sincos is
// a sincospi with an exact range reduction.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GW_HD __host__ __device__ __forceinline__
#define GW_HD_NOINLINE inline __host__ __device__ __noinline__
#else
#define GW_HD inline
#define GW_HD_NOINLINE inline
#endif

namespace nmma {
namespace gw {

constexpr double kPi = 3.141592653589793;
constexpr double kGamma = 0.5772156649015329;
constexpr double kMTSun = 4.925490947641266978e-06;   // G M_sun / c^3 [s]
constexpr double kMRSun = 1.476625038050124729e+03;   // G M_sun / c^2 [m]
constexpr double kParsec = 3.085677581491367e16;
constexpr double kC = 299792458.0;
constexpr double kFCut = 0.2;          // M f above which IMRPhenomD is zero
constexpr double kAmpFJoin = 0.014;    // inspiral -> intermediate amplitude (M f)
constexpr double kPhiFJoin = 0.018;    // inspiral -> intermediate phase (M f)
constexpr int kMaxIfo = 4;

// Khan et al. 2016, Table V (rows in the order rho1-3, v2, gamma1-3, sigma1-4, beta1-3, alpha1-5).
enum Fit { RHO1, RHO2, RHO3, V2, GAMMA1, GAMMA2, GAMMA3, SIGMA1, SIGMA2, SIGMA3, SIGMA4, BETA1, BETA2, BETA3,
           ALPHA1, ALPHA2, ALPHA3, ALPHA4, ALPHA5, N_FIT };

GW_HD_NOINLINE double table_fit(int row, double eta, double chi_pn) {
    static const double T[N_FIT][11] = {
        {3931.8979897196696, -17395.758706812805, 3132.375545898835, 343965.86092361377, -1.2162565819981997e6,
         -70698.00600428853, 1.383907177859705e6, -3.9662761890979446e6, -60017.52423652596, 803515.1181825735,
         -2.091710365941658e6},
        {-40105.47653771657, 112253.0169706701, 23561.696065836168, -3.476180699403351e6, 1.137593670849482e7,
         754313.1127166454, -1.308476044625268e7, 3.6444584853928134e7, 596226.612472288, -7.4277901143564405e6,
         1.8928977514040343e7},
        {83208.35471266537, -191237.7264145924, -210916.2454782992, 8.71797508352568e6, -2.6914942420669552e7,
         -1.9889806527362722e6, 3.0888029960154563e7, -8.390870279256162e7, -1.4535031953446497e6, 1.7063528990822166e7,
         -4.2748659731120914e7},
        {0.8149838730507785, 2.5747553517454658, 1.1610198035496786, -2.3627771785551537, 6.771038707057573,
         0.7570782938606834, -2.7256896890432474, 7.1140380397149965, 0.1766934149293479, -0.7978690983168183,
         2.1162391502005153},
        {0.006927402739328343, 0.03020474290328911, 0.006308024337706171, -0.12074130661131138, 0.26271598905781324,
         0.0034151773647198794, -0.10779338611188374, 0.27098966966891747, 0.0007374185938559283, -0.02749621038376281,
         0.0733150789135702},
        {1.010344404799477, 0.0008993122007234548, 0.283949116804459, -4.049752962958005, 13.207828172665366,
         0.10396278486805426, -7.025059158961947, 24.784892370130475, 0.03093202475605892, -2.6924023896851663,
         9.609374464684983},
        {1.3081615607036106, -0.005537729694807678, -0.06782917938621007, -0.6689834970767117, 3.403147966134083,
         -0.05296577374411866, -0.9923793203111362, 4.820681208409587, -0.006134139870393713, -0.38429253308696365,
         1.7561754421985984},
        {2096.551999295543, 1463.7493168261553, 1312.5493286098522, 18307.330017082117, -43534.1440746107,
         -833.2889543511114, 32047.31997183187, -108609.45037520859, 452.25136398112204, 8353.439546391714,
         -44531.3250037322},
        {-10114.056472621156, -44631.01109458185, -6541.308761668722, -266959.23419307504, 686328.3229317984,
         3405.6372187679685, -437507.7208209015, 1.6318171307344697e6, -7462.648563007646, -114585.25177153319,
         674402.4689098676},
        {22933.658273436497, 230960.00814979506, 14961.083974183695, 1.1940181342318142e6, -3.1042239693052764e6,
         -3038.166617199259, 1.8720322849093592e6, -7.309145012085539e6, 42738.22871475411, 467502.018616601,
         -3.064853498512499e6},
        {-14621.71522218357, -377812.8579387104, -9608.682631509726, -1.7108925257214056e6, 4.332924601416521e6,
         -22366.683262266528, -2.5019716386377467e6, 1.0274495902259542e7, -85360.30079034246, -570025.3441737515,
         4.396844346849777e6},
        {97.89747327985583, -42.659730877489224, 153.48421037904913, -1417.0620760768954, 2752.8614143665027,
         138.7406469558649, -1433.6585075135881, 2857.7418952430758, 41.025109467376126, -423.680737974639,
         850.3594335657173},
        {-3.282701958759534, -9.051384468245866, -12.415449742258042, 55.4716447709787, -106.05109938966335,
         -11.953044553690658, 76.80704618365418, -155.33172948098394, -3.4129261592393263, 25.572377569952536,
         -54.408036707740465},
        {-0.000025156429818799565, 0.000019750256942201327, -0.000018370671469295915, 0.000021886317041311973,
         0.00008250240316860033, 7.157371250566708e-6, -0.000055780000112270685, 0.00019142082884072178,
         5.447166261464217e-6, -0.00003220610095021982, 0.00007974016714984341},
        {43.31514709695348, 638.6332679188081, -32.85768747216059, 2415.8938269370315, -5766.875169379177,
         -61.85459307173841, 2953.967762459948, -8986.29057591497, -21.571435779762044, 981.2158224673428,
         -3239.5664895930286},
        {-0.07020209449091723, -0.16269798450687084, -0.1872514685185499, 1.138313650449945, -2.8334196304430046,
         -0.17137955686840617, 1.7197549338119527, -4.539717148261272, -0.049983437357548705, 0.6062072055948309,
         -1.682769616644546},
        {9.5988072383479, -397.05438595557433, 16.202126189517813, -1574.8286986717037, 3600.3410843831093,
         27.092429659075467, -1786.482357315139, 5152.919378666511, 11.175710130033895, -577.7999423177481,
         1808.730762932043},
        {-0.02989487384493607, 1.4022106448583738, -0.07356049468633846, 0.8337006542278661, 0.2240008282397391,
         -0.055202870001177226, 0.5667186343606578, 0.7186931973380503, -0.015507437354325743, 0.15750322779277187,
         0.21076815715176228},
        {0.9974408278363099, -0.007884449714907203, -0.059046901195591035, 1.3958712396764088, -4.516631601676276,
         -0.05585343136869692, 1.7516580039343603, -5.990208965347804, -0.017945336522161195, 0.5965097794825992,
         -2.0608879367971804},
    };
    const double* c = T[row];
    const double xi = chi_pn - 1.0;
    return c[0] + c[1] * eta + xi * (c[2] + c[3] * eta + c[4] * eta * eta) + xi * xi * (c[5] + c[6] * eta + c[7] * eta * eta) +
           xi * xi * xi * (c[8] + c[9] * eta + c[10] * eta * eta);
}

// ---- final state (LALSimIMRPhenomD_internals.c: FinalSpin0815, EradRational0815) and the l = m = 2, n = 0 Kerr mode
//      (Berti, Cardoso & Will 2006 Table VIII in place of LAL's interpolation table: stated deviation)
GW_HD double final_spin_0815(double eta, double chi1, double chi2) {
    const double seta = sqrt(fmax(1.0 - 4.0 * eta, 0.0));
    const double m1 = 0.5 * (1.0 + seta), m2 = 0.5 * (1.0 - seta);
    const double s = m1 * m1 * chi1 + m2 * m2 * chi2;
    const double eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta2 * eta2;
    const double s2 = s * s, s3 = s2 * s, s4 = s2 * s2;
    return 3.4641016151377544 * eta - 4.399247300629289 * eta2 + 9.397292189321194 * eta3 - 13.180949901606242 * eta4 +
           (1 - 0.0850917821418767 * eta - 5.837029316602263 * eta2) * s + (0.1014665242971878 * eta - 2.0967746996832157 * eta2) * s2 +
           (-1.3546806617824356 * eta + 4.108962025369336 * eta2) * s3 + (-0.8676969352555539 * eta + 2.064046835273906 * eta2) * s4;
}

GW_HD double erad_rational_0815(double eta, double chi1, double chi2) {
    const double seta = sqrt(fmax(1.0 - 4.0 * eta, 0.0));
    const double m1 = 0.5 * (1.0 + seta), m2 = 0.5 * (1.0 - seta);
    const double s = (m1 * m1 * chi1 + m2 * m2 * chi2) / (m1 * m1 + m2 * m2);
    const double eta2 = eta * eta, eta3 = eta2 * eta;
    return ((0.055974469826360077 * eta + 0.5809510763115132 * eta2 - 0.9606726679372312 * eta3 + 3.352411249771192 * eta3 * eta) *
            (1. + (-0.0030302335878845507 - 2.0066110851351073 * eta + 7.7050567802399215 * eta2) * s)) /
           (1. + (-0.6714403054720589 - 1.4756929437702908 * eta + 7.304676214885011 * eta2) * s);
}

// ---- universal relations of NRTidalv2 (LALSimUniversalRelations.c; Yagi & Yunes 2017)
GW_HD double quadrupole_from_lambda(double lam) {
    if (lam < 1.0) return 1.0 + lam * (0.427688866723244 + lam * (-0.324336526985068 + lam * 0.1107439432180572));
    const double ll = log(lam);
    return exp(0.1940 + 0.09163 * ll + 0.04812 * ll * ll - 4.283e-3 * ll * ll * ll + 1.245e-4 * ll * ll * ll * ll);
}
GW_HD double octupole_from_quadrupole(double qm) {
    const double lq = log(qm);
    return exp(0.003131 + 2.071 * lq - 0.7152 * lq * lq + 0.2458 * lq * lq * lq - 0.03309 * lq * lq * lq * lq);
}

// One parameter vector as bilby's source model sees it (bilby/gw/source.py: lal_binary_neutron_star, aligned spins).
struct GwParams {
    double mass_1, mass_2;        // detector-frame solar masses
    double chi_1, chi_2, lambda_1, lambda_2;
    double luminosity_distance;   // Mpc
    double theta_jn, phase, ra, dec, psi, geocent_time;
};

// Everything frequency-independent of one parameter vector, in the units the bin loop wants: f in Hz, f13 = cbrt(f),
// phases in units of pi.  Plain doubles only: the kernels read it through the scalar cache.
struct GwSource {
    double valid;                 // 0: a non-finite or unphysical input -> the sample gets the floor
    double distance;              // Mpc: the luminosity distance the amplitude was scaled with (distance marginalisation rescales from it)
    double jitter;                // s: bilby's time_jitter of the row (time marginalisation with jitter_time), else 0
    // ---- amplitude: A(f) = amp_scale * f^(-7/6) * (bracket(f) + tidal bracket) * taper
    double amp_scale;
    double fa1, fa3;              // Hz: inspiral | intermediate | merger-ringdown boundaries of the amplitude
    double f_cut;                 // Hz: zero above
    double ai[8];                 // inspiral bracket 1 + sum_k ai[k-2] f13^k, k = 2..9
    double iu_scale;              // intermediate: u = (f - fa1) * iu_scale, quartic ip[0..4] in u
    double ip[5];
    double fRD, mw, mg2w, mg1w;   // merger-ringdown bracket exp(-(f - fRD) mg2w) * mg1w / ((f - fRD)^2 + mw^2)
    // ---- phase / pi
    double fp1, fp2;              // Hz: inspiral | intermediate | merger-ringdown boundaries of the phase
    double pc0;                   // inspiral: constant
    double pcm5, pcm3, pcm2, pcm1, pc1, pc2;     // powers f13^-5, ^-3, ^-2, ^-1, ^1, ^2
    double pl5, pl6;              // (pl5 + pl6 f13) * ln f13
    double ho2;                   // every region: + ho2 f13^2 (3.5PN spin-squared / spin-cubed terms of NRTidalv2)
    double ps1, ps2, ps3, ps4;    // f, f f13, f f13^2, f^2
    double ic0, ic1, icm3, icl;   // intermediate: ic0 + ic1 f + icm3 f^-3 + icl ln f13
    double mc0, mc1, mcm1, mc34, mcat, mfa5, minv_fdm;  // merger-ringdown: mc0 + mc1 f + mcm1 / f + mc34 f^(3/4) + mcat atan((f - mfa5) minv_fdm)
    // ---- NRTidalv2
    double has_tides;
    double xa;                    // x^(1/2) = xa * f13 with x = (pi M f)^(2/3)
    double tphase;                // tidal phase / pi = tphase * xh^5 * N(xh) / D(xh)
    double tamp;                  // tidal amplitude bracket = tamp * x^5 * (1 + n1 x + n289 x^2.89) / (1 + d x^4)
    double xa578;                 // xa^5.78: x^2.89 = xa578 * f13^5.78 (the second factor is tabulated per bin)
    double ft1, ft2;              // Hz: Planck taper between the merger frequency and 1.2 x it
    // ---- projection onto each detector: h_ifo = (k_re + i k_im) * A * exp(-i pi (P(f) + 2 f dt))
    double k_re[kMaxIfo], k_im[kMaxIfo], k_sq[kMaxIfo], dt[kMaxIfo];
    // exp(-2 pi i stride dt) for the frequency stride of the bin loop: the linear part of the phase advances by a complex
    // multiplication from one of a lane's bins to its next (gw_logl_kernel)
    double rs_re[kMaxIfo], rs_im[kMaxIfo];
};

struct GwDetector {
    double tensor[9];
    double vertex[3];
};

// per-bin basis
struct GwBin {
    double f, f13, inv13, lnf13, fm76;
    double p578;      // f13^5.78 = f^(2.89 * 2/3): x^2.89 of the tidal amplitude is a per-sample constant times this
};
GW_HD GwBin make_bin(double f) {
    GwBin b;
    b.f = f;
    b.f13 = cbrt(f);
    b.inv13 = 1.0 / b.f13;
    b.lnf13 = log(f) / 3.0;
    b.fm76 = 1.0 / (f * sqrt(b.f13));
    b.p578 = exp(5.78 * b.lnf13);
    return b;
}

namespace detail {
// amplitude brackets in geometric frequency (set-up only)
struct AmpSetup {
    double a[10];         // a[k], k = 2..9
    double g1, g2, g3, fRD, fDM;
};
GW_HD double amp_ins_bracket(const AmpSetup& s, double f) {
    const double f13 = cbrt(f);
    double p = f13 * f13, out = 1.0;
    for (int k = 2; k < 10; ++k) { out += s.a[k] * p; p *= f13; }
    return out;
}
GW_HD double amp_ins_bracket_d(const AmpSetup& s, double f) {
    const double f13 = cbrt(f);
    double p = f13 * f13, out = 0.0;
    for (int k = 2; k < 10; ++k) { out += s.a[k] * (k / 3.0) * p; p *= f13; }
    return out / f;
}
GW_HD double amp_mrd_bracket(const AmpSetup& s, double f) {
    const double w = s.g3 * s.fDM, x = f - s.fRD;
    return exp(-x * s.g2 / w) * (w * s.g1) / (x * x + w * w);
}
GW_HD double amp_mrd_bracket_d(const AmpSetup& s, double f) {
    const double w = s.g3 * s.fDM, x = f - s.fRD;
    return amp_mrd_bracket(s, f) * (-s.g2 / w - 2.0 * x / (x * x + w * w));
}
struct PhaseSetup {
    double v[8], vl[8];   // pfaN-scaled PN coefficients
    double eta, s1, s2, s3, s4, b1, b2, b3, al[5], fRD, fDM;
};
GW_HD double phi_ins(const PhaseSetup& p, double f) {
    const double v = cbrt(kPi * f), lv = log(v);
    double out = -kPi / 4.0, pw = 1.0 / (v * v * v * v * v);
    for (int k = 0; k < 8; ++k) { out += (p.v[k] + p.vl[k] * lv) * pw; pw *= v; }
    const double f13 = cbrt(f);
    return out + (p.s1 * f + 0.75 * p.s2 * f * f13 + 0.6 * p.s3 * f * f13 * f13 + 0.5 * p.s4 * f * f) / p.eta;
}
GW_HD double dphi_ins(const PhaseSetup& p, double f) {
    const double v = cbrt(kPi * f), lv = log(v);
    double out = 0.0, pw = 1.0 / (v * v * v * v * v);
    for (int k = 0; k < 8; ++k) { out += ((k - 5) * (p.v[k] + p.vl[k] * lv) + p.vl[k]) * pw; pw *= v; }
    const double f13 = cbrt(f);
    return out / (3.0 * f) + (p.s1 + p.s2 * f13 + p.s3 * f13 * f13 + p.s4 * f) / p.eta;
}
GW_HD double phi_int(const PhaseSetup& p, double f) { return (p.b1 * f - p.b3 / (3.0 * f * f * f) + p.b2 * log(f)) / p.eta; }
GW_HD double dphi_int(const PhaseSetup& p, double f) { return (p.b1 + p.b3 / (f * f * f * f) + p.b2 / f) / p.eta; }
GW_HD double phi_mrd(const PhaseSetup& p, double f) {
    return (p.al[0] * f - p.al[1] / f + 4.0 / 3.0 * p.al[2] * sqrt(f * sqrt(f)) + p.al[3] * atan((f - p.al[4] * p.fRD) / p.fDM)) / p.eta;
}
GW_HD double dphi_mrd(const PhaseSetup& p, double f) {
    const double y = (f - p.al[4] * p.fRD) / p.fDM;
    return (p.al[0] + p.al[1] / (f * f) + p.al[2] / sqrt(sqrt(f)) + p.al[3] / (p.fDM * (1.0 + y * y))) / p.eta;
}
GW_HD double phase_geometric(const PhaseSetup& p, double f, double f1, double f2, double C1i, double C2i, double C1m, double C2m) {
    if (f < f1) return phi_ins(p, f);
    if (f < f2) return phi_int(p, f) + C1i + C2i * f;
    return phi_mrd(p, f) + C1m + C2m * f;
}
}  // namespace detail

// Source model set-up: everything of IMRPhenomD_NRTidalv2 that does not depend on frequency
// (LAL: XLALSimIMRPhenomDNRTidal -> IMRPhenomDSetupAmpAndPhaseCoefficients, ComputeIMRPhenomDAmplitudeCoefficients,
//  ComputeIMRPhenomDPhaseCoefficients, ComputeIMRPhenDPhaseConnectionCoefficients, XLALSimNRTunedTides*).
// tidal = true: IMRPhenomD_NRTidalv2 (tidal phase and amplitude, spin-induced multipoles from the universal relations, Planck
// taper at the merger frequency -- applied whatever the deformabilities are, as LAL does); false: plain IMRPhenomD.
GW_HD_NOINLINE void setup_source(const GwParams& q, double f_ref, bool tidal, GwSource& S) {
    double m1 = q.mass_1, m2 = q.mass_2, chi1 = q.chi_1, chi2 = q.chi_2, lam1 = q.lambda_1, lam2 = q.lambda_2;
    if (m1 < m2) {       // LAL: body 1 is the heavier one
        double t;
        t = m1; m1 = m2; m2 = t;
        t = chi1; chi1 = chi2; chi2 = t;
        t = lam1; lam1 = lam2; lam2 = t;
    }
    const bool ok = m1 > 0 && m2 > 0 && fabs(chi1) <= 1.0 && fabs(chi2) <= 1.0 && lam1 >= 0 && lam2 >= 0 && q.luminosity_distance > 0 &&
                    isfinite(m1) && isfinite(m2) && isfinite(lam1) && isfinite(lam2) && isfinite(q.luminosity_distance) &&
                    isfinite(q.theta_jn) && isfinite(q.phase) && isfinite(q.ra) && isfinite(q.dec) && isfinite(q.psi) && isfinite(q.geocent_time);
    S.valid = ok ? 1.0 : 0.0;
    if (!ok) return;
    const double M = m1 + m2, M_sec = M * kMTSun;
    double eta = m1 * m2 / (M * M);
    if (eta > 0.25) eta = 0.25;
    const double seta = sqrt(fmax(1.0 - 4.0 * eta, 0.0));
    const double chi_pn = 0.5 * (chi1 + chi2) * (1.0 - eta * 76.0 / 113.0) + seta * 0.5 * (chi1 - chi2);
    double fit[N_FIT];
    for (int r = 0; r < N_FIT; ++r) fit[r] = table_fit(r, eta, chi_pn);
    const double fs = final_spin_0815(eta, chi1, chi2), erad = erad_rational_0815(eta, chi1, chi2);
    const double j = fmin(fmax(fs, -0.999), 0.999);
    const double fr = (1.5251 - 1.1568 * pow(1.0 - j, 0.1292)) / (2.0 * kPi);
    const double qq = 0.7000 + 1.4187 * pow(1.0 - j, -0.4990);
    const double fRD = fr / (1.0 - erad), fDM = fr / (2.0 * qq) / (1.0 - erad);
    const double qm1 = tidal ? quadrupole_from_lambda(lam1) : 1.0, qm2 = tidal ? quadrupole_from_lambda(lam2) : 1.0;
    const double oct1 = tidal ? octupole_from_quadrupole(qm1) : 1.0, oct2 = tidal ? octupole_from_quadrupole(qm2) : 1.0;
    const double pi2 = kPi * kPi;

    // ---------------- amplitude
    detail::AmpSetup A;
    {
        const double eta2 = eta * eta, eta3 = eta2 * eta, chi12 = chi1 * chi1, chi22 = chi2 * chi2, sp1 = 1.0 + seta;
        A.a[0] = A.a[1] = 0.0;
        A.a[2] = ((-969 + 1804 * eta) * pow(kPi, 2.0 / 3.0)) / 672.0;
        A.a[3] = ((chi1 * (81 * sp1 - 44 * eta) + chi2 * (81 - 81 * seta - 44 * eta)) * kPi) / 48.0;
        A.a[4] = ((-27312085.0 - 10287648 * chi22 - 10287648 * chi12 * sp1 + 10287648 * chi22 * seta +
                   24 * (-1975055 + 857304 * chi12 - 994896 * chi1 * chi2 + 857304 * chi22) * eta + 35371056 * eta2) *
                  pow(kPi, 4.0 / 3.0)) / 8.128512e6;
        A.a[5] = (pow(kPi, 5.0 / 3.0) * (chi2 * (-285197 * (-1 + seta) + 4 * (-91902 + 1579 * seta) * eta - 35632 * eta2) +
                                         chi1 * (285197 * sp1 - 4 * (91902 + 1579 * seta) * eta - 35632 * eta2) +
                                         42840 * (-1.0 + 4 * eta) * kPi)) / 32256.0;
        A.a[6] = -(pi2 * (-336 * (-3248849057.0 + 2943675504 * chi12 - 3339284256 * chi1 * chi2 + 2943675504 * chi22) * eta2 -
                          324322727232 * eta3 -
                          7 * (-177520268561 + 107414046432 * chi22 + 107414046432 * chi12 * sp1 - 107414046432 * chi22 * seta +
                               11087290368 * (chi1 + chi2 + chi1 * seta - chi2 * seta) * kPi) +
                          12 * eta * (-545384828789 - 176491177632 * chi1 * chi2 + 202603761360 * chi22 +
                                      77616 * chi12 * (2610335 + 995766 * seta) - 77287373856 * chi22 * seta +
                                      5841690624 * (chi1 + chi2) * kPi + 21384760320 * pi2))) / 6.0085960704e10;
        A.a[7] = fit[RHO1]; A.a[8] = fit[RHO2]; A.a[9] = fit[RHO3];
        A.g1 = fit[GAMMA1]; A.g2 = fit[GAMMA2]; A.g3 = fit[GAMMA3]; A.fRD = fRD; A.fDM = fDM;
    }
    double fmax_g;
    if (A.g2 <= 1.0) fmax_g = fabs(fRD + fDM * (-1.0 + sqrt(1.0 - A.g2 * A.g2)) * A.g3 / A.g2);
    else fmax_g = fabs(fRD + fDM * (-1.0) * A.g3 / A.g2);
    {
        // quartic through (f1, v1, d1), (f2, v2), (f3, v3, d2) in u = (f - f1)/(f3 - f1): cubic Hermite + c u^2 (1-u)^2
        const double f1 = kAmpFJoin, f3 = fmax_g, L = f3 - f1;
        const double v1 = detail::amp_ins_bracket(A, f1), d1 = detail::amp_ins_bracket_d(A, f1) * L;
        const double v3 = detail::amp_mrd_bracket(A, f3), d2 = detail::amp_mrd_bracket_d(A, f3) * L;
        const double c = 16.0 * (fit[V2] - 0.5 * (v1 + v3) - (d1 - d2) / 8.0);
        // H(u) = v1 + d1 u + (-3 v1 - 2 d1 + 3 v3 - d2) u^2 + (2 v1 + d1 - 2 v3 + d2) u^3;  c u^2 (1 - 2u + u^2)
        S.ip[0] = v1;
        S.ip[1] = d1;
        S.ip[2] = -3.0 * v1 - 2.0 * d1 + 3.0 * v3 - d2 + c;
        S.ip[3] = 2.0 * v1 + d1 - 2.0 * v3 + d2 - 2.0 * c;
        S.ip[4] = c;
        S.fa1 = f1 / M_sec; S.fa3 = f3 / M_sec;
        S.iu_scale = M_sec / L;
    }
    S.f_cut = kFCut / M_sec;
    {
        const double m13 = cbrt(M_sec);
        double p = m13 * m13;
        for (int k = 2; k < 10; ++k) { S.ai[k - 2] = A.a[k] * p; p *= m13; }
    }
    S.fRD = fRD / M_sec;
    S.mw = A.g3 * fDM / M_sec;
    S.mg2w = A.g2 / S.mw;
    S.mg1w = S.mw * A.g1 / M_sec;
    const double amp0_strain = 2.0 * sqrt(5.0 / (64.0 * kPi)) * M * kMRSun * M * kMTSun / (q.luminosity_distance * 1e6 * kParsec);
    const double amp0 = sqrt(2.0 * eta / 3.0) * pow(kPi, -1.0 / 6.0);
    S.amp_scale = amp0_strain * amp0 * pow(M_sec, -7.0 / 6.0);

    // ---------------- phase
    detail::PhaseSetup P;
    const double m1M = m1 / M, m2M = m2 / M, dm = (m1 - m2) / M, pfaN = 3.0 / (128.0 * eta);
    {
        double v[8] = {0}, vl[8] = {0};
        v[0] = 1.0;
        v[2] = 5.0 * (74.3 / 8.4 + 11.0 * eta) / 9.0;
        v[3] = -16.0 * kPi;
        v[4] = 5.0 * (3058.673 / 7.056 + 5429.0 / 7.0 * eta + 617.0 * eta * eta) / 72.0;
        v[5] = 5.0 / 9.0 * (772.9 / 8.4 - 13.0 * eta) * kPi;
        vl[5] = 5.0 / 3.0 * (772.9 / 8.4 - 13.0 * eta) * kPi;
        v[6] = (11583.231236531 / 4.694215680 - 640.0 / 3.0 * pi2 - 684.8 / 2.1 * kGamma) +
               eta * (-15737.765635 / 3.048192 + 225.5 / 1.2 * pi2) + eta * eta * 76.055 / 1.728 - eta * eta * eta * 127.825 / 1.296 -
               684.8 / 2.1 * log(4.0);
        vl[6] = -684.8 / 2.1;
        v[7] = kPi * (770.96675 / 2.54016 + 378.515 / 1.512 * eta - 740.45 / 7.56 * eta * eta);
        const double SL = m1M * m1M * chi1 + m2M * m2M * chi2, dSigmaL = dm * (m2M * chi2 - m1M * chi1);
        const double c1s = chi1 * chi1, c2s = chi2 * chi2;
        // 3PN spin-spin: only the part proportional to the excess quadrupoles survives LAL's subtraction of the BBH value
        const double ss3_excess = (4703.5 / 8.4 + 2935.0 / 6.0 * m1M - 120.0 * m1M * m1M) * (qm1 - 1.0) * m1M * m1M * c1s +
                                  (4703.5 / 8.4 + 2935.0 / 6.0 * m2M - 120.0 * m2M * m2M) * (qm2 - 1.0) * m2M * m2M * c2s;
        double pn_sigma = eta * (721.0 / 48.0 * chi1 * chi2 - 247.0 / 48.0 * chi1 * chi2);
        pn_sigma += (720.0 * qm1 - 1.0) / 96.0 * m1M * m1M * c1s;
        pn_sigma += (720.0 * qm2 - 1.0) / 96.0 * m2M * m2M * c2s;
        pn_sigma -= (240.0 * qm1 - 7.0) / 96.0 * m1M * m1M * c1s;
        pn_sigma -= (240.0 * qm2 - 7.0) / 96.0 * m2M * m2M * c2s;
        const double pn_gamma = (554345.0 / 1134.0 + 110.0 * eta / 9.0) * SL + (13915.0 / 84.0 - 10.0 * eta / 3.0) * dSigmaL;
        v[7] += (-8980424995.0 / 762048.0 + 6586595.0 * eta / 756.0 - 305.0 * eta * eta / 36.0) * SL -
                (170978035.0 / 48384.0 - 2876425.0 * eta / 672.0 - 4735.0 * eta * eta / 144.0) * dSigmaL;
        v[6] += kPi * (3760.0 * SL + 1490.0 * dSigmaL) / 3.0 + ss3_excess;
        v[5] += -pn_gamma;
        vl[5] += -3.0 * pn_gamma;
        v[4] += -10.0 * pn_sigma;
        v[3] += 188.0 * SL / 3.0 + 25.0 * dSigmaL;
        for (int k = 0; k < 8; ++k) { P.v[k] = pfaN * v[k]; P.vl[k] = pfaN * vl[k]; }
    }
    P.eta = eta;
    P.s1 = fit[SIGMA1]; P.s2 = fit[SIGMA2]; P.s3 = fit[SIGMA3]; P.s4 = fit[SIGMA4];
    P.b1 = fit[BETA1]; P.b2 = fit[BETA2]; P.b3 = fit[BETA3];
    for (int i = 0; i < 5; ++i) P.al[i] = fit[ALPHA1 + i];
    P.fRD = fRD; P.fDM = fDM;
    const double fi = kPhiFJoin, fm = 0.5 * fRD;
    const double C2i = detail::dphi_ins(P, fi) - detail::dphi_int(P, fi);
    const double C1i = detail::phi_ins(P, fi) - detail::phi_int(P, fi) - C2i * fi;
    const double C2m = (C2i + detail::dphi_int(P, fm)) - detail::dphi_mrd(P, fm);
    const double C1m = (detail::phi_int(P, fm) + C1i + C2i * fm) - detail::phi_mrd(P, fm) - C2m * fm;
    const double t0 = detail::dphi_mrd(P, fmax_g);
    const double Mf_ref = f_ref * M_sec;
    const double phi_ref = detail::phase_geometric(P, Mf_ref, fi, fm, C1i, C2i, C1m, C2m);
    // phi(f) = IMRPhenDPhase(M f) - t0 (M f - M f_ref) - (2 phase + phi_ref)   [+ tides + 3.5PN spin terms], all / pi below
    const double all_const = t0 * Mf_ref - (2.0 * q.phase + phi_ref);
    const double all_lin = -t0 * M_sec;
    const double ipi = 1.0 / kPi;
    const double a = cbrt(kPi * M_sec), ln_a = log(a);
    // 3.5PN spin-squared / spin-cubed terms of NRTidalv2: 3/(128 eta) v^2 (SS + SSS)
    double ho_spin;
    {
        const double XA = m1M, XB = m2M, XA2 = XA * XA, XB2 = XB * XB, c1s = chi1 * chi1, c2s = chi2 * chi2;
        const double ss = -400.0 * kPi * (qm1 - 1.0) * c1s * XA2 - 400.0 * kPi * (qm2 - 1.0) * c2s * XB2;
        const double sss = 10.0 * ((XA2 + 308.0 / 3.0 * XA) * chi1 + (XB2 - 89.0 / 3.0 * XB) * chi2) * (qm1 - 1.0) * XA2 * c1s +
                           10.0 * ((XB2 + 308.0 / 3.0 * XB) * chi2 + (XA2 - 89.0 / 3.0 * XA) * chi1) * (qm2 - 1.0) * XB2 * c2s -
                           440.0 * (oct1 - 1.0) * XA * XA2 * c1s * chi1 - 440.0 * (oct2 - 1.0) * XB * XB2 * c2s * chi2;
        ho_spin = pfaN * (ss + sss);
    }
    {
        const double a2 = a * a, ia = 1.0 / a, ia2 = ia * ia;
        S.pc0 = ipi * (-kPi / 4.0 + P.v[5] + P.vl[5] * ln_a + all_const);
        S.pcm5 = ipi * P.v[0] * ia2 * ia2 * ia;
        S.pcm3 = ipi * P.v[2] * ia2 * ia;
        S.pcm2 = ipi * P.v[3] * ia2;
        S.pcm1 = ipi * P.v[4] * ia;
        S.pc1 = ipi * (P.v[6] + P.vl[6] * ln_a) * a;
        S.pc2 = ipi * P.v[7] * a2;
        S.ho2 = ipi * ho_spin * a2;
        S.pl5 = ipi * P.vl[5];
        S.pl6 = ipi * P.vl[6] * a;
        const double m13 = cbrt(M_sec);
        S.ps1 = ipi * (P.s1 / eta * M_sec + all_lin);
        S.ps2 = ipi * 0.75 * P.s2 / eta * M_sec * m13;
        S.ps3 = ipi * 0.6 * P.s3 / eta * M_sec * m13 * m13;
        S.ps4 = ipi * 0.5 * P.s4 / eta * M_sec * M_sec;
        S.ic0 = ipi * (C1i + P.b2 / eta * log(M_sec) + all_const);
        S.ic1 = ipi * ((P.b1 / eta + C2i) * M_sec + all_lin);
        S.icm3 = ipi * (-P.b3 / (3.0 * eta)) / (M_sec * M_sec * M_sec);
        S.icl = ipi * 3.0 * P.b2 / eta;
        S.mc0 = ipi * (C1m + all_const);
        S.mc1 = ipi * ((P.al[0] / eta + C2m) * M_sec + all_lin);
        S.mcm1 = ipi * (-P.al[1] / eta) / M_sec;
        S.mc34 = ipi * (4.0 / 3.0 * P.al[2] / eta) * sqrt(M_sec * sqrt(M_sec));
        S.mcat = ipi * P.al[3] / eta;
        S.mfa5 = P.al[4] * fRD / M_sec;
        S.minv_fdm = M_sec / fDM;
        S.fp1 = fi / M_sec; S.fp2 = fm / M_sec;
    }
    // ---------------- tides
    {
        const double XA = m1M, XB = m2M;
        const double XA5 = XA * XA * XA * XA * XA, XB5 = XB * XB * XB * XB * XB;
        const double kappa = 3.0 / 13.0 * ((1.0 + 12.0 * XB / XA) * XA5 * lam1 + (1.0 + 12.0 * XA / XB) * XB5 * lam2);
        S.has_tides = tidal ? 1.0 : 0.0;
        S.xa = a;
        S.tphase = ipi * (-kappa * 2.4375 / (XA * XB));
        S.tamp = -9.0 * kappa;
        S.xa578 = exp(5.78 * ln_a);
        const double num = 1.0 + 3.35411203e-2 * kappa + 4.31460284e-5 * kappa * kappa;
        const double den = 1.0 + 7.54224145e-2 * kappa + 2.23626859e-4 * kappa * kappa;
        S.ft1 = 0.3586 / sqrt(m1 / m2) * num / den / M_sec / (2.0 * kPi);
        S.ft2 = 1.2 * S.ft1;
    }
}

// Antenna response and arrival-time shift of one detector (bilby_cython.geometry: get_polarization_tensor,
// time_delay_from_geocenter; Interferometer.get_detector_response): fills k_re / k_im / k_sq / dt of slot `i`.
//   signal = F+ h+ + Fx hx = (F+ (1 + cos^2 i)/2 - i Fx cos i) h,   shifted by dt = (t_c - start_time) + delay.
// gmst = gmst_ref + gmst_rate (t_c - gmst_ref_time): the host evaluates LAL's GMST polynomial (leap seconds included) once.
GW_HD void project_source(const GwParams& q, const GwDetector& D, int i, double start_time, double gmst_ref_time, double gmst_ref,
                          double gmst_rate, double stride_hz, GwSource& S) {
    const double gmst = fmod(gmst_ref + gmst_rate * (q.geocent_time - gmst_ref_time), 2.0 * kPi);
    const double phi = q.ra - gmst, theta = kPi / 2.0 - q.dec;
    const double cphi = cos(phi), sphi = sin(phi), cth = cos(theta), sth = sin(theta), cpsi = cos(q.psi), spsi = sin(q.psi);
    const double u[3] = {cphi * cth, cth * sphi, -sth};
    const double v[3] = {-sphi, cphi, 0.0};
    double m[3], n[3];
    for (int k = 0; k < 3; ++k) { m[k] = -u[k] * spsi - v[k] * cpsi; n[k] = -u[k] * cpsi + v[k] * spsi; }
    double fp = 0.0, fc = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            fp += D.tensor[3 * r + c] * (m[r] * m[c] - n[r] * n[c]);
            fc += D.tensor[3 * r + c] * (m[r] * n[c] + n[r] * m[c]);
        }
    const double omega[3] = {sth * cphi, sth * sphi, cth};
    const double delay = -(omega[0] * D.vertex[0] + omega[1] * D.vertex[1] + omega[2] * D.vertex[2]) / kC;
    const double ci = cos(q.theta_jn);
    S.k_re[i] = fp * 0.5 * (1.0 + ci * ci);
    S.k_im[i] = -fc * ci;
    S.k_sq[i] = S.k_re[i] * S.k_re[i] + S.k_im[i] * S.k_im[i];
    S.dt[i] = (q.geocent_time - start_time) + delay;
    double turns = stride_hz * S.dt[i];
    turns -= rint(turns);
    S.rs_re[i] = cos(2.0 * kPi * turns);
    S.rs_im[i] = -sin(2.0 * kPi * turns);
}

// a / b to (nearly) full precision without the IEEE division sequence: hardware reciprocal estimate + two Newton steps
// (the per-bin loop divides twice per sample; relative error <= 2 ulp, far inside the phase budget)
GW_HD double fast_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return a * r;
#else
    return a / b;
#endif
}

// The waveform at one frequency bin: amplitude (strain, before the antenna factor) and phase / pi, such that
//   h(f) = amp * exp(-i pi phase_over_pi).
// LAL: IMRPhenDAmplitude, IMRPhenDPhase, SimNRTunedTidesFDTidalPhase_v2, SimNRTunedTidesFDTidalAmplitude, PlanckTaper.
// PR / AR: the region of the phase / of the amplitude when the caller knows it for every lane it runs (0 inspiral,
// 1 intermediate, 2 merger-ringdown; -1: decide per bin); PLAIN: the caller knows that the bin is below the Planck taper and below
// f_cut, so neither is tested.  The sample-per-lane kernel classifies a whole chunk of bins once and runs the matching
// instantiation: only that region's constants are live in registers.
template <int PR, int AR, bool PLAIN>
GW_HD void eval_bin_t(const GwSource& S, const GwBin& b, double& amp, double& phase_over_pi) {
    const double f = b.f, f13 = b.f13, inv13 = b.inv13;
    const double f23 = f13 * f13;
    // ---- phase
    double ph;
    if (PR == 0 || (PR < 0 && f < S.fp1)) {
        const double inv2 = inv13 * inv13;
        ph = S.pc0 + inv13 * (S.pcm1 + inv13 * (S.pcm2 + inv13 * (S.pcm3 + inv2 * S.pcm5))) + f13 * (S.pc1 + f13 * S.pc2) +
             (S.pl5 + S.pl6 * f13) * b.lnf13 + f * (S.ps1 + f13 * (S.ps2 + f13 * S.ps3) + f * S.ps4);
    } else if (PR == 1 || (PR < 0 && f < S.fp2)) {
        const double inv3 = inv13 * inv13 * inv13;
        ph = S.ic0 + S.ic1 * f + S.icm3 * (inv3 * inv3 * inv3) + S.icl * b.lnf13;
    } else {
        ph = S.mc0 + S.mc1 * f + S.mcm1 / f + S.mc34 * sqrt(f * sqrt(f)) + S.mcat * atan((f - S.mfa5) * S.minv_fdm);
    }
    ph += S.ho2 * f23;
    // ---- amplitude bracket
    double br;
    if (AR == 0 || (AR < 0 && f < S.fa1)) {
        br = 1.0 + f23 * (S.ai[0] + f13 * (S.ai[1] + f13 * (S.ai[2] + f13 * (S.ai[3] + f13 * (S.ai[4] + f13 * (S.ai[5] + f13 * (S.ai[6] + f13 * S.ai[7])))))));
    } else if (AR == 1 || (AR < 0 && f < S.fa3)) {
        const double u = (f - S.fa1) * S.iu_scale;
        br = S.ip[0] + u * (S.ip[1] + u * (S.ip[2] + u * (S.ip[3] + u * S.ip[4])));
    } else {
        const double x = f - S.fRD;
        br = exp(-x * S.mg2w) * S.mg1w / (x * x + S.mw * S.mw);
    }
    double taper = 1.0;
    if (S.has_tides != 0.0) {
        const double xh = S.xa * f13, x = xh * xh, x2 = x * x;
        const double num = 1.0 + x * (-12.615214237993088 + xh * 19.0537346970349 + x * (-21.166863146081035 + xh * 90.55082156324926 + x * -60.25357801943598));
        const double den = 1.0 + x * (-15.11120782773667 + xh * 22.195327350624694 + x * 8.064109635305156);
        // one reciprocal for the two rational functions: N/D = N Q r, P/Q = P D r with r = 1 / (D Q)
        const double x289 = S.xa578 * b.p578;
        const double pnum = 1.0 + 4.157407407407407 * x + 2519.111111111111 * x289, pden = 1.0 + 13477.8073677 * x2 * x2;
        const double r = fast_div(1.0, den * pden);
        ph += S.tphase * (x2 * xh) * (num * pden * r);
        br += S.tamp * (x2 * x2 * x) * (pnum * den * r);
        if (!PLAIN && f > S.ft1) {
            if (f >= S.ft2) taper = 0.0;
            else {
                const double w = S.ft2 - S.ft1;
                taper = 1.0 - 1.0 / (exp(w / (f - S.ft1) + w / (f - S.ft2)) + 1.0);
            }
        }
    }
    if (PLAIN) amp = S.amp_scale * b.fm76 * br;
    else amp = (f > S.f_cut || taper == 0.0) ? 0.0 : S.amp_scale * b.fm76 * br * taper;
    phase_over_pi = ph;
}

GW_HD void eval_bin(const GwSource& S, const GwBin& b, double& amp, double& phase_over_pi) {
    eval_bin_t<-1, -1, false>(S, b, amp, phase_over_pi);
}

// ln I0(x) for x >= 0 (phase marginalisation: bilby's ln_i0 = log(ive(0, x)) + x): power series below 15, the asymptotic
// series above.  Checked against scipy.special.ive in tests/test_hostcheck_gw.py.
GW_HD double ln_bessel_i0(double x) {
    x = fabs(x);
    if (x < 15.0) {
        const double q = 0.25 * x * x;
        double term = 1.0, sum = 1.0;
        for (int k = 1; k < 80; ++k) {
            term *= q / ((double)k * (double)k);
            sum += term;
            if (term < 1e-17 * sum) break;
        }
        return log(sum);
    }
    const double y = 1.0 / (8.0 * x);
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 30; ++k) {
        const double t = (2.0 * k - 1.0);
        const double next = term * t * t / (double)k * y;
        if (fabs(next) >= fabs(term)) break;
        term = next;
        sum += term;
        if (fabs(term) < 1e-17 * sum) break;
    }
    return x - 0.5 * log(2.0 * kPi * x) + log(sum);
}

}  // namespace gw
}  // namespace nmma
