// em_math.h -- scalar fp64 building blocks of the EM likelihood, shared by the HIP
// kernels (device) and by tests/hostcheck (host build of the very same source, so the
// branchy numerics can be checked on a CPU against scipy / the oracle).
//
// Everything here follows numpy / scipy semantics of the reference's call sites:
//   np.interp                     numpy/_core/src/multiarray/compiled_base.c (arr_interp)
//   scipy.stats.truncnorm.logpdf  nmma/em/em_likelihood.py:252-256
//   scipy.stats.norm.logsf        nmma/em/em_likelihood.py:247-249
// Build with -ffp-contract=off: the reference's expressions are not fused.
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/nmma_hip.h"

#if defined(__HIPCC__)
#define NM_HD __host__ __device__ __forceinline__
// rarely-taken, transcendental-heavy paths: keep them out of line so they do not
// inflate the register budget of the MFMA loop they share a kernel with
#define NM_HD_COLD inline __host__ __device__ __noinline__
#else
#define NM_HD inline
#define NM_HD_COLD inline
#define NM_TAB_QUAL static const
#endif
#include "logphi_tab.h"

namespace nmma {

constexpr double kPi = 3.141592653589793;            // np.pi
constexpr double kSqrt1_2 = 0.7071067811865476;      // M_SQRT1_2
constexpr double kNormPdfLogC = 0.9189385332046727;  // log(sqrt(2*pi)) (scipy _norm_pdf_logC)

NM_HD double dinf() { return HUGE_VAL; }
NM_HD double dnan() { return HUGE_VAL - HUGE_VAL; }

// ---------------------------------------------------------------------------
// theta column -> physical scalar (nmma/core/conversion.py:119-126, em/model.py:272-286)
// ---------------------------------------------------------------------------
NM_HD_COLD double apply_slot_op(double v, int op) {
    switch (op) {
        case NMMA_OP_RAD2DEG: return v * 180.0 / kPi;
        case NMMA_OP_DEG2RAD: return v / 180.0 * kPi;
        case NMMA_OP_LOG10: return log10(v);
        case NMMA_OP_POW10: return pow(10.0, v);
        case NMMA_OP_THETAJN2DEG: { const double t = fmin(v, kPi - v); return t * 180.0 / kPi; }
        case NMMA_OP_COSTHETAJN2DEG: { double t = acos(v); t = fmin(t, kPi - t); return t * 180.0 / kPi; }
        case NMMA_OP_ACOS: return acos(v);
        default: return v;
    }
}

// (RowPtr: any pointer to double -- generic, or an LDS/global address-space pointer in device code)
// The cheap conversions are evaluated in line; only the transcendental ones take the out-of-line call.
template <class RowPtr>
NM_HD double apply_slot(const nmma_slot& s, RowPtr row) {
    if (s.col < 0) return s.value;
    const double v = row[s.col];
    if (s.op == NMMA_OP_IDENT) return v;
    if (s.op == NMMA_OP_RAD2DEG) return v * 180.0 / kPi;
    if (s.op == NMMA_OP_DEG2RAD) return v / 180.0 * kPi;
    return apply_slot_op(v, s.op);
}

// ---------------------------------------------------------------------------
// autocomplete_data's finite mask on a filter group's SAMPLED node values (em/utils.py:634-645 via systematics.py:288-291,
// extrapolate = "constant"): non-finite nodes are dropped, np.interp runs over the rest with constant ends, fewer than two finite
// nodes give ref_value = +inf everywhere.
//   repair_nodes: in place on v[0], v[stride], ... -- a dropped node takes the value the interpolant through its finite neighbours
//     has at its time (the constant end value beyond the first / last finite node), which leaves that interpolant unchanged;
//   masked_nodes_at: the interpolant's value at time t straight from the nodes (the paths that read theta per datum).
// Both are cold: a bounded prior never produces the input.
// ---------------------------------------------------------------------------
NM_HD_COLD void repair_nodes(double* v, const long stride, const double* xn, const int n) {
    int nfin = 0;
    for (int j = 0; j < n; ++j) nfin += (v[j * stride] - v[j * stride] == 0.0) ? 1 : 0;
    if (nfin == n) return;
    // (fewer than two finite nodes: sigma_sys = +inf for every datum -- written as a huge finite value so that the consumers' node
    //  arithmetic, (v1 - v0) / dx * off + v0, stays finite; sigma_tot^2 overflows to +inf either way: every datum an upper limit)
    if (nfin < 2) { for (int j = 0; j < n; ++j) v[j * stride] = 1e300; return; }
    // (two sweeps over the ORIGINAL finiteness: replaced values are marked by a NaN until the second sweep fills them)
    for (int j = 0; j < n; ++j) if (!(v[j * stride] - v[j * stride] == 0.0)) v[j * stride] = dnan();
    for (int j = 0; j < n; ++j) {
        if (v[j * stride] == v[j * stride]) continue;
        int a = j - 1, b = j + 1;
        while (a >= 0 && !(v[a * stride] == v[a * stride])) --a;
        while (b < n && !(v[b * stride] == v[b * stride])) ++b;
        // (a replaced node to the left has been filled by this sweep already: it lies on the same interpolant)
        double r;
        if (a >= 0 && b < n) {
            const double slope = (v[b * stride] - v[a * stride]) / (xn[b] - xn[a]);
            r = slope * (xn[j] - xn[a]) + v[a * stride];
        } else r = a >= 0 ? v[a * stride] : v[b * stride];
        v[j * stride] = r;
    }
}

template <class RowPtr>
NM_HD double masked_nodes_at(const nmma_slot* sv, const double* xn, const int n, const double t, RowPtr row) {
    int nfin = 0, first = -1, last = -1;
    for (int j = 0; j < n; ++j) {
        const double v = apply_slot(sv[j], row);
        if (v - v == 0.0) { ++nfin; if (first < 0) first = j; last = j; }
    }
    if (nfin < 2) return dinf();
    if (t <= xn[first]) return apply_slot(sv[first], row);
    if (t >= xn[last]) return apply_slot(sv[last], row);
    int a = first, b = last;
    for (int j = first; j <= last; ++j) {
        const double v = apply_slot(sv[j], row);
        if (!(v - v == 0.0)) continue;
        if (xn[j] <= t) a = j;
        else { b = j; break; }
    }
    const double va = apply_slot(sv[a], row), vb = apply_slot(sv[b], row);
    return ((vb - va) / (xn[b] - xn[a])) * (t - xn[a]) + va;
}
// ---------------------------------------------------------------------------
// np.interp on explicit arrays (cosmology grid, systematics nodes)
// ---------------------------------------------------------------------------
// Piecewise-linear value between two nodes exactly as arr_interp evaluates it,
// including its NaN fallbacks.
NM_HD double lerp_np(double x, double x0, double x1, double y0, double y1) {
    const double slope = (y1 - y0) / (x1 - x0);
    double r = slope * (x - x0) + y0;
    if (r != r) {
        r = slope * (x - x1) + y1;
        if (r != r && y0 == y1) r = y0;
    }
    return r;
}

// np.interp(x, xp[0..n), fp[0..n), left, right); xp increasing, n >= 1.
template <class XPtr, class FPtr>
NM_HD double interp_np(double x, XPtr xp, FPtr fp, int n, double left, double right) {
    if (x != x) return x;
    if (n == 1) return x < xp[0] ? left : (x > xp[0] ? right : fp[0]);
    if (x < xp[0]) return left;
    if (x > xp[n - 1]) return right;
    int lo = 0, hi = n - 1;  // xp[lo] <= x <= xp[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    if (x == xp[n - 1]) return fp[n - 1];
    if (xp[lo] == x) return fp[lo];
    return lerp_np(x, xp[lo], xp[lo + 1], fp[lo], fp[lo + 1]);
}

// ---------------------------------------------------------------------------
// scipy.special pieces (xsf): ndtr, log_ndtr
// ---------------------------------------------------------------------------
NM_HD double erfcx_pos_or_neg(double t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ::erfcx(t);
#else
    // host build (tests only): libm has no erfcx; large-t continued fraction keeps the
    // product finite where exp(t*t) would overflow.
    if (t < 25.0) return exp(t * t) * erfc(t);
    const double t2 = t * t;
    return (1.0 / (t * 1.772453850905516)) * (1.0 - 0.5 / t2 + 0.75 / (t2 * t2));
#endif
}

// xsf::cephes::ndtr
NM_HD double ndtr(double a) {
    if (a != a) return a;
    const double x = a * kSqrt1_2;
    const double z = fabs(x);
    if (z < kSqrt1_2) return 0.5 + 0.5 * erf(x);
    double y = 0.5 * erfc(z);
    if (x > 0) y = 1.0 - y;
    return y;
}

// xsf::log_ndtr (scipy >= 1.9): log(erfcx(-t)/2) - t^2 for x < -1, log1p(-erfc(t)/2) otherwise
NM_HD double log_ndtr_inline(double x) {
    const double t = x * kSqrt1_2;
    if (x < -1.0) return log(erfcx_pos_or_neg(-t) / 2) - t * t;
    return log1p(-erfc(t) / 2);
}
NM_HD_COLD double log_ndtr(double x) { return log_ndtr_inline(x); }

// scipy.stats._continuous_distns._log_gauss_mass(a = -inf, b)
NM_HD_COLD double log_gauss_mass_neginf(double b) {
#ifdef NMMA_DBG_NOMASS      // measurement build: what the truncation mass costs the general lean task
    return b * 1e-300;
#endif
    if (b <= 0) return log_ndtr(b);       // case_left: log_ndtr(b) + log1p(-exp(-inf)) = log_ndtr(b)
    if (b > 0) return log1p(-ndtr(-b));   // case_central: log1p(-ndtr(a) - ndtr(-b)), ndtr(-inf) = 0
    return dnan();
}

// One detection: truncnorm.logpdf(m, a=-inf, b=(lim-est)/sigma, loc=est, scale=sigma)
// rv_continuous.logpdf: NaN for invalid args (b NaN, b <= a, scale <= 0), -inf outside [a, b].
NM_HD double detection_term(double m, double est, double sigma, double log_sigma, double lim) {
    if (lim == dinf()) {   // untruncated: b = +inf for finite est (mass = log 1 = 0), NaN otherwise
        if (!(est < dinf()) || !(sigma > 0)) return dnan();
        const double x = (m - est) / sigma;
        if (x != x) return dnan();
        return ((-(x * x) / 2.0 - kNormPdfLogC) - 0.0) - log_sigma;
    }
    const double b = (lim - est) / sigma;
    if (!(b > -dinf()) || !(sigma > 0)) return dnan();   // also catches b = NaN (est = +inf)
    const double x = (m - est) / sigma;
    if (x != x) return dnan();
    if (x > b) return -dinf();
    const double mass = (lim == dinf()) ? 0.0 : log_gauss_mass_neginf(b);
    return ((-(x * x) / 2.0 - kNormPdfLogC) - mass) - log_sigma;
}

// log Phi(b) -- the truncation mass of a detection under a finite limit, and norm.logsf of an upper limit -- from the table of
// logphi_tab.h (`tab`: the table, in LDS for the kernels): relative error <= 5e-16 on [-9.5, 8.5) (4e-16 absolute above -1); 0 beyond 8.5
// (9.5e-18: absorbed by the sum it enters, see the generator); scipy's own formula below -9.5 and for NaN (out of line).  One clamp, one
// index, fourteen coefficients, thirteen FMAs -- against erfc + log1p or erfcx + log (~210 instructions).
template <typename TabPtr>
NM_HD double log_gauss_mass_tab(double b, TabPtr tab) {
    double bc = b > LOGPHI_LO ? b : LOGPHI_LO;                 // (NaN -> LO: the value is replaced below)
    bc = bc < LOGPHI_HI ? bc : LOGPHI_HI;
    int idx = (int)((bc - LOGPHI_LO) * LOGPHI_INV_H);
    idx = idx > LOGPHI_NINT - 1 ? LOGPHI_NINT - 1 : idx;
    const double t = (bc - (LOGPHI_LO + (idx + 0.5) * LOGPHI_H)) * (2.0 * LOGPHI_INV_H);
    const TabPtr row = tab + idx * LOGPHI_ROW;
    double acc = row[0];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 1; k <= LOGPHI_DEG; ++k) acc = fma(acc, t, row[k]);
    acc = b >= LOGPHI_HI ? 0.0 : acc;
    if (!(b >= LOGPHI_LO)) acc = log_ndtr(b);       // (= _log_gauss_mass(-inf, b) for b <= 0; NaN for NaN.  ONE out-of-line call: a nested one costs the kernel a stack)
    return acc;
}

// detection_term with the truncation mass from the table (the general lean task of em_logl: finite detection limits)
template <typename TabPtr>
NM_HD double detection_term_tab(double m, double est, double sigma, double log_sigma, double lim, TabPtr tab) {
    if (lim == dinf()) {   // untruncated, as in detection_term
        if (!(est < dinf()) || !(sigma > 0)) return dnan();
        const double x = (m - est) / sigma;
        if (x != x) return dnan();
        return ((-(x * x) / 2.0 - kNormPdfLogC) - 0.0) - log_sigma;
    }
    const double b = (lim - est) / sigma;
    if (!(b > -dinf()) || !(sigma > 0)) return dnan();   // also catches b = NaN (est = +inf)
    const double x = (m - est) / sigma;
    if (x != x) return dnan();
    if (x > b) return -dinf();
    const double mass = log_gauss_mass_tab(b, tab);
    return ((-(x * x) / 2.0 - kNormPdfLogC) - mass) - log_sigma;
}

// upper_limit_term with log Phi from the table (the lean tasks of em_logl: one or a few upper limits per filter, evaluated on the lanes
// that hold one while the rest of the wave waits -- 1.05 of 28 us at BASELINE config 2 with scipy's formula out of line)
template <typename TabPtr>
NM_HD double upper_limit_term_tab(double m, double est, double sigma_sys, TabPtr tab) {
    const double x = (m - est) / sigma_sys;
    double r = log_gauss_mass_tab(-x, tab);
    r = (x == -dinf()) ? 0.0 : r;
    r = (x == dinf()) ? -dinf() : r;
    return (!(sigma_sys > 0) || x != x) ? dnan() : r;
}

// The same two terms with the table read from global memory (kLogPhiTab: 2 KB, cache-resident), OUT OF LINE: for the kernels where the
// term is rare per lane and the code size matters more than the call (em_lc_loglike's general datum, the fallback flavours of em_logl)
NM_HD_COLD double upper_limit_term_gtab(double m, double est, double sigma_sys) {
    return upper_limit_term_tab(m, est, sigma_sys, static_cast<const double*>(kLogPhiTab));
}
NM_HD_COLD double detection_term_gtab(double m, double est, double sigma, double log_sigma, double lim) {
    return detection_term_tab(m, est, sigma, log_sigma, lim, static_cast<const double*>(kLogPhiTab));
}

// One upper limit: norm.logsf(m, est, sigma_sys) = log_ndtr(-(m - est)/sigma_sys);
// rv_continuous.logsf: scale <= 0 or NaN args -> NaN; x at the lower support edge -> 0.
NM_HD double upper_limit_term(double m, double est, double sigma_sys) {
#ifdef NMMA_DBG_NOUL        // measurement build: what the out-of-line log_ndtr of an upper limit costs a lean task
    return (m - est) * sigma_sys * 1e-300;
#endif
    if (!(sigma_sys > 0)) return dnan();
    const double x = (m - est) / sigma_sys;
    if (x != x) return dnan();
    if (x == -dinf()) return 0.0;
    if (x == dinf()) return -dinf();
    return log_ndtr(-x);
}
// the same without a function call (for a kernel whose register file is full of in-flight loads: a call would spill)
NM_HD double upper_limit_term_inline(double m, double est, double sigma_sys) {
    if (!(sigma_sys > 0)) return dnan();
    const double x = (m - est) / sigma_sys;
    if (x != x) return dnan();
    if (x == -dinf()) return 0.0;
    if (x == dinf()) return -dinf();
    return log_ndtr_inline(-x);
}

// ---------------------------------------------------------------------------
// distance quantities (nmma/core/conversion.py:30-34, em/model.py:389)
// ---------------------------------------------------------------------------
NM_HD double distance_modulus(double d_lum) { return 5.0 * (5 + log10(d_lum)); }
NM_HD double redshift_correction(double z) { return -2.5 * log10(1 + z); }

// ---------------------------------------------------------------------------
// Host-galaxy extinction, Pei (1992) SMC curve: extinctionFactorP92SMC (nmma/em/utils.py:373-428) followed by
// get_extinction_mags' -2.5 log10 (em/model.py:323-342).  The curve itself is third-party there
// (dust_extinction.shapes.P92, v1.x): A(lam)/A(V) = sum_i a_i / ((lam/lam_i)^n_i + (lam/lam_i)^-n_i + b_i) over
// the six terms BKG, FUV, NUV, SIL1, SIL2, FIR in this order, amplitudes converted from the B to the V reference
// with AbAv = 1/3.08 + 1, valid for 1e-3 <= 1/lam[um] <= 1e3.  nu: observer-frame filter frequency [Hz];
// the curve is read at the HOST-frame wavelength c / (nu (1+z)); R_V = 2.93.  Outside the curve's range (or
// above the reference's 2e16 Hz cut-off) the factor is 1.  Parity vs dust_extinction is unpinned (absent here).
// ---------------------------------------------------------------------------
NM_HD double p92_term(double lam, double amp, double cen, double b, double n) {
    const double l_norm = lam / cen;
    return amp / (pow(l_norm, n) + pow(l_norm, -1 * n) + b);
}
NM_HD double p92_smc_ext_mag(double nu, double zp1, double ebv) {
    const double c_cgs = 29979245800.0;
    const double nu_lo = (1.0 / 1e3) * 1e4 * c_cgs;
    double nu_hi = (1.0 / 1e-3) * 1e4 * c_cgs;
    if (2e16 < nu_hi) nu_hi = 2e16;
    const double nu_host = nu * zp1;
    double ext = 1.0;
    if (nu_host >= nu_lo && nu_host <= nu_hi) {
        const double x = 1.0 / ((c_cgs / nu_host) * 1e4);      // wavenumber in 1/um
        const double lam = 1.0 / x;
        const double abav = 1.0 / 3.08 + 1.0;
        double axav = p92_term(lam, 185.0 * abav, 0.042, 90.0, 2.0);
        axav = axav + p92_term(lam, 27 * abav, 0.08, 5.5, 4.0);
        axav = axav + p92_term(lam, 0.005 * abav, 0.22, -1.95, 2.0);
        axav = axav + p92_term(lam, 0.010 * abav, 9.7, -1.95, 2.0);
        axav = axav + p92_term(lam, 0.012 * abav, 18.0, -1.80, 2.0);
        axav = axav + p92_term(lam, 0.030 * abav, 25.0, 0.0, 2.0);
        const double av = 2.93 * ebv;
        ext = pow(10.0, -0.4 * axav * av);
    }
    return -2.5 * log10(ext);
}
// Chebyshev series sum_k c_k T_k(t), t = (z - row[0]) * row[1], c_k = row[2 + k], k < 14 (Clenshaw): the lean task's Pei-1992
// extinction per unit E(B-V) as a function of the redshift (EmDev::p92_cheb)
template <typename RowPtr>
NM_HD double cheb14_eval(RowPtr row, double z) {
    const double t = (z - row[0]) * row[1], t2 = t + t;
    double b1 = 0.0, b2 = 0.0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 13; k >= 1; --k) {
        const double b0 = fma(t2, b1, row[2 + k] - b2);
        b2 = b1; b1 = b0;
    }
    return fma(t, b1, row[2] - b2);
}

// extinction magnitude of one (sample, model filter): `coeff` is ebv_coeff[m] for the linear law and the filter
// frequency for P92 (law: enum nmma_extinction_law); nothing is applied at Ebv == 0 (model.py:328-330)
NM_HD double extinction_mag(int law, double coeff, double zp1, double ebv) {
    if (ebv == 0.0) return 0.0;
    return law == 1 ? p92_smc_ext_mag(coeff, zp1, ebv) : coeff * ebv;
}

}  // namespace nmma
