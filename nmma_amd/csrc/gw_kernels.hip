// gw_kernels.hip -- gfx950 kernels and host API of the gravitational-wave leg of the joint likelihood
// (BASELINE config 5).  Second translation unit of libnmma_hip.so.
//
// Reference path (third-party arithmetic, see gw_math.h): nmma/gw/gw_likelihood.py:97-247 ->
//   bilby.gw.likelihood.GravitationalWaveTransient.log_likelihood_ratio:
//     waveform_generator.frequency_domain_strain(theta)         lalsimulation IMRPhenomD_NRTidalv2 on the full frequency array
//     for each interferometer: get_detector_response            F+ h+ + Fx hx, time shift to the detector
//                              <d|h>, <h|h>                      4/T sum conj(a) b / S over the frequency mask
//     sum_ifo Re<d|h> - <h|h>/2   (or ln I0(|<d|h>|) - <h|h>/2 with phase marginalisation)
// i.e. per sample ~n_freq x (waveform + n_ifo projections) -- 2.6e5 bins x 3 detectors for a 128 s segment sampled at 4096 Hz.
//
// Here: the strain is never written.
//   gw_source_kernel   one thread per sample: theta row -> GwSource record (all frequency-independent quantities: Table V fits,
//                      PN coefficients folded with the total mass, connection coefficients, tidal constants, per-detector antenna
//                      factor and arrival time).  Heavy scalar code, run once per sample.
//   gw_logl_kernel    the hot loop.  A 256-thread workgroup owns a chunk of 8192 consecutive bins and a group of 16 samples;
//                      lanes map to consecutive bins (coalesced 32-byte loads of the per-bin basis and of the pre-weighted
//                      data, L2-resident: sample groups are the fast grid index, so the workgroups in flight share a handful of
//                      chunks).  Samples are taken ONE at a time: the chunk's region of phase and amplitude is then a scalar
//                      decision and the matching instantiation of the bin loop keeps just that region's constants in SGPRs.
//                      The linear phase exp(-2 pi i f dt) per detector (antenna factor folded in) advances by a complex
//                      multiplication per pass; one sincos per (bin, sample).  Per-lane sums are reduced with DPP and leave
//                      through LDS in a fixed order.
//   gw_finish_kernel   sums the chunk partials in chunk order (the order does not depend on the batch size), applies 4/T,
//                      the phase marginalisation and the floor.
// Roofline: fp64 vector FMA (78.6 TFLOP/s); algorithmic flops per (bin, sample) are counted in DESIGN section 3.5, which also
// lists the three formulations measured on the way (43.0 -> 28.9 ms at config 5's shape).
//
// Also here: gw_loglike_ratio_kernel, the HBM-streaming reduction for strain supplied by the caller (round 2).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "em_math.h"
#include "gw_math.h"
#include "nmma_common.h"

namespace nmma {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double gw_dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes of a wave; the total lands in the LAST row (lanes 48..63)
__device__ __forceinline__ double gw_wave_sum(double v) {
    v += gw_dpp_mov_f64<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v += gw_dpp_mov_f64<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v += gw_dpp_mov_f64<0x141, 0xf>(v);    // row_half_mirror
    v += gw_dpp_mov_f64<0x140, 0xf>(v);    // row_mirror: every lane holds its row's sum
    v += gw_dpp_mov_f64<0x142, 0xA>(v);    // row_bcast15 into rows 1 and 3
    v += gw_dpp_mov_f64<0x143, 0xC>(v);    // row_bcast31 into rows 2 and 3
    return v;
}

// =======================================================================================
// strain supplied by the caller (round 2): one pass over strain[B][n_ifo * n_freq]
// =======================================================================================
constexpr int GW_THREADS = 1024;

template <int GW_SAMPLES>
__global__ __launch_bounds__(GW_THREADS) void gw_loglike_ratio_kernel(
    const double2* __restrict__ strain,      // [B][n_ifo * n_freq]
    const double2* __restrict__ data,        // [n_ifo * n_freq]
    const double* __restrict__ weight,       // [n_ifo * n_freq]  = mask_f / S_f
    const long B, const long n, const double four_over_T, double* __restrict__ out) {
    __shared__ double wsum[GW_SAMPLES][GW_THREADS / 64];
    typedef double f64x2v __attribute__((ext_vector_type(2)));
    const long b0 = (long)blockIdx.x * GW_SAMPLES;
    const f64x2v* h[GW_SAMPLES];
#pragma unroll
    for (int q = 0; q < GW_SAMPLES; ++q) {
        const long b = b0 + q < B ? b0 + q : B - 1;      // (a sample beyond the batch re-reads the last one; nothing is stored)
        h[q] = reinterpret_cast<const f64x2v*>(strain + b * n);
    }
    double acc[GW_SAMPLES][2];
#pragma unroll
    for (int q = 0; q < GW_SAMPLES; ++q) { acc[q][0] = 0.0; acc[q][1] = 0.0; }
    auto term = [&](const f64x2v hv, const double2 dv, const double w) -> double {
        const double dh = dv.x * hv[0] + dv.y * hv[1];
        const double hh = hv[0] * hv[0] + hv[1] * hv[1];
        return (dh - hh / 2.0) * w;
    };
    long i = threadIdx.x;
    for (; i + GW_THREADS < n; i += 2 * GW_THREADS) {
        f64x2v hv[GW_SAMPLES][2];
#pragma unroll
        for (int q = 0; q < GW_SAMPLES; ++q) {
            hv[q][0] = __builtin_nontemporal_load(h[q] + i);
            hv[q][1] = __builtin_nontemporal_load(h[q] + i + GW_THREADS);
        }
        const double2 d0 = data[i], d1 = data[i + GW_THREADS];
        const double w0 = weight[i], w1 = weight[i + GW_THREADS];
#pragma unroll
        for (int q = 0; q < GW_SAMPLES; ++q) { acc[q][0] += term(hv[q][0], d0, w0); acc[q][1] += term(hv[q][1], d1, w1); }
    }
    for (; i < n; i += GW_THREADS) {
        const double2 d0 = data[i];
        const double w0 = weight[i];
#pragma unroll
        for (int q = 0; q < GW_SAMPLES; ++q) acc[q][0] += term(__builtin_nontemporal_load(h[q] + i), d0, w0);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < GW_SAMPLES; ++q) {
        const double a = gw_wave_sum(acc[q][0] + acc[q][1]);
        if (lane == 63) wsum[q][wave] = a;
    }
    __syncthreads();
    if (threadIdx.x < GW_SAMPLES && b0 + threadIdx.x < B) {
        double s = 0.0;
        for (int w = 0; w < GW_THREADS / 64; ++w) s += wsum[threadIdx.x][w];
        out[b0 + threadIdx.x] = four_over_T * s;
    }
}

// =======================================================================================
// GW log-likelihood from parameters
// =======================================================================================
constexpr int GWL_THREADS = 256;
constexpr int GWL_WAVES = GWL_THREADS / 64;

struct GwDev {
    int32_t n_ifo, tidal, mass_mode, phase_marg;
    int32_t n_dist, pad_dist;      // distance marginalisation: grid points (0: off)
    const double* dist_grid;       // [n_dist] Mpc
    const double* dist_logw;       // [n_dist] ln(prior(d_j) delta_d)
    const double* time_logw;       // time marginalisation: [n_freq - 1] ln(prior(t_j) delta_t), or null
    int64_t tm_lo, tm_hi;          // first / one past the last time index that can carry weight (one node of slack with jitter)
    int32_t tm_jitter, pad_tm;     // bilby's jitter_time: per-row weights from the prior's bounds
    double tm_min, tm_max, tm_dt;  // time prior bounds [GPS s], spacing of the shifts
    nmma_slot time_jitter;
    int64_t n_bins;                // bins k0 .. k0 + n_bins - 1 of the frequency array
    int64_t k0, n_freq;
    int32_t n_chunks, n_dim;       // chunks of GWL_CHUNK bins
    double df, f_ref, start_time, gmst_ref_time, gmst_ref, gmst_rate, four_over_T;
    const double4* basis;          // [n_bins] {f13, 1/f13, ln f13, f^(-7/6)}
    const double* basis5;          // [n_bins] f13^5.78
    const double4* dat;            // [n_ifo][n_bins] {w d_re, w d_im, w, 0},  w = mask / S
    gw::GwDetector det[gw::kMaxIfo];
    nmma_slot mass_a, mass_b, chi_1, chi_2, lambda_1, lambda_2, luminosity_distance, theta_jn, phase, ra, dec, psi, geocent_time;
};

__global__ __launch_bounds__(64) void gw_source_kernel(const GwDev* __restrict__ Pp, const double* __restrict__ theta, const long B,
                                                       const long ld, const double stride_hz, gw::GwSource* __restrict__ src) {
    const GwDev& P = *Pp;
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* row = theta + b * ld;
    gw::GwParams q;
    const double ma = apply_slot(P.mass_a, row), mb = apply_slot(P.mass_b, row);
    if (P.mass_mode == NMMA_GW_CHIRP_MASS_RATIO) {
        // bilby/gw/conversion.py: chirp_mass_and_mass_ratio_to_total_mass, total_mass_and_mass_ratio_to_component_masses
        const double total = ma * pow(1.0 + mb, 1.2) / pow(mb, 0.6);
        q.mass_1 = total / (1.0 + mb);
        q.mass_2 = q.mass_1 * mb;
    } else {
        q.mass_1 = ma; q.mass_2 = mb;
    }
    q.chi_1 = apply_slot(P.chi_1, row); q.chi_2 = apply_slot(P.chi_2, row);
    q.lambda_1 = apply_slot(P.lambda_1, row); q.lambda_2 = apply_slot(P.lambda_2, row);
    q.luminosity_distance = apply_slot(P.luminosity_distance, row);
    q.theta_jn = apply_slot(P.theta_jn, row); q.phase = apply_slot(P.phase, row);
    q.ra = apply_slot(P.ra, row); q.dec = apply_slot(P.dec, row); q.psi = apply_slot(P.psi, row);
    q.geocent_time = apply_slot(P.geocent_time, row);
    if (P.tm_jitter) q.geocent_time += apply_slot(P.time_jitter, row);       // bilby: parameters['geocent_time'] += parameters['time_jitter']
    gw::GwSource S;
    for (int i = 0; i < gw::kMaxIfo; ++i) { S.k_re[i] = 0.0; S.k_im[i] = 0.0; S.k_sq[i] = 0.0; S.dt[i] = 0.0; S.rs_re[i] = 1.0; S.rs_im[i] = 0.0; }
    gw::setup_source(q, P.f_ref, P.tidal != 0, S);
    S.distance = q.luminosity_distance;
    S.jitter = P.tm_jitter ? apply_slot(P.time_jitter, row) : 0.0;
    if (S.valid != 0.0)
        for (int i = 0; i < P.n_ifo; ++i) gw::project_source(q, P.det[i], i, P.start_time, P.gmst_ref_time, P.gmst_ref, P.gmst_rate, stride_hz, S);
    src[b] = S;
}

// exp(i pi t) = (re, im) for any finite t of moderate size (|t| < 2^51: phases here stay below 1e7).  Exact reduction to
// y in [-1/4, 1/4] (t - 2 rint(t / 2) and the quadrant k = rint(2 r) are exact in binary floating point), Taylor polynomials of
// sin(pi y) / y and cos(pi y) in y^2 (truncation error < 5e-17 on the interval), quadrant fix-up by swap and sign.  31
// instructions against the 45 of the library's sincospi, which also covers inf / NaN / huge arguments.
__device__ __forceinline__ void gw_cispi(const double t, double& re, double& im) {
    const double r = t - 2.0 * rint(0.5 * t);          // [-1, 1]
    const double kq = rint(2.0 * r);                   // -2 .. 2
    const double y = r - 0.5 * kq, y2 = y * y;
    double sp = -2.1915353447830204e-05;
    sp = fma(sp, y2, 0.00046630280576761234);
    sp = fma(sp, y2, -0.007370430945714348);
    sp = fma(sp, y2, 0.08214588661112819);
    sp = fma(sp, y2, -0.5992645293207919);
    sp = fma(sp, y2, 2.550164039877345);
    sp = fma(sp, y2, -5.167712780049969);
    sp = fma(sp, y2, 3.141592653589793);
    sp *= y;
    double cp = 4.303069587032944e-06;
    cp = fma(cp, y2, -0.00010463810492484565);
    cp = fma(cp, y2, 0.001929574309403922);
    cp = fma(cp, y2, -0.02580689139001405);
    cp = fma(cp, y2, 0.23533063035889312);
    cp = fma(cp, y2, -1.3352627688545893);
    cp = fma(cp, y2, 4.058712126416768);
    cp = fma(cp, y2, -4.934802200544679);
    cp = fma(cp, y2, 1.0);
    // pi t = pi y + k pi / 2:  k = 0: (c, s); 1: (-s, c); 2: (-c, -s); -1: (s, -c); -2: (-c, -s)
    const int k = (int)kq & 3;
    const double c1 = (k & 1) ? sp : cp, s1 = (k & 1) ? cp : sp;
    re = (k == 1 || k == 2) ? -c1 : c1;
    im = (k == 2 || k == 3) ? -s1 : s1;
}

// ---------------------------------------------------------------------------------------
// gw_logl: lanes = bins, ONE sample at a time, loops specialised by region.  A 256-thread workgroup owns GWL_ITERS x 256
// consecutive bins and GWL_GROUP samples.  For each sample the chunk's region of phase and amplitude is a SCALAR decision (the
// chunk's frequency range against the sample's region boundaries), so the matching instantiation of the bin loop runs with
// just that region's constants -- few enough to stay in SGPRs for the whole loop (no scalar reloads, no waits) -- while the
// per-bin records arrive as coalesced vector loads (in-order counter: the next pass's loads are in flight during this one).
// The linear phase exp(-2 pi i f dt_d), with the antenna factor folded in, advances by one complex multiplication per pass.
// ---------------------------------------------------------------------------------------
constexpr int GWL_ITERS = 32;
constexpr int GWL_CHUNK = GWL_THREADS * GWL_ITERS;        // 8192 bins
constexpr int GWL_GROUP = 16;

typedef const __attribute__((address_space(4))) double* gw_const_dp;      // constant address space: scalar loads
typedef double gw_d4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) gw_d4v* gw_gd4p;             // global address space: global_load, in-order vmcnt
typedef const __attribute__((address_space(1))) double* gw_gdp;

template <int NIFO, bool PM, int PR, int AR, bool PLAIN>
__device__ __forceinline__ void gw_sample_chunk(const gw_const_dp sp, const gw_gd4p basis, const gw_gdp basis5, const gw_gd4p dat, const long nb, const long bin0, const long k0,
                                                 const double df, double& a_re, double& a_im, double& a_hh) {
    // the sample's record through the scalar cache, INSIDE the specialised loop's function: only the fields this instantiation
    // uses are loaded (the rest of the copy is dead code), few enough to stay in SGPRs for the whole loop
    gw::GwSource L;
    {
        constexpr int NSRC = (int)(sizeof(gw::GwSource) / sizeof(double));
        double* lp = reinterpret_cast<double*>(&L);
#pragma unroll
        for (int q = 0; q < NSRC; ++q) lp[q] = sp[q];
    }
    // E_d at this lane's first bin, and the step of GWL_THREADS bins
    double er[NIFO], ei[NIFO], rsr[NIFO], rsi[NIFO], ksq[NIFO];
    const double f_first = (double)(k0 + bin0 + threadIdx.x) * df;
#pragma unroll
    for (int k = 0; k < NIFO; ++k) {
        double cr, ci;
        gw_cispi(-2.0 * f_first * L.dt[k], cr, ci);
        er[k] = L.k_re[k] * cr - L.k_im[k] * ci;
        ei[k] = L.k_re[k] * ci + L.k_im[k] * cr;
        rsr[k] = L.rs_re[k]; rsi[k] = L.rs_im[k]; ksq[k] = L.k_sq[k];
    }
    // (two passes per trip: two independent dependency chains per lane; four were slower under the 128-VGPR cap)
#pragma unroll 2
    for (int j = 0; j < GWL_ITERS; ++j) {
        const long i = bin0 + (long)j * GWL_THREADS + threadIdx.x;
        if (i < nb) {
            const gw_d4v bs = basis[i];
            gw::GwBin bin;
            bin.f = (double)(k0 + i) * df;
            bin.f13 = bs[0]; bin.inv13 = bs[1]; bin.lnf13 = bs[2]; bin.fm76 = bs[3];
            bin.p578 = basis5[i];
            double amp, ph;
            gw::eval_bin_t<PR, AR, PLAIN>(L, bin, amp, ph);
            double cs, sn;
            gw_cispi(-ph, cs, sn);
            double qr = 0.0, qi = 0.0, hh = 0.0;
#pragma unroll
            for (int k = 0; k < NIFO; ++k) {
                const gw_d4v d = dat[(long)k * nb + i];
                qr += d[0] * er[k] + d[1] * ei[k];               // conj(w d) E
                qi += d[0] * ei[k] - d[1] * er[k];
                hh += d[2] * ksq[k];
            }
            a_re += amp * (cs * qr - sn * qi);
            if (PM) a_im += amp * (cs * qi + sn * qr);
            a_hh += amp * amp * hh;
        }
#pragma unroll
        for (int k = 0; k < NIFO; ++k) {
            const double t = er[k] * rsr[k] - ei[k] * rsi[k];
            ei[k] = er[k] * rsi[k] + ei[k] * rsr[k];
            er[k] = t;
        }
    }
}

template <int NIFO, bool PM>
// (4 waves per SIMD = 128 VGPRs: the kernel then spills ~50 registers outside its bin loops and is still the fastest form --
//  measured at config 5's shape: 29.2 ms against 30.1 ms with 3 waves / 168 VGPRs / 16 spilled and 36.0 ms with 2 / 188 / none)
#ifndef GWL_MIN_WAVES
#define GWL_MIN_WAVES 4
#endif
__global__ __launch_bounds__(GWL_THREADS, GWL_MIN_WAVES) void gw_logl_kernel(const GwDev* __restrict__ Pp, const gw::GwSource* __restrict__ src,
                                                               const long B, double* __restrict__ partial) {
    const GwDev& P = *Pp;
    __shared__ double red[GWL_GROUP][GWL_WAVES][3];
    const long n_groups = (B + GWL_GROUP - 1) / GWL_GROUP;
    const long g = (long)blockIdx.x % n_groups;       // sample group: the fast index (workgroups in flight share chunks)
    const long c = (long)blockIdx.x / n_groups;       // chunk
    const long b0 = g * GWL_GROUP;
    const long bin0 = c * GWL_CHUNK;
    const long nb = P.n_bins, k0 = P.k0;
    const double df = P.df;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (global address space: global_load, in-order vmcnt only)
    const gw_gd4p basis = (gw_gd4p)(uintptr_t)P.basis;
    const gw_gdp basis5 = (gw_gdp)(uintptr_t)P.basis5;
    const gw_gd4p dat = (gw_gd4p)(uintptr_t)P.dat;
    const long i_last = (bin0 + GWL_CHUNK < nb ? bin0 + GWL_CHUNK : nb) - 1;
    const double f_lo = (double)(k0 + bin0) * df, f_hi = (double)(k0 + i_last) * df;
    for (int s = 0; s < GWL_GROUP; ++s) {
        const long b = b0 + s < B ? b0 + s : B - 1;       // (a sample beyond the batch re-evaluates the last one; not stored)
        const gw_const_dp sp = (gw_const_dp)(uintptr_t)(src + b);
        const gw::GwSource* const so = nullptr;
#define GW_FIELD(name) sp[(reinterpret_cast<const char*>(&so->name) - reinterpret_cast<const char*>(so)) / 8]
        double a_re = 0.0, a_im = 0.0, a_hh = 0.0;
        if (GW_FIELD(valid) != 0.0) {
            const double fp1 = GW_FIELD(fp1), fp2 = GW_FIELD(fp2), fa1 = GW_FIELD(fa1), fa3 = GW_FIELD(fa3);
            const int pr = f_hi < fp1 ? 0 : ((f_lo >= fp1 && f_hi < fp2) ? 1 : -1);
            const int ar = f_hi < fa1 ? 0 : ((f_lo >= fa1 && f_hi < fa3) ? 1 : -1);
            const bool plain = (GW_FIELD(has_tides) == 0.0 || f_hi <= GW_FIELD(ft1)) && f_hi <= GW_FIELD(f_cut);
#undef GW_FIELD
            const int code = (pr >= 0 && ar >= 0 && plain) ? pr * 2 + ar : -1;
            if (code == 0) gw_sample_chunk<NIFO, PM, 0, 0, true>(sp, basis, basis5, dat, nb, bin0, k0, df, a_re, a_im, a_hh);
            else if (code == 1) gw_sample_chunk<NIFO, PM, 0, 1, true>(sp, basis, basis5, dat, nb, bin0, k0, df, a_re, a_im, a_hh);
            else if (code == 3) gw_sample_chunk<NIFO, PM, 1, 1, true>(sp, basis, basis5, dat, nb, bin0, k0, df, a_re, a_im, a_hh);
            else gw_sample_chunk<NIFO, PM, -1, -1, false>(sp, basis, basis5, dat, nb, bin0, k0, df, a_re, a_im, a_hh);
        }
        const double t0 = gw_wave_sum(a_re), t2 = gw_wave_sum(a_hh);
        double t1 = 0.0;
        if (PM) t1 = gw_wave_sum(a_im);
        if (lane == 63) { red[s][wave][0] = t0; red[s][wave][1] = t1; red[s][wave][2] = t2; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < GWL_GROUP * 3; t += GWL_THREADS) {
        const int s = t / 3, k = t - 3 * s;
        if (b0 + s < B) {
            double v = 0.0;
            for (int w = 0; w < GWL_WAVES; ++w) v += red[s][w][k];
            partial[((long)c * 3 + k) * B + b0 + s] = v;
        }
    }
}

// mode 0: log-likelihood ratio (floor for invalid / non-finite); mode 1: the three inner products parts[b][3]
// Distance marginalisation: log sum_j w_j exp(x(d_j)), x(d) = dh (ds / d) - hh (ds / d)^2 / 2 with dh = Re<d|h> (ln I0(dh ds / d) - ...
// with dh = |<d|h>| when the phase is marginalised too), <d|h> and <h|h> given at the distance ds the row was evaluated at
// (bilby/gw/likelihood/base.py: distance_marginalized_likelihood + _create_lookup_table, evaluated instead of tabulated).
__device__ inline double gw_distance_marginalised(const GwDev& P, const double dh, const double hh, const double ds) {
    // One pass with a running maximum (the sum is rescaled whenever a larger term appears), walked OUTWARD from the node nearest the
    // peak of x(d): with s = ds / d the exponent is dh s - hh s^2 / 2 + ln w (ln I0(y) <= y), a parabola in s peaking at s = dh / hh
    // plus the slowly varying prior weight, so once the bound has fallen 50 below the running maximum on one side every further node
    // on that side adds less than e^-50 of the sum -- of bilby's 10^4 nodes a few hundred carry the integral for a loud signal.
    const int n = P.n_dist;
    int j0 = n - 1;
    if (dh > 0.0 && hh > 0.0) {
        const double d_peak = ds * hh / dh;
        int lo = 0, hi = n;                                    // first node >= d_peak (the grid increases)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.dist_grid[mid] < d_peak) lo = mid + 1; else hi = mid; }
        j0 = lo < n ? lo : n - 1;
    }
    double mx = -dinf(), acc = 0.0;
    for (int dir = 0; dir < 2; ++dir) {
        int seen = 0;
        for (int j = dir == 0 ? j0 : j0 - 1; dir == 0 ? j < n : j >= 0; j += dir == 0 ? 1 : -1) {
            const double lw = P.dist_logw[j];
            if (!(lw > -dinf())) continue;
            const double sc = ds / P.dist_grid[j];
            const double bound = dh * sc - hh * sc * sc / 2.0 + lw;
            if (++seen > 8 && bound < mx - 50.0) break;
            const double x = P.phase_marg ? gw::ln_bessel_i0(dh * sc) - hh * sc * sc / 2.0 + lw : bound;
            if (x > mx) { acc = acc * exp(mx - x) + 1.0; mx = x; }
            else acc += exp(x - mx);
        }
    }
    return mx + log(acc);
}

__global__ __launch_bounds__(256) void gw_finish_kernel(const GwDev* __restrict__ Pp, const gw::GwSource* __restrict__ src,
                                                        const double* __restrict__ partial, const long B, const int mode,
                                                        const int n_chunks, double* __restrict__ out) {
    const GwDev& P = *Pp;
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double re = 0.0, im = 0.0, hh = 0.0;
    for (int c = 0; c < n_chunks; ++c) {
        re += partial[((long)c * 3 + 0) * B + b];
        im += partial[((long)c * 3 + 1) * B + b];
        hh += partial[((long)c * 3 + 2) * B + b];
    }
    re *= P.four_over_T; im *= P.four_over_T; hh *= P.four_over_T;
    const bool valid = src[b].valid != 0.0;
    if (mode == 1) {
        out[3 * b + 0] = valid ? re : dnan();
        out[3 * b + 1] = valid ? im : dnan();
        out[3 * b + 2] = valid ? hh : dnan();
        return;
    }
    double r;
    if (P.n_dist > 0) {
        r = gw_distance_marginalised(P, P.phase_marg ? sqrt(re * re + im * im) : re, hh, src[b].distance);
    } else if (P.phase_marg) r = gw::ln_bessel_i0(sqrt(re * re + im * im)) - hh / 2.0;      // bilby: ln_i0(abs(d_inner_h)) - optimal_snr_squared / 2
    else r = re - hh / 2.0;
    out[b] = (valid && isfinite(r)) ? r : NMMA_LOGL_FLOOR;                            // core/base.py:82, :181
}

template <int NIFO>
__global__ __launch_bounds__(256) void gw_strain_kernel(const GwDev* __restrict__ Pp, const gw::GwSource* __restrict__ src, const long B,
                                                        double2* __restrict__ strain) {
    const GwDev& P = *Pp;
    const long b = blockIdx.y;
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;     // bin of the full frequency array
    if (k >= P.n_freq) return;
    const gw::GwSource& S = src[b];
    const long i = k - P.k0;
    double amp = 0.0, ph = 0.0;
    gw::GwBin bin;
    bin.f = (double)k * P.df;
    const bool in_band = i >= 0 && i < P.n_bins && S.valid != 0.0;
    if (in_band) {
        const double4 bs = P.basis[i];
        bin.f13 = bs.x; bin.inv13 = bs.y; bin.lnf13 = bs.z; bin.fm76 = bs.w;
        bin.p578 = P.basis5[i];
        gw::eval_bin(S, bin, amp, ph);
    }
#pragma unroll
    for (int d = 0; d < NIFO; ++d) {
        double2 h = make_double2(0.0, 0.0);
        // (zero where the detector's own mask is zero, like bilby's get_detector_response)
        if (in_band && P.dat[(long)d * P.n_bins + i].z != 0.0) {
            double t = -(ph + 2.0 * bin.f * S.dt[d]);
            t -= 2.0 * rint(0.5 * t);
            double sn, cs;
            sincospi(t, &sn, &cs);
            h.x = amp * (S.k_re[d] * cs - S.k_im[d] * sn);
            h.y = amp * (S.k_re[d] * sn + S.k_im[d] * cs);
        }
        strain[((long)b * NIFO + d) * P.n_freq + k] = h;
    }
}

// =======================================================================================
// Time marginalisation (bilby/gw/likelihood/base.py: calculate_snrs + time_marginalized_likelihood)
// gw_integrand_kernel: I[b][k] = sum_ifo conj(d_k) h_k / S_k for k = 0 .. n_freq - 2 (zero outside the evaluated band), the
//   array whose forward FFT is <d|h> as a function of the coalescence-time shift j * duration / (n_freq - 1);
// gw_tm_fft_kernel / gw_tm_shift_kernel: its FFT, pruned to the shifts the time prior supports; gw_tm_term_kernel: the terms of the
//   time sum, one thread per (row, shift); gw_tm_logsum_kernel: log sum_j w_j exp(x_j), one workgroup per row.
// =======================================================================================
constexpr int GW_TM_ROWS = 16;      // rows per workgroup of gw_integrand_kernel: a bin's basis and data are loaded once for all of them
template <int NIFO>
__global__ __launch_bounds__(256) void gw_integrand_kernel(const GwDev* __restrict__ Pp, const gw::GwSource* __restrict__ src, const long b0,
                                                           const long nb, double2* __restrict__ out) {
    const GwDev& P = *Pp;
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long N = P.n_freq - 1;
    if (k >= N) return;
    const long i = k - P.k0;
    const bool in_band = i >= 0 && i < P.n_bins;
    gw::GwBin bin;
    bin.f = (double)k * P.df;
    double4 wd[NIFO];
    if (in_band) {
        const double4 bs = P.basis[i];
        bin.f13 = bs.x; bin.inv13 = bs.y; bin.lnf13 = bs.z; bin.fm76 = bs.w;
        bin.p578 = P.basis5[i];
#pragma unroll
        for (int d = 0; d < NIFO; ++d) wd[d] = P.dat[(long)d * P.n_bins + i];      // {w d_re, w d_im, w, 0}
    }
    for (int r = 0; r < GW_TM_ROWS; ++r) {
        const long bl = (long)blockIdx.y * GW_TM_ROWS + r;
        if (bl >= nb) break;
        double2 acc = make_double2(0.0, 0.0);
        const gw::GwSource& S = src[b0 + bl];
        if (in_band && S.valid != 0.0) {
            double amp = 0.0, ph = 0.0;
            gw::eval_bin(S, bin, amp, ph);
#pragma unroll
            for (int d = 0; d < NIFO; ++d) {
                double t = -(ph + 2.0 * bin.f * S.dt[d]);
                t -= 2.0 * rint(0.5 * t);
                double sn, cs;
                sincospi(t, &sn, &cs);
                const double hr = amp * (S.k_re[d] * cs - S.k_im[d] * sn), hi = amp * (S.k_re[d] * sn + S.k_im[d] * cs);
                acc.x += wd[d].x * hr + wd[d].y * hi;                        // conj(d) h w
                acc.y += wd[d].x * hi - wd[d].y * hr;
            }
        }
        out[bl * N + k] = acc;
    }
}

// First stage of the N = GW_TM_N1 x N2 decomposition (k = N2 k1 + k2, j = j1 + GW_TM_N1 j2): one workgroup per (k2, row) gathers
// the row's integrand at k = N2 k1 + k2, does the GW_TM_N1-point FFT over k1 in LDS (radix 2, decimation in time), applies the twiddle
// exp(-2 pi i j1 k2 / N) and stores G[row][k2][j1].  The second stage is pruned: gw_tm_shift_kernel sums over k2 only for the shifts
// j the time prior supports (a few hundred of the 2.6e5 of config 5).
constexpr int GW_TM_N1 = 1024;
constexpr int GW_TM_K2B = 4;        // residues k2 per workgroup: the gather then reads whole 64-byte sectors (one k2 alone reads 16 of every 64)
__global__ __launch_bounds__(256) void gw_tm_fft_kernel(const double2* __restrict__ I, const long N, const int N2, const int j1_first,
                                                       const int j1_count, double2* __restrict__ G) {
    __shared__ double2 x[GW_TM_K2B][GW_TM_N1];
    __shared__ double2 tw[GW_TM_N1 / 2];                  // exp(-2 pi i k / 1024), k < 512: every stage's twiddles (stride 1024 / len)
    for (int k = threadIdx.x; k < GW_TM_N1 / 2; k += 256) {
        double sn, cs;
        sincospi(-2.0 * (double)k / (double)GW_TM_N1, &sn, &cs);
        tw[k] = make_double2(cs, sn);
    }
    const int k2b = blockIdx.x * GW_TM_K2B;
    const long bl = blockIdx.y;
    const double2* Ib = I + bl * N;
    for (int idx = threadIdx.x; idx < GW_TM_K2B * GW_TM_N1; idx += 256) {
        const int q = idx % GW_TM_K2B, t = idx / GW_TM_K2B;
        x[q][__brev((unsigned)t) >> 22] = (k2b + q < N2) ? Ib[(long)N2 * t + k2b + q] : make_double2(0.0, 0.0);      // bit-reversed (10 bits)
    }
    __syncthreads();
    for (int len = 2; len <= GW_TM_N1; len <<= 1) {
        const int half = len >> 1;
        for (int idx = threadIdx.x; idx < GW_TM_K2B * (GW_TM_N1 / 2); idx += 256) {
            const int q = idx / (GW_TM_N1 / 2), t = idx - q * (GW_TM_N1 / 2);
            const int grp = t / half, pos = t - grp * half;
            const int i0 = grp * len + pos, i1 = i0 + half;
            const double2 e = tw[pos * (GW_TM_N1 / len)];
            const double cs = e.x, sn = e.y;
            const double2 a = x[q][i0], c = x[q][i1];
            const double2 w = make_double2(c.x * cs - c.y * sn, c.x * sn + c.y * cs);
            x[q][i0] = make_double2(a.x + w.x, a.y + w.y);
            x[q][i1] = make_double2(a.x - w.x, a.y - w.y);
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < GW_TM_K2B * GW_TM_N1; idx += 256) {
        const int q = idx / GW_TM_N1, j1 = idx - q * GW_TM_N1;
        const int k2 = k2b + q;
        if (k2 >= N2) continue;
        // (only the residues j1 = j mod 1024 of the shifts the time prior supports are read by the second stage)
        if (((j1 - j1_first) & (GW_TM_N1 - 1)) >= j1_count) continue;
        double sn, cs;
        sincospi(-2.0 * ((double)j1 * (double)k2) / (double)N, &sn, &cs);
        const double2 v = x[q][j1];
        G[(bl * N2 + k2) * GW_TM_N1 + j1] = make_double2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
    }
}

// Second, pruned stage: F_j = 4/T sum_k2 G[k2][j1] exp(-2 pi i j2 k2 / N2) for the shifts j the time prior supports only,
// stored per row as (re, im) in slot j - tm_lo.
__global__ __launch_bounds__(256) void gw_tm_shift_kernel(const GwDev* __restrict__ Pp, const double2* __restrict__ G, const long b0, const int N2,
                                                         double2* __restrict__ Fs) {
    const GwDev& P = *Pp;
    const long n_sup = P.tm_hi - P.tm_lo;
    const double2* Gb = G + (long)blockIdx.x * N2 * GW_TM_N1;
    for (long slot = threadIdx.x; slot < n_sup; slot += 256) {
        const long j = P.tm_lo + slot;
        const int j1 = (int)(j & (GW_TM_N1 - 1));
        const long j2 = j >> 10;
        double re = 0.0, im = 0.0;
        for (int k2 = 0; k2 < N2; ++k2) {
            double sn, cs;
            sincospi(-2.0 * (double)((j2 * k2) % N2) / (double)N2, &sn, &cs);
            const double2 g = Gb[(long)k2 * GW_TM_N1 + j1];
            re += g.x * cs - g.y * sn;
            im += g.x * sn + g.y * cs;
        }
        Fs[(b0 + blockIdx.x) * n_sup + slot] = make_double2(re * P.four_over_T, im * P.four_over_T);
    }
}

// x[b][slot] = the (distance-marginalised) log-likelihood ratio at shift slot + ln w_slot: one thread per (row, shift) -- with the
// distance marginalised too this is the expensive step (10^4 distance nodes per pair), spread over the whole batch at once.
__global__ __launch_bounds__(256) void gw_tm_term_kernel(const GwDev* __restrict__ Pp, const gw::GwSource* __restrict__ src,
                                                        const double2* __restrict__ Fs, const double* __restrict__ parts, const long B,
                                                        double* __restrict__ x) {
    const GwDev& P = *Pp;
    const long n_sup = P.tm_hi - P.tm_lo;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n_sup) return;
    const long b = idx / n_sup, slot = idx - b * n_sup;
    double lw = P.time_logw[P.tm_lo + slot];
    if (P.tm_jitter) {
        // bilby: time_prior_array = prior.prob(times + time_jitter) * delta_tc -- the node's own weight, or its neighbour's where the
        // shifted time enters the prior's support, and nothing where it leaves it
        const long j = P.tm_lo + slot;
        const double t = P.start_time + (double)j * P.tm_dt + src[b].jitter;
        if (!(lw > -dinf())) {
            const long N = P.n_freq - 1;
            const double ln = j > 0 ? P.time_logw[j - 1] : -dinf(), lp = j + 1 < N ? P.time_logw[j + 1] : -dinf();
            lw = ln > -dinf() ? ln : lp;
        }
        if (!(t >= P.tm_min && t <= P.tm_max)) lw = -dinf();
    }
    double v = -dinf();
    if (lw > -dinf()) {
        const double2 f = Fs[idx];
        const double hh = parts[3 * b + 2];
        const double dh = P.phase_marg ? sqrt(f.x * f.x + f.y * f.y) : f.x;
        if (P.n_dist > 0) v = gw_distance_marginalised(P, dh, hh, src[b].distance) + lw;
        else v = (P.phase_marg ? gw::ln_bessel_i0(dh) : dh) - hh / 2.0 + lw;
    }
    x[idx] = v;
}

// log sum over the shifts, one workgroup per row
__global__ __launch_bounds__(256) void gw_tm_logsum_kernel(const GwDev* __restrict__ Pp, const gw::GwSource* __restrict__ src,
                                                          const double* __restrict__ x, double* __restrict__ out) {
    const GwDev& P = *Pp;
    const long n_sup = P.tm_hi - P.tm_lo;
    const long b = blockIdx.x;
    const double* xb = x + b * n_sup;
    __shared__ double red[256];
    double mx = -dinf();
    for (int pass = 0; pass < 2; ++pass) {
        double acc = 0.0, m = -dinf();
        for (long slot = threadIdx.x; slot < n_sup; slot += 256) {
            const double v = xb[slot];
            if (!(v > -dinf())) continue;
            if (pass == 0) m = v > m ? v : m; else acc += exp(v - mx);
        }
        red[threadIdx.x] = pass == 0 ? m : acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] = pass == 0 ? (red[threadIdx.x] > red[threadIdx.x + st] ? red[threadIdx.x] : red[threadIdx.x + st])
                                                                   : red[threadIdx.x] + red[threadIdx.x + st];
            __syncthreads();
        }
        if (pass == 0) mx = red[0];
        else if (threadIdx.x == 0) {
            const double r = mx + log(red[0]);
            out[b] = (src[b].valid != 0.0 && isfinite(r)) ? r : NMMA_LOGL_FLOOR;
        }
        __syncthreads();
    }
}

// =======================================================================================
// sum over messengers + floor (MultiMessengerLikelihood.sub_log_likelihood, joint/joint_likelihood.py:62-67)
// =======================================================================================
struct LoglParts {
    const double* p[8];
    int n;
};
__global__ __launch_bounds__(256) void logl_sum_floor_kernel(const LoglParts parts, const long B, double* __restrict__ out) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double total = 0.0;
    for (int k = 0; k < parts.n; ++k) total += parts.p[k][b];      // messenger order, like Python's sum()
    // a messenger's own floor (-1.797e308) plus anything stays at or below the floor (two floors overflow to -inf):
    // `logl if np.isfinite(logl) else nan_to_num(-inf)`, and a floored messenger floors the sum
    out[b] = (isfinite(total) && total > NMMA_LOGL_FLOOR) ? total : NMMA_LOGL_FLOOR;
}

}  // namespace nmma

// =======================================================================================
// host API
// =======================================================================================
using nmma::fail;

#define GW_HIP(call)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (call);                                                                                   \
        if (_e != hipSuccess) return fail(std::string(#call) + " failed: " + hipGetErrorString(_e));              \
    } while (0)

struct nmma_gw_handle {
    nmma::GwDev dev{};
    nmma::GwDev* dev_d = nullptr;
    int device = 0;
    double noise_logl = 0.0;
    std::vector<void*> owned;
    nmma::gw::GwSource* src = nullptr;
    double* partial = nullptr;
    int64_t cap = 0;
    bool prof_on = false;
    std::vector<hipEvent_t> ev;
    int prof_max = 0;
    // time marginalisation: the per-bin integrand of tm_batch rows and its first FFT stage; <d|h>, <h|h> per row
    int64_t tm_batch = 0, tm_parts_cap = 0;
    double2* tm_buf = nullptr;      // [2][tm_batch][n_freq - 1]: integrand | first FFT stage
    double* tm_parts = nullptr;
    double2* tm_F = nullptr;        // [rows][shifts in the prior's support] <d|h>(t_j)
    double* tm_x = nullptr;         // [rows][shifts] terms of the time sum
};

extern "C" {

int32_t nmma_gw_loglike_ratio(const double* strain_dev, const double* data_dev, const double* weight_dev, int64_t B,
                              int32_t n_ifo, int64_t n_freq, double duration, double* out_dev, int32_t device, void* stream) {
    using namespace nmma;
    if (!strain_dev || !data_dev || !weight_dev || !out_dev || B < 0 || n_ifo < 1 || n_freq < 1 || !(duration > 0))
        return fail("nmma_gw_loglike_ratio: bad argument");
    if (B == 0) return 0;
    GW_HIP(hipSetDevice(device));
#define NM_GW(S)                                                                                                         \
    hipLaunchKernelGGL(gw_loglike_ratio_kernel<S>, dim3((unsigned)((B + S - 1) / S)), dim3(GW_THREADS), 0,                \
                       static_cast<hipStream_t>(stream), reinterpret_cast<const double2*>(strain_dev),                    \
                       reinterpret_cast<const double2*>(data_dev), weight_dev, (long)B, (long)n_ifo * (long)n_freq,       \
                       4.0 / duration, out_dev)
    // samples per workgroup: as many as leave one workgroup per CU (the data / weight arrays are then read once per group)
    if (B >= 8 * 256) NM_GW(8);
    else if (B >= 4 * 256) NM_GW(4);
    else if (B >= 2 * 256) NM_GW(2);
    else NM_GW(1);
#undef NM_GW
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_gw_loglike_ratio launch failed: ") + hipGetErrorString(e));
    return 0;
}

int32_t nmma_logl_sum_floor(const double* const* parts_dev, int32_t n_parts, int64_t B, double* out_dev, int32_t device, void* stream) {
    using namespace nmma;
    if (!parts_dev || n_parts < 1 || n_parts > 8 || B < 0 || !out_dev) return fail("nmma_logl_sum_floor: bad argument (1..8 messengers)");
    if (B == 0) return 0;
    LoglParts parts{};
    parts.n = n_parts;
    for (int k = 0; k < n_parts; ++k) {
        if (!parts_dev[k]) return fail("nmma_logl_sum_floor: null messenger array");
        parts.p[k] = parts_dev[k];
    }
    GW_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(logl_sum_floor_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), parts, (long)B, out_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_logl_sum_floor launch failed: ") + hipGetErrorString(e));
    return 0;
}

void nmma_gw_destroy(nmma_gw_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->src) (void)hipFree(h->src);
    if (h->partial) (void)hipFree(h->partial);
    if (h->tm_buf) (void)hipFree(h->tm_buf);
    if (h->tm_parts) (void)hipFree(h->tm_parts);
    if (h->tm_F) (void)hipFree(h->tm_F);
    if (h->tm_x) (void)hipFree(h->tm_x);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    delete h;
}

int32_t nmma_gw_create(const nmma_gw_config* c, nmma_gw_handle** out) {
    using namespace nmma;
    if (!c || !out) return fail("nmma_gw_create: null argument");
    *out = nullptr;
    if (c->abi_version != NMMA_ABI_VERSION) return fail("nmma_gw_create: ABI version mismatch");
    if (c->n_ifo < 1 || c->n_ifo > NMMA_GW_MAX_IFO) return fail("nmma_gw_create: n_ifo must be 1..4");
    if (c->n_freq < 2 || !(c->duration > 0)) return fail("nmma_gw_create: bad frequency array");
    if (!c->data || !c->psd || !c->mask || !c->detector_tensor || !c->vertex) return fail("nmma_gw_create: null array");
    if (!(c->reference_frequency > 0)) return fail("nmma_gw_create: reference_frequency must be positive");
    if (c->mass_mode != NMMA_GW_CHIRP_MASS_RATIO && c->mass_mode != NMMA_GW_COMPONENT_MASSES) return fail("nmma_gw_create: bad mass_mode");
    const nmma_slot* slots[13] = {&c->mass_a, &c->mass_b, &c->chi_1, &c->chi_2, &c->lambda_1, &c->lambda_2, &c->luminosity_distance,
                                  &c->theta_jn, &c->phase, &c->ra, &c->dec, &c->psi, &c->geocent_time};
    for (const nmma_slot* s : slots)
        if (s->col >= c->n_dim || s->op < 0 || s->op > NMMA_OP_ACOS) return fail("nmma_gw_create: parameter slot out of range");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail("nmma_gw_create: no HIP device (nmma_amd has no CPU fallback)");
    if (c->device < 0 || c->device >= n_dev) return fail("nmma_gw_create: device ordinal out of range");
    GW_HIP(hipSetDevice(c->device));
    const int n_ifo = c->n_ifo;
    const int64_t NF = c->n_freq;
    const double df = 1.0 / c->duration;
    // evaluated band: union of the detectors' masks inside the waveform's own band (source.py: frequency_bounds), f > 0
    const double fmax_w = c->waveform_maximum_frequency > 0 ? c->waveform_maximum_frequency : HUGE_VAL;
    int64_t k_lo = NF, k_hi = -1;
    for (int d = 0; d < n_ifo; ++d)
        for (int64_t k = 1; k < NF; ++k) {
            const double f = (double)k * df;
            if (c->mask[(size_t)d * NF + k] && f >= c->waveform_minimum_frequency && f <= fmax_w) {
                k_lo = std::min(k_lo, k);
                k_hi = std::max(k_hi, k);
            }
        }
    if (k_hi < k_lo) return fail("nmma_gw_create: the frequency masks and the waveform band do not overlap");
    const int64_t nb = k_hi - k_lo + 1;
    nmma_gw_handle* h = new nmma_gw_handle();
    h->device = c->device;
    GwDev& P = h->dev;
    P.n_ifo = n_ifo; P.tidal = c->tidal ? 1 : 0; P.mass_mode = c->mass_mode; P.phase_marg = c->phase_marginalization ? 1 : 0;
    P.n_bins = nb; P.k0 = k_lo; P.n_freq = NF; P.n_dim = c->n_dim;
    P.n_chunks = (int32_t)((nb + GWL_CHUNK - 1) / GWL_CHUNK);
    P.df = df; P.f_ref = c->reference_frequency; P.start_time = c->start_time;
    P.gmst_ref_time = c->gmst_ref_time; P.gmst_ref = c->gmst_ref; P.gmst_rate = c->gmst_rate;
    P.four_over_T = 4.0 / c->duration;
    for (int d = 0; d < n_ifo; ++d) {
        std::memcpy(P.det[d].tensor, c->detector_tensor + 9 * d, 9 * sizeof(double));
        std::memcpy(P.det[d].vertex, c->vertex + 3 * d, 3 * sizeof(double));
    }
    P.mass_a = c->mass_a; P.mass_b = c->mass_b; P.chi_1 = c->chi_1; P.chi_2 = c->chi_2; P.lambda_1 = c->lambda_1; P.lambda_2 = c->lambda_2;
    P.luminosity_distance = c->luminosity_distance; P.theta_jn = c->theta_jn; P.phase = c->phase; P.ra = c->ra; P.dec = c->dec;
    P.psi = c->psi; P.geocent_time = c->geocent_time;
    std::vector<double> basis((size_t)4 * nb), basis5((size_t)nb), dat((size_t)4 * nb * n_ifo, 0.0);
    for (int64_t i = 0; i < nb; ++i) {
        const gw::GwBin b = gw::make_bin((double)(k_lo + i) * df);
        basis[4 * i] = b.f13; basis[4 * i + 1] = b.inv13; basis[4 * i + 2] = b.lnf13; basis[4 * i + 3] = b.fm76;
        basis5[i] = b.p578;
    }
    double noise = 0.0;
    for (int d = 0; d < n_ifo; ++d) {
        double acc = 0.0;
        for (int64_t k = 0; k < NF; ++k) {
            if (!c->mask[(size_t)d * NF + k]) continue;
            const double S = c->psd[(size_t)d * NF + k];
            if (!(S > 0) || !std::isfinite(S)) { delete h; return fail("nmma_gw_create: the PSD must be positive and finite inside the frequency mask"); }
            const double re = c->data[2 * ((size_t)d * NF + k)], im = c->data[2 * ((size_t)d * NF + k) + 1];
            acc += (re * re + im * im) / S;
            const double f = (double)k * df;
            if (k >= k_lo && k <= k_hi && f >= c->waveform_minimum_frequency && f <= fmax_w) {
                const size_t o = 4 * ((size_t)d * nb + (k - k_lo));
                dat[o] = re / S; dat[o + 1] = im / S; dat[o + 2] = 1.0 / S;
            }
        }
        noise -= 0.5 * (4.0 / c->duration) * acc;       // bilby: noise_log_likelihood = -sum_ifo <d|d> / 2
    }
    h->noise_logl = noise;
    auto up = [&](const void* src, size_t bytes, void** dst) -> hipError_t {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        h->owned.push_back(p);
        e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
        *dst = p;
        return e;
    };
    void* p = nullptr;
    hipError_t e = up(basis.data(), basis.size() * 8, &p);
    if (e == hipSuccess) { P.basis = reinterpret_cast<const double4*>(p); e = up(basis5.data(), basis5.size() * 8, &p); }
    if (e == hipSuccess) { P.basis5 = reinterpret_cast<const double*>(p); e = up(dat.data(), dat.size() * 8, &p); }
    if (e == hipSuccess) P.dat = reinterpret_cast<const double4*>(p);
    P.n_dist = 0; P.dist_grid = nullptr; P.dist_logw = nullptr;
    if (e == hipSuccess && c->n_distance > 0) {        // distance marginalisation: the grid and ln(prior x step) per node
        if (!c->distance_grid || !c->distance_log_weight) { nmma_gw_destroy(h); return fail("nmma_gw_create: null distance grid"); }
        for (int j = 0; j < c->n_distance; ++j)
            if (!(c->distance_grid[j] > 0)) { nmma_gw_destroy(h); return fail("nmma_gw_create: distance grid must be positive"); }
        e = up(c->distance_grid, (size_t)c->n_distance * 8, &p);
        if (e == hipSuccess) { P.dist_grid = reinterpret_cast<const double*>(p); e = up(c->distance_log_weight, (size_t)c->n_distance * 8, &p); }
        if (e == hipSuccess) { P.dist_logw = reinterpret_cast<const double*>(p); P.n_dist = c->n_distance; }
    }
    P.time_logw = nullptr; P.tm_lo = 0; P.tm_hi = 0; P.tm_jitter = 0; P.tm_min = 0.0; P.tm_max = 0.0; P.tm_dt = 0.0;
    P.time_jitter.col = -1; P.time_jitter.op = 0; P.time_jitter.value = 0.0;
    if (e == hipSuccess && c->time_log_weight != nullptr) {      // time marginalisation
        const int64_t N = NF - 1;
        if (N % GW_TM_N1 != 0) { nmma_gw_destroy(h); return fail("nmma_gw_create: time marginalisation needs n_freq - 1 to be a multiple of 1024"); }
        int64_t lo = N, hi = 0;
        for (int64_t j = 0; j < N; ++j)
            if (c->time_log_weight[j] > -HUGE_VAL) { lo = std::min(lo, j); hi = std::max(hi, j + 1); }
        if (hi <= lo) { nmma_gw_destroy(h); return fail("nmma_gw_create: the time prior has no support inside the data segment"); }
        P.tm_jitter = (c->time_jitter.col >= 0 || c->time_jitter.value != 0.0) ? 1 : 0;
        P.time_jitter = c->time_jitter;
        P.tm_min = c->time_prior_minimum; P.tm_max = c->time_prior_maximum; P.tm_dt = c->duration / (double)N;
        if (P.tm_jitter) {
            if (c->time_jitter.col >= c->n_dim || !(P.tm_max > P.tm_min)) { nmma_gw_destroy(h); return fail("nmma_gw_create: bad time_jitter slot or time prior bounds"); }
            lo = std::max<int64_t>(0, lo - 1); hi = std::min<int64_t>(N, hi + 1);
        }
        e = up(c->time_log_weight, (size_t)N * 8, &p);
        if (e == hipSuccess) { P.time_logw = reinterpret_cast<const double*>(p); P.tm_lo = lo; P.tm_hi = hi; }
    }
    if (e == hipSuccess) e = up(&P, sizeof(P), &p);
    if (e != hipSuccess) { nmma_gw_destroy(h); return fail(std::string("nmma_gw_create: ") + hipGetErrorString(e)); }
    h->dev_d = reinterpret_cast<GwDev*>(p);
    *out = h;
    return 0;
}

double nmma_gw_noise_log_likelihood(const nmma_gw_handle* h) { return h ? h->noise_logl : 0.0; }
int64_t nmma_gw_n_bins(const nmma_gw_handle* h) { return h ? h->dev.n_bins : 0; }

static int32_t gw_reserve(nmma_gw_handle* h, int64_t B) {
    if (B <= h->cap) return 0;
    // (outside any capture; the previous buffers may still be in use by launches in flight on the caller's stream)
    GW_HIP(hipDeviceSynchronize());
    if (h->src) (void)hipFree(h->src);
    if (h->partial) (void)hipFree(h->partial);
    h->src = nullptr; h->partial = nullptr; h->cap = 0;
    const int64_t cap = std::max<int64_t>(B, 256);
    GW_HIP(hipMalloc(reinterpret_cast<void**>(&h->src), (size_t)cap * sizeof(nmma::gw::GwSource)));
    GW_HIP(hipMalloc(reinterpret_cast<void**>(&h->partial), (size_t)cap * h->dev.n_chunks * 3 * sizeof(double)));
    h->cap = cap;
    return 0;
}

static int32_t gw_run(nmma_gw_handle* h, const double* theta_dev, int64_t B, int64_t ld, double* out_dev, void* stream, int mode,
                      const char* what) {
    using namespace nmma;
    if (!h || !theta_dev || !out_dev || B < 0 || ld < h->dev.n_dim) return fail(std::string(what) + ": bad argument");
    if (B == 0) return 0;
    GW_HIP(hipSetDevice(h->device));
    if (gw_reserve(h, B) != 0) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const GwDev& P = h->dev;
    hipLaunchKernelGGL(gw_source_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->dev_d, theta_dev, (long)B, (long)ld,
                       (double)GWL_THREADS * h->dev.df, h->src);      // (the step of a lane's bin loop)
    const dim3 grid((unsigned)(((B + GWL_GROUP - 1) / GWL_GROUP) * P.n_chunks));
    const bool pm = P.phase_marg != 0 || mode == 1;
    const bool prof = h->prof_on && (int)h->ev.size() + 2 <= 2 * h->prof_max;
    if (prof) {
        hipEvent_t a, b;
        GW_HIP(hipEventCreate(&a)); GW_HIP(hipEventCreate(&b));
        h->ev.push_back(a); h->ev.push_back(b);
        GW_HIP(hipEventRecord(a, s));
    }
#define GW_LAUNCH(N)                                                                                                                   \
    do {                                                                                                                               \
        if (pm) hipLaunchKernelGGL((gw_logl_kernel<N, true>), grid, dim3(GWL_THREADS), 0, s, h->dev_d, h->src, (long)B, h->partial);   \
        else hipLaunchKernelGGL((gw_logl_kernel<N, false>), grid, dim3(GWL_THREADS), 0, s, h->dev_d, h->src, (long)B, h->partial);     \
    } while (0)
    switch (P.n_ifo) {
        case 1: GW_LAUNCH(1); break;
        case 2: GW_LAUNCH(2); break;
        case 3: GW_LAUNCH(3); break;
        default: GW_LAUNCH(4); break;
    }
#undef GW_LAUNCH
    if (prof) GW_HIP(hipEventRecord(h->ev.back(), s));
    const bool tm = P.time_logw != nullptr && mode == 0;
    if (tm) {
        // time marginalisation: <h|h> from the fused kernel (mode 1 of the finish step), then per chunk of rows the per-bin
        // integrand, the first stage of its FFT over the coalescence-time shifts and the weighted log-sum over the prior's support
        const int64_t N = P.n_freq - 1;
        if (h->tm_parts_cap < B) {
            GW_HIP(hipDeviceSynchronize());
            if (h->tm_parts) (void)hipFree(h->tm_parts);
            if (h->tm_F) (void)hipFree(h->tm_F);
            if (h->tm_x) (void)hipFree(h->tm_x);
            h->tm_parts = nullptr; h->tm_F = nullptr; h->tm_x = nullptr; h->tm_parts_cap = 0;
            const int64_t cap = std::max<int64_t>(B, 256), n_sup = P.tm_hi - P.tm_lo;
            GW_HIP(hipMalloc(reinterpret_cast<void**>(&h->tm_parts), (size_t)cap * 3 * sizeof(double)));
            GW_HIP(hipMalloc(reinterpret_cast<void**>(&h->tm_F), (size_t)cap * n_sup * sizeof(double2)));
            GW_HIP(hipMalloc(reinterpret_cast<void**>(&h->tm_x), (size_t)cap * n_sup * sizeof(double)));
            h->tm_parts_cap = cap;
        }
        const int N2 = (int)(N / GW_TM_N1);
        if (h->tm_buf == nullptr) {
            const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(1024, ((int64_t)1 << 30) / (N * 16)));      // 2 x <= 1 GiB (of 288 GB)
            GW_HIP(hipMalloc(reinterpret_cast<void**>(&h->tm_buf), (size_t)2 * batch * N * sizeof(double2)));
            h->tm_batch = batch;
        }
        double2* const integrand = h->tm_buf;
        double2* const stage1 = h->tm_buf + h->tm_batch * N;
        hipLaunchKernelGGL(gw_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, h->dev_d, h->src, h->partial, (long)B, 1,
                           P.n_chunks, h->tm_parts);
        for (int64_t b0 = 0; b0 < B; b0 += h->tm_batch) {
            const int64_t nb = std::min<int64_t>(h->tm_batch, B - b0);
            const dim3 g((unsigned)((N + 255) / 256), (unsigned)((nb + GW_TM_ROWS - 1) / GW_TM_ROWS));
            switch (P.n_ifo) {
                case 1: hipLaunchKernelGGL(gw_integrand_kernel<1>, g, dim3(256), 0, s, h->dev_d, h->src, (long)b0, (long)nb, integrand); break;
                case 2: hipLaunchKernelGGL(gw_integrand_kernel<2>, g, dim3(256), 0, s, h->dev_d, h->src, (long)b0, (long)nb, integrand); break;
                case 3: hipLaunchKernelGGL(gw_integrand_kernel<3>, g, dim3(256), 0, s, h->dev_d, h->src, (long)b0, (long)nb, integrand); break;
                default: hipLaunchKernelGGL(gw_integrand_kernel<4>, g, dim3(256), 0, s, h->dev_d, h->src, (long)b0, (long)nb, integrand); break;
            }
            const int64_t n_sup = P.tm_hi - P.tm_lo;
            hipLaunchKernelGGL(gw_tm_fft_kernel, dim3((unsigned)((N2 + GW_TM_K2B - 1) / GW_TM_K2B), (unsigned)nb), dim3(256), 0, s, integrand, (long)N, N2,
                               (int)(P.tm_lo & (GW_TM_N1 - 1)), (int)std::min<int64_t>(n_sup, GW_TM_N1), stage1);
            hipLaunchKernelGGL(gw_tm_shift_kernel, dim3((unsigned)nb), dim3(256), 0, s, h->dev_d, stage1, (long)b0, N2, h->tm_F);
        }
        const int64_t pairs = B * (P.tm_hi - P.tm_lo);
        hipLaunchKernelGGL(gw_tm_term_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, h->dev_d, h->src, h->tm_F, h->tm_parts, (long)B,
                           h->tm_x);
        hipLaunchKernelGGL(gw_tm_logsum_kernel, dim3((unsigned)B), dim3(256), 0, s, h->dev_d, h->src, h->tm_x, out_dev);
    } else {
        hipLaunchKernelGGL(gw_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, h->dev_d, h->src, h->partial, (long)B, mode,
                           P.n_chunks, out_dev);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string(what) + " launch failed: " + hipGetErrorString(e));
    return 0;
}

int32_t nmma_gw_loglike(nmma_gw_handle* h, const double* theta_dev, int64_t B, int64_t ld, double* out_dev, void* stream) {
    return gw_run(h, theta_dev, B, ld, out_dev, stream, 0, "nmma_gw_loglike");
}

int32_t nmma_gw_inner_products(nmma_gw_handle* h, const double* theta_dev, int64_t B, int64_t ld, double* parts_dev, void* stream) {
    return gw_run(h, theta_dev, B, ld, parts_dev, stream, 1, "nmma_gw_inner_products");
}

int32_t nmma_gw_strain(nmma_gw_handle* h, const double* theta_dev, int64_t B, int64_t ld, double* strain_dev, void* stream) {
    using namespace nmma;
    if (!h || !theta_dev || !strain_dev || B < 0 || B > 65535 || ld < h->dev.n_dim) return fail("nmma_gw_strain: bad argument");
    if (B == 0) return 0;
    GW_HIP(hipSetDevice(h->device));
    if (gw_reserve(h, B) != 0) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const GwDev& P = h->dev;
    hipLaunchKernelGGL(gw_source_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->dev_d, theta_dev, (long)B, (long)ld, h->dev.df, h->src);
    const dim3 grid((unsigned)((P.n_freq + 255) / 256), (unsigned)B);
    double2* o = reinterpret_cast<double2*>(strain_dev);
    switch (P.n_ifo) {
        case 1: hipLaunchKernelGGL(gw_strain_kernel<1>, grid, dim3(256), 0, s, h->dev_d, h->src, (long)B, o); break;
        case 2: hipLaunchKernelGGL(gw_strain_kernel<2>, grid, dim3(256), 0, s, h->dev_d, h->src, (long)B, o); break;
        case 3: hipLaunchKernelGGL(gw_strain_kernel<3>, grid, dim3(256), 0, s, h->dev_d, h->src, (long)B, o); break;
        default: hipLaunchKernelGGL(gw_strain_kernel<4>, grid, dim3(256), 0, s, h->dev_d, h->src, (long)B, o); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_gw_strain launch failed: ") + hipGetErrorString(e));
    return 0;
}

int32_t nmma_gw_profile_begin(nmma_gw_handle* h, int32_t max_launches) {
    if (!h || max_launches < 1) return fail("nmma_gw_profile_begin: bad argument");
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear();
    h->prof_on = true;
    h->prof_max = max_launches;
    return 0;
}

int32_t nmma_gw_profile_end(nmma_gw_handle* h, double* kernel_ms_total, int32_t* n_launches) {
    if (!h || !kernel_ms_total || !n_launches) return fail("nmma_gw_profile_end: bad argument");
    GW_HIP(hipSetDevice(h->device));
    GW_HIP(hipDeviceSynchronize());
    double total = 0.0;
    for (size_t i = 0; i + 1 < h->ev.size(); i += 2) {
        float ms = 0.f;
        GW_HIP(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        total += ms;
    }
    *kernel_ms_total = total;
    *n_launches = (int32_t)(h->ev.size() / 2);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear();
    h->prof_on = false;
    return 0;
}

}  // extern "C"
