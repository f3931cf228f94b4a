// em_kernels.hip -- gfx950 kernels of the batched EM light-curve log-likelihood.
//
// One launch of em_fused<R, WPB, KP> evaluates, for a tile of TS = 16*R parameter vectors
// and ONE observed filter (blockIdx.y), the whole per-filter chain of the reference:
//
//   x = (theta - pmin)/(pmax - pmin)              lightcurve_generation.py:193-194
//   c = Dense(relu)(x) -> Dense                   lightcurve_generation.py:198 (Keras fp32)
//   mag = (VA[:, :NC] @ c) * (maxs - mins) + mins lightcurve_generation.py:214-216 (fp64)
//   stage-1 lerp onto sample_times, +inf outside  lightcurve_generation.py:177 -> utils.py:642-645
//   t_obs = t*(1+z)+timeshift, app = mag+ext+distmod-2.5log10(1+z)   model.py:374, :381-404
//   stage-2 lerp onto the data epochs, +inf outside                   em_likelihood.py:313-335
//   sum of truncated-Gaussian / logsf terms                          em_likelihood.py:224-256
//
// Phase A (the FLOPs): both Dense layers on the f32 MFMA pipe, chained without a
// transpose: layer 1 produces H^T[hidden 16 x sample 16] whose accumulator registers ARE
// the B operands of layer 2 (C^T[coef 16 x sample 16] += W2^T[coef x 4 hidden] H^T).
// Each of the WPB waves owns NH/WPB hidden units and streams its pre-swizzled weight
// records straight from L2 into VGPRs (no LDS: nothing is shared between waves).
// Phase B (fp64 VALU): reconstruction of the NT-point light curve into LDS, then lane
// groups walk the ragged data of the filter (binary search on the redshifted grid).
//
// em_combine adds the per-filter partial sums in the reference's order and applies the
// floor (core/base.py:82, :180-181).
#include <hip/hip_runtime.h>

#include "em_device.h"
#include "em_math.h"

namespace nmma {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Pointers read out of the EmDev record have no provable address space; tell the
// compiler they are global so it emits global_load (vmcnt only) instead of flat_load.
typedef const __attribute__((address_space(1))) float* gcf32p;
typedef const __attribute__((address_space(1))) f32x4* gcf32x4p;
__device__ __forceinline__ gcf32p as_global(const float* p) { return (gcf32p)(uintptr_t)p; }

// relu on an MFMA result: one v_max (fmaxf would add a canonicalising v_max first)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// Opaque identity: stops InstCombine from folding phi(load, load) into load(phi(addr)),
// which would move every prefetched weight load back to its use (no latency hiding).
__device__ __forceinline__ void opaque(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque(float& v) { asm volatile("" : "+v"(v)); }

// Byte offsets of the LDS carve-up (computed by the host with lds_layout()).
struct LdsOff {
    int32_t part, cd, xs, praw, scal, sysv, mag, est, total;
    int32_t SB;        // samples per reconstruction sub-batch
    int32_t nf_max;    // widest averaged filter (est buffer row length), 0 if none
};

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }

__host__ inline LdsOff lds_layout(int R, int WPB, int NC, int NT, int kmax, int nf_avg_max) {
    const int TS = 16 * R;
    LdsOff L{};
    int off = 0;
    L.part = off; off = align16(off + WPB * TS * 16 * 4);
    L.cd = off;   off = align16(off + TS * NC * 8);
    L.xs = off;   off = align16(off + TS * 8 * 4);
    L.praw = off; off = align16(off + TS * 8 * 8);
    L.scal = off; off = align16(off + TS * 8 * 8);
    L.sysv = off; off = align16(off + TS * (kmax > 0 ? kmax : 1) * 8);
    // reconstruction buffer: as many samples as fit ~32 KiB, at most the tile
    int SB = TS;
    while (SB > 1 && SB * NT * 8 > 32 * 1024) SB >>= 1;
    L.SB = SB;
    L.mag = off;  off = align16(off + SB * NT * 8);
    L.nf_max = nf_avg_max;
    L.est = off;  off = align16(off + TS * nf_avg_max * 8);
    L.total = off;
    return L;
}

enum ScalIdx { S_ZP1 = 0, S_TS = 1, S_DMOD = 2, S_RC = 3, S_EBV = 4, S_BAD = 5 };

template <int R, int WPB, int KP>
__global__ __launch_bounds__(64 * WPB, 2) void em_fused(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const int mode,
    const LdsOff L, double* __restrict__ chi_out, double* __restrict__ gp_out,
    float* __restrict__ coeff_out, double* __restrict__ tobs_out, double* __restrict__ mag_out) {
    constexpr int TS = 16 * R;
    constexpr int NTHR = 64 * WPB;
    constexpr int RECF = rec_floats(KP);

    const EmDev& P = *Pp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* part = reinterpret_cast<float*>(smem + L.part);
    double* cd = reinterpret_cast<double*>(smem + L.cd);
    float* xs = reinterpret_cast<float*>(smem + L.xs);
    double* praw = reinterpret_cast<double*>(smem + L.praw);
    double* scal = reinterpret_cast<double*>(smem + L.scal);
    double* sysv = reinterpret_cast<double*>(smem + L.sysv);
    double* magb = reinterpret_cast<double*>(smem + L.mag);
    double* estb = reinterpret_cast<double*>(smem + L.est);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long tile0 = (long)blockIdx.x * TS;
    const int o = blockIdx.y;  // observed filter (MODE_LOGL) / model filter (other modes)
    const int NP = P.NP, NC = P.NC, NT = P.NT, NS = P.NS;
    const int kmax = P.kmax > 0 ? P.kmax : 1;

    // ------------------------------------------------------------------ prologue
    // per-sample scalars: em_parameter_setup (model.py:288-303) + conversions
    if (tid < TS) {
        long b = tile0 + tid;
        if (b >= B) b = B - 1;
        const double* row = theta + b * ld;
        for (int p = 0; p < NMMA_MAX_PARAMS; ++p)
            praw[tid * 8 + p] = (p < NP) ? apply_slot(P.model_param[p], row) : 0.0;
        const double d_l = apply_slot(P.lumdist, row);
        double z = 0.0;
        if (P.redshift_mode == NMMA_Z_SLOT) {
            z = apply_slot(P.redshift, row);
        } else if (P.redshift_mode == NMMA_Z_GRID) {
            z = interp_np(d_l, P.dist_grid, P.z_grid, P.n_cosmo, P.z_grid[0], P.z_grid[P.n_cosmo - 1]);
        }
        scal[tid * 8 + S_ZP1] = 1 + z;
        scal[tid * 8 + S_TS] = apply_slot(P.timeshift, row);
        scal[tid * 8 + S_DMOD] = distance_modulus(d_l);
        scal[tid * 8 + S_RC] = redshift_correction(z);
        scal[tid * 8 + S_EBV] = P.has_ebv ? apply_slot(P.ebv, row) : 0.0;
        // a non-finite input makes the reference return the floor (NaN propagates through
        // relu/np.dot to every magnitude); fmaxf would swallow the NaN, so flag it here.
        double chk = d_l + z + scal[tid * 8 + S_TS] + scal[tid * 8 + S_EBV];
        for (int p = 0; p < NP; ++p) chk += praw[tid * 8 + p];
        if (mode == MODE_LOGL) {
            const int nn = P.sys_nn[o];
            const int so = P.sys_off[o];
            for (int k = 0; k < nn; ++k) {
                const double v = apply_slot(P.sys_slots[so + k], row);
                sysv[tid * kmax + k] = v;
                chk += v;
            }
        }
        scal[tid * 8 + S_BAD] = (chk - chk == 0.0) ? 0.0 : 1.0;
    }

    const int nsrc = (mode == MODE_LOGL) ? P.nsrc[o] : 1;

    for (int ks = 0; ks < nsrc; ++ks) {
        const int m = (mode == MODE_LOGL) ? P.src[o * NMMA_MAX_SOURCES + ks] : o;
        __syncthreads();  // praw ready (ks = 0) / previous source fully consumed

        // normalised surrogate inputs, cast to fp32 as Keras does
        for (int idx = tid; idx < TS * 8; idx += NTHR) {
            const int p = idx & 7;
            float v = 0.f;
            if (p < NP) v = (float)((praw[idx] - P.pmin[m * NP + p]) / P.pspan[m * NP + p]);
            xs[idx] = v;
        }
        __syncthreads();

        // -------------------------------------------------------------- phase A: MLP on MFMA
        {
            float xB[R][KP];
#pragma unroll
            for (int rb = 0; rb < R; ++rb)
#pragma unroll
                for (int kp = 0; kp < KP; ++kp)
                    xB[rb][kp] = xs[(rb * 16 + (lane & 15)) * 8 + 4 * kp + (lane >> 4)];

            const int HBW = P.HB / WPB;
            gcf32p rec = as_global(P.wrec) + ((size_t)m * (P.HB + 2) + (size_t)wave * HBW) * RECF;
            const int boff = 256 + 64 * KP + (lane >> 4) * 4;

            f32x4 acc[R][2];
#pragma unroll
            for (int rb = 0; rb < R; ++rb) { acc[rb][0] = f32x4{0, 0, 0, 0}; acc[rb][1] = f32x4{0, 0, 0, 0}; }

            // record 0: layer-1 pre-activations; record 1 in flight
            f32x4 a2_cur = *reinterpret_cast<gcf32x4p>(rec + lane * 4);
            f32x4 d[R];
            {
                float a1[KP];
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) a1[kp] = rec[256 + kp * 64 + lane];
                const f32x4 bias = *reinterpret_cast<gcf32x4p>(rec + boff);
#pragma unroll
                for (int rb = 0; rb < R; ++rb) {
                    d[rb] = bias;
#pragma unroll
                    for (int kp = 0; kp < KP; ++kp)
                        d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kp], xB[rb][kp], d[rb], 0, 0, 0);
                }
            }
            f32x4 a2_nxt = *reinterpret_cast<gcf32x4p>(rec + RECF + lane * 4);
            float a1_nxt[KP];
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) a1_nxt[kp] = rec[RECF + 256 + kp * 64 + lane];
            f32x4 b_nxt = *reinterpret_cast<gcf32x4p>(rec + RECF + boff);
            opaque(a2_cur); opaque(a2_nxt); opaque(b_nxt);
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) opaque(a1_nxt[kp]);

            for (int i = 0; i < HBW; ++i) {
                // relu of this record's hidden units: the B operands of layer 2
                f32x4 h[R];
#pragma unroll
                for (int rb = 0; rb < R; ++rb) {
                    h[rb][0] = relu1(d[rb][0]); h[rb][1] = relu1(d[rb][1]);
                    h[rb][2] = relu1(d[rb][2]); h[rb][3] = relu1(d[rb][3]);
                }
                // layer 1 of the NEXT record (independent of the layer-2 chain below)
#pragma unroll
                for (int rb = 0; rb < R; ++rb) {
                    d[rb] = b_nxt;
#pragma unroll
                    for (int kp = 0; kp < KP; ++kp)
                        d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1_nxt[kp], xB[rb][kp], d[rb], 0, 0, 0);
                }
                const f32x4 a2 = a2_cur;
                a2_cur = a2_nxt;
                // prefetch record i+2 (two zero records pad the end of every filter)
                gcf32p rn = rec + (size_t)(i + 2) * RECF;
                a2_nxt = *reinterpret_cast<gcf32x4p>(rn + lane * 4);
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) a1_nxt[kp] = rn[256 + kp * 64 + lane];
                b_nxt = *reinterpret_cast<gcf32x4p>(rn + boff);
                // layer 2: C^T[coef][sample] += W2^T[coef][4 hidden] * H^T[4 hidden][sample]
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int rb = 0; rb < R; ++rb)
                        acc[rb][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[r], h[rb][r], acc[rb][r & 1], 0, 0, 0);
            }
            // partial C^T of this wave's hidden slice -> LDS
#pragma unroll
            for (int rb = 0; rb < R; ++rb) {
                const f32x4 s = acc[rb][0] + acc[rb][1];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    part[((wave * R + rb) * 16 + (lane >> 4) * 4 + r) * 16 + (lane & 15)] = s[r];
            }
        }
        __syncthreads();

        // cross-wave reduction (fixed order) + bias of the second Dense -> fp64 coefficients
        for (int idx = tid; idx < TS * 16; idx += NTHR) {
            const int rb = idx >> 8, rem = idx & 255, coef = rem >> 4, sidx = rem & 15;
            float c = 0.f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) c += part[((w * R + rb) * 16 + coef) * 16 + sidx];
            c += P.b2[m * 16 + coef];
            if (coef < NC) {
                const int s = rb * 16 + sidx;
                cd[s * NC + coef] = (double)c;
                if (mode == MODE_COEFF && tile0 + s < B)
                    coeff_out[((tile0 + s) * P.M + m) * NC + coef] = c;
            }
        }
        if (mode == MODE_COEFF) return;
        __syncthreads();

        // -------------------------------------------------------------- phase B
        const int SB = L.SB;
        const int jlo = P.s1_range[m * 4 + 0], jhi = P.s1_range[m * 4 + 1];
        const bool identity = P.s1_range[m * 4 + 2] != 0;
        const int* s1_idx = P.s1_idx + (size_t)m * NS;
        const double* s1_dx = P.s1_dx + (size_t)m * NS;
        const double* s1_off = P.s1_off + (size_t)m * NS;
        const double* st = P.st;
        const double ebvc = P.has_ebv ? P.ebv_coeff[m] : 0.0;

        for (int sb0 = 0; sb0 < TS; sb0 += SB) {
            // ---- B2: mag_abs[s][t] = (VA[t,:] . c[s]) * span[t] + mins[t]
            {
                const double* VAt = P.VAt + (size_t)m * NC * NT;
                for (int t = tid; t < NT; t += NTHR) {
                    const double sp = P.span[m * NT + t], mn = P.mins[m * NT + t];
                    if (NC == 10) {
                        double va[10];
#pragma unroll
                        for (int j = 0; j < 10; ++j) va[j] = VAt[j * NT + t];
                        for (int s = 0; s < SB; ++s) {
                            const double* c = cd + (sb0 + s) * 10;
                            double a = va[0] * c[0];
#pragma unroll
                            for (int j = 1; j < 10; ++j) a = fma(va[j], c[j], a);
                            magb[s * NT + t] = a * sp + mn;
                        }
                    } else {
                        for (int s = 0; s < SB; ++s) {
                            const double* c = cd + (sb0 + s) * NC;
                            double a = VAt[t] * c[0];
                            for (int j = 1; j < NC; ++j) a = fma(VAt[j * NT + t], c[j], a);
                            magb[s * NT + t] = a * sp + mn;
                        }
                    }
                }
            }
            __syncthreads();

            // apparent magnitude at sample node j of sub-batch sample sl (model.py:374-404)
            auto app_mag = [&](const double* magrow, int j, double ext, double dmod, double rc) -> double {
                const int i1 = s1_idx[j];
                double v;
                if (identity) {
                    v = magrow[i1];
                } else {
                    const double y0 = magrow[i1];
                    const double y1 = magrow[i1 + 1 < NT ? i1 + 1 : NT - 1];
                    const double slope = (y1 - y0) / s1_dx[j];
                    v = slope * s1_off[j] + y0;
                }
                if (ext != 0.0) v = v + ext;
                return (v + dmod) + rc;
            };

            if (mode == MODE_LC) {
                for (int idx = tid; idx < SB * NS; idx += NTHR) {
                    const int sl = idx / NS, j = idx - sl * NS;
                    const int s = sb0 + sl;
                    const long b = tile0 + s;
                    if (b >= B) continue;
                    const double zp1 = scal[s * 8 + S_ZP1], tsh = scal[s * 8 + S_TS];
                    const double ebv = scal[s * 8 + S_EBV];
                    const double ext = (ebv != 0.0) ? ebvc * ebv : 0.0;
                    double v = dinf();
                    if (j >= jlo && j <= jhi && jhi > jlo)
                        v = app_mag(magb + sl * NT, j, ext, scal[s * 8 + S_DMOD], scal[s * 8 + S_RC]);
                    mag_out[(b * P.M + m) * NS + j] = v;
                    if (m == 0) tobs_out[b * NS + j] = st[j] * zp1 + tsh;
                }
            } else {
                // ---- B3: lane groups walk the ragged data of observed filter o
                const int G = P.group[o];
                const int gpb = NTHR / G;           // groups per block
                const int g = tid / G, gi = tid - g * G;
                const int d0 = P.doff[o], d1 = P.doff[o + 1];
                const int nf = d1 - d0;
                const int kind = P.sys_kind[o];
                const double lim = P.lim[o];
                const int npass = (SB + gpb - 1) / gpb;
                for (int pass = 0; pass < npass; ++pass) {
                    const int sl = pass * gpb + g;
                    const bool active = sl < SB;
                    const int s = sb0 + (active ? sl : 0);
                    const double zp1 = scal[s * 8 + S_ZP1], tsh = scal[s * 8 + S_TS];
                    const double dmod = scal[s * 8 + S_DMOD], rc = scal[s * 8 + S_RC];
                    const double ebv = scal[s * 8 + S_EBV];
                    const double ext = (ebv != 0.0) ? ebvc * ebv : 0.0;
                    const double* magrow = magb + (active ? sl : 0) * NT;
                    const double t_lo = st[jlo] * zp1 + tsh, t_hi = st[jhi] * zp1 + tsh;
                    double chi = 0.0, gp = 0.0;
                    if (active) {
                        for (int dd = gi; dd < nf; dd += G) {
                            const int di = d0 + dd;
                            const double t = P.dt[di];
                            // stage-2: np.interp(t, t_obs[jlo..jhi], app, left=right=+inf)
                            double est;
                            if (!(jhi > jlo) || t < t_lo || t > t_hi || t != t) {
                                est = (t != t) ? t : dinf();
                            } else if (t == t_hi) {
                                est = app_mag(magrow, jhi, ext, dmod, rc);
                            } else {
                                int lo = jlo, hi = jhi;
                                while (hi - lo > 1) {
                                    const int mid = (lo + hi) >> 1;
                                    if (st[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                                }
                                const double x0 = st[lo] * zp1 + tsh;
                                const double y0 = app_mag(magrow, lo, ext, dmod, rc);
                                if (x0 == t) {
                                    est = y0;
                                } else {
                                    const double x1 = st[lo + 1] * zp1 + tsh;
                                    const double y1 = app_mag(magrow, lo + 1, ext, dmod, rc);
                                    est = lerp_np(t, x0, x1, y0, y1);
                                }
                            }
                            if (nsrc > 1) {  // averaged band: (a + b [+ c]) / n  (utils.py:566-584)
                                double acc_e = est;
                                if (ks > 0) acc_e = estb[s * L.nf_max + dd] + est;
                                if (ks < nsrc - 1) { estb[s * L.nf_max + dd] = acc_e; continue; }
                                est = acc_e / (double)nsrc;
                            }
                            // systematics (systematics.py:279-296) and combined sigma (em_likelihood.py:341)
                            const double sd = P.dsig[di];
                            double e, sig, lsig;
                            if (kind == NMMA_SYS_CONST) {
                                e = P.sys_const[o]; sig = P.dsigtot[di]; lsig = P.dlogsig[di];
                            } else {
                                const double* v = sysv + s * kmax;
                                if (kind == NMMA_SYS_PARAM) {
                                    e = v[0];
                                } else {
                                    const int K = P.sys_nn[o];
                                    const int ni = P.sys_nidx[di];
                                    if (ni < 0) e = v[0];
                                    else if (ni >= K - 1) e = v[K - 1];
                                    else { const double sl2 = (v[ni + 1] - v[ni]) / P.sys_ndx[di]; e = sl2 * P.sys_noff[di] + v[ni]; }
                                }
                                sig = sqrt(sd * sd + e * e);
                                lsig = log(sig);
                            }
                            const double mobs = P.dm[di];
                            if (sig - sig == 0.0) {   // np.isfinite(data_sigma): detection
                                chi += detection_term(mobs, est, sig, lsig, lim);
                            } else {                  // infinite error: upper limit
                                gp += upper_limit_term(mobs, est, e);
                            }
                        }
                    }
                    if (nsrc > 1 && ks < nsrc - 1) continue;  // uniform per block
                    // group reduction (G lanes, same wave); inactive groups carry zeros
                    for (int off = G >> 1; off > 0; off >>= 1) {
                        chi += __shfl_xor(chi, off);
                        gp += __shfl_xor(gp, off);
                    }
                    if (active && gi == 0 && tile0 + s < B) {
                        if (scal[s * 8 + S_BAD] != 0.0) chi = dnan();
                        chi_out[(long)o * B + tile0 + s] = chi;
                        gp_out[(long)o * B + tile0 + s] = gp;
                    }
                }
            }
            __syncthreads();  // magb reused by the next sub-batch / source
        }
    }
}

// Sum over observed filters in the reference's order and floor non-finite results
// (em_likelihood.py:337-352; core/base.py:178-182).
__global__ void em_combine(const double* __restrict__ chi, const double* __restrict__ gp, long B, int O,
                           int always_floor, double* __restrict__ out) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double c = 0.0, g = 0.0;
    bool bad = always_floor != 0;
    for (int o = 0; o < O; ++o) {
        const double x = chi[(long)o * B + b];
        if (x != x) bad = true;
        c += x;
        g += gp[(long)o * B + b];
    }
    double tot = c + g;
    if (bad || !(tot - tot == 0.0)) tot = NMMA_LOGL_FLOOR;
    out[b] = tot;
}

}  // namespace nmma

#include "em_api.inc"
