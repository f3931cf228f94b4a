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
// Each wave owns a contiguous run of hidden units and streams its pre-swizzled weight
// records straight from L2 into VGPRs (no LDS: nothing is shared between waves).
// Phase B (fp64 VALU): lane groups walk the ragged data of the filter; every datum
// brackets its epoch on the redshifted grid and reconstructs ONLY the light-curve nodes
// it interpolates between (2, or 4 when sample_times differ from the SVD grid) from the
// LDS-resident basis rows -- same arithmetic per node as the dense reconstruction.
// MODE_LC (gen_detector_lc for plots/tests) reconstructs the whole curve instead.
//
// em_combine adds the per-filter partial sums in the reference's order and applies the
// floor (core/base.py:82, :180-181).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "em_device.h"
#include "em_math.h"

namespace nmma {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Pointers read out of the EmDev record have no provable address space; tell the
// compiler they are global so it emits global_load (vmcnt only) instead of flat_load.
typedef const __attribute__((address_space(1))) float* gcf32p;
typedef const __attribute__((address_space(1))) f32x4* gcf32x4p;
typedef const __attribute__((address_space(1))) double* gcf64p;
typedef const __attribute__((address_space(1))) int* gci32p;
__device__ __forceinline__ gcf32p as_global(const float* p) { return (gcf32p)(uintptr_t)p; }
__device__ __forceinline__ gcf64p as_global(const double* p) { return (gcf64p)(uintptr_t)p; }
__device__ __forceinline__ gci32p as_global(const int* p) { return (gci32p)(uintptr_t)p; }

// relu on an MFMA result through a builtin the compiler can see: v_med3_f32(x, 0, +inf).
// (An inline-asm v_max is invisible to hipcc's hazard recogniser -- it left only 1 wait
// state between the asm's VGPR write and the MFMA reading it as SrcB, and the R=1/KP=2
// instantiation read stale operands.)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }

// Opaque identity: stops InstCombine from folding phi(load, load) into load(phi(addr)),
// which would move every prefetched weight load back to its use (no latency hiding).
__device__ __forceinline__ void opaque(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque(float& v) { asm volatile("" : "+v"(v)); }

// Hidden units are always split into NSLICE partial sums added in slice order, so the
// fp32 result does not depend on the launch geometry (R, WPB) chosen for a batch size.
constexpr int NSLICE = 8;

// Byte offsets of the LDS carve-up (computed by the host with lds_layout()).
struct LdsOff {
    int32_t part, cd, xs, praw, scal, sysv, stl, s1i, s1dx, s1of, va, span, mins, mag, est, total;
    int32_t SB;        // MODE_LC: samples per dense reconstruction sub-batch
    int32_t nf_max;    // widest averaged filter (est buffer row length), 0 if none
};

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }

__host__ inline LdsOff lds_layout(int mode, int R, int NC, int NT, int NS, int kmax, int nf_avg_max) {
    const int TS = 16 * R;
    LdsOff L{};
    int off = 0;
    L.part = off; off = align16(off + NSLICE * TS * 16 * 4);
    L.cd = off;   off = align16(off + TS * NC * 8);
    L.xs = off;   off = align16(off + TS * 8 * 4);
    L.praw = off; off = align16(off + TS * 8 * 8);
    L.scal = off; off = align16(off + TS * 8 * 8);
    L.sysv = off; off = align16(off + TS * (kmax > 0 ? kmax : 1) * 8);
    L.stl = off;  off = align16(off + NS * 8);      // sample times
    L.s1i = off;  off = align16(off + NS * 4);      // stage-1 tables of the current model filter
    L.s1dx = off; off = align16(off + NS * 8);
    L.s1of = off; off = align16(off + NS * 8);
    L.SB = 0;
    L.va = L.span = L.mins = L.mag = off;
    if (mode == MODE_LOGL) {
        L.va = off;   off = align16(off + NT * NC * 8);   // basis rows of the current model filter
        L.span = off; off = align16(off + NT * 8);
        L.mins = off; off = align16(off + NT * 8);
    } else if (mode == MODE_LC) {
        int SB = TS;   // dense buffer: as many samples as fit ~32 KiB
        while (SB > 1 && SB * NT * 8 > 32 * 1024) SB >>= 1;
        L.SB = SB;
        L.mag = off;  off = align16(off + SB * NT * 8);
    }
    L.nf_max = nf_avg_max;
    L.est = off;  off = align16(off + TS * nf_avg_max * 8);
    L.total = off;
    return L;
}

enum ScalIdx { S_ZP1 = 0, S_TS = 1, S_DMOD = 2, S_RC = 3, S_EBV = 4, S_BAD = 5, S_IZP1 = 6 };

template <int MODE, int R, int WPB, int KP>
__global__ __launch_bounds__(64 * WPB, 3) void em_fused(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld,
    const LdsOff L, double* __restrict__ chi_out, double* __restrict__ gp_out,
    float* __restrict__ coeff_out, double* __restrict__ tobs_out, double* __restrict__ mag_out) {
    constexpr int TS = 16 * R;
    constexpr int NTHR = 64 * WPB;
    constexpr int RECF = rec_floats(KP);

    constexpr int mode = MODE;
    const EmDev& P = *Pp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* part = reinterpret_cast<float*>(smem + L.part);
    double* cd = reinterpret_cast<double*>(smem + L.cd);
    float* xs = reinterpret_cast<float*>(smem + L.xs);
    double* praw = reinterpret_cast<double*>(smem + L.praw);
    double* scal = reinterpret_cast<double*>(smem + L.scal);
    double* sysv = reinterpret_cast<double*>(smem + L.sysv);
    double* stl = reinterpret_cast<double*>(smem + L.stl);
    int* s1i = reinterpret_cast<int*>(smem + L.s1i);
    double* s1dx = reinterpret_cast<double*>(smem + L.s1dx);
    double* s1of = reinterpret_cast<double*>(smem + L.s1of);
    double* val = reinterpret_cast<double*>(smem + L.va);
    double* spanl = reinterpret_cast<double*>(smem + L.span);
    double* minsl = reinterpret_cast<double*>(smem + L.mins);
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
        scal[tid * 8 + S_IZP1] = 1.0 / (1 + z);   // only seeds the bracket guess (exactly re-checked)
        scal[tid * 8 + S_TS] = apply_slot(P.timeshift, row);
        scal[tid * 8 + S_DMOD] = distance_modulus(d_l);
        scal[tid * 8 + S_RC] = redshift_correction(z);
        scal[tid * 8 + S_EBV] = P.has_ebv ? apply_slot(P.ebv, row) : 0.0;
        // a non-finite input makes the reference return the floor (NaN propagates through
        // relu/np.dot to every magnitude); v_med3 would swallow the NaN, so flag it here.
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
    for (int j = tid; j < NS; j += NTHR) stl[j] = P.st[j];

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
        // static tables of this model filter -> LDS (consumed after later barriers)
        const bool identity = P.s1_range[m * 4 + 2] != 0;
        {
            gci32p gi1 = as_global(P.s1_idx) + (size_t)m * NS;
            for (int j = tid; j < NS; j += NTHR) s1i[j] = gi1[j];
            if (!identity) {
                gcf64p gdx = as_global(P.s1_dx) + (size_t)m * NS, gof = as_global(P.s1_off) + (size_t)m * NS;
                for (int j = tid; j < NS; j += NTHR) { s1dx[j] = gdx[j]; s1of[j] = gof[j]; }
            }
            if (mode == MODE_LOGL) {
                gcf64p gva = as_global(P.VA) + (size_t)m * NT * NC;
                for (int j = tid; j < NT * NC; j += NTHR) val[j] = gva[j];
                gcf64p gsp = as_global(P.span) + (size_t)m * NT, gmn = as_global(P.mins) + (size_t)m * NT;
                for (int j = tid; j < NT; j += NTHR) { spanl[j] = gsp[j]; minsl[j] = gmn[j]; }
            }
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

            constexpr int SPW = NSLICE / WPB;          // slices handled by this wave
            const int HBS = P.HB / NSLICE;             // hidden blocks (records) per slice
            gcf32p rec = as_global(P.wrec) + ((size_t)m * (P.HB + 2) + (size_t)wave * SPW * HBS) * RECF;
            const int boff = 256 + 64 * KP + (lane >> 4) * 4;

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

            int g = 0;   // record index within this wave's contiguous run
            for (int sl = 0; sl < SPW; ++sl) {
                f32x4 acc[R][2];
#pragma unroll
                for (int rb = 0; rb < R; ++rb) { acc[rb][0] = f32x4{0, 0, 0, 0}; acc[rb][1] = f32x4{0, 0, 0, 0}; }
                for (int i = 0; i < HBS; ++i, ++g) {
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
                    // prefetch record g+2 (two zero records pad the end of every filter)
                    gcf32p rn = rec + (size_t)(g + 2) * RECF;
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
                // partial C^T of this hidden slice -> LDS
                const int slice = wave * SPW + sl;
#pragma unroll
                for (int rb = 0; rb < R; ++rb) {
                    const f32x4 s = acc[rb][0] + acc[rb][1];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        part[((slice * R + rb) * 16 + (lane >> 4) * 4 + r) * 16 + (lane & 15)] = s[r];
                }
            }
        }
        __syncthreads();

        // slice reduction (fixed order) + bias of the second Dense -> fp64 coefficients
        for (int idx = tid; idx < TS * 16; idx += NTHR) {
            const int rb = idx >> 8, rem = idx & 255, coef = rem >> 4, sidx = rem & 15;
            float c = 0.f;
#pragma unroll
            for (int w = 0; w < NSLICE; ++w) c += part[((w * R + rb) * 16 + coef) * 16 + sidx];
            c += P.b2[m * 16 + coef];
            if (coef < NC) {
                const int s = rb * 16 + sidx;
                cd[s * NC + coef] = (double)c;
                if (mode == MODE_COEFF && tile0 + s < B)
                    coeff_out[((tile0 + s) * P.M + m) * NC + coef] = c;
            }
        }
        if constexpr (MODE == MODE_COEFF) return;
        __syncthreads();

        // -------------------------------------------------------------- phase B
        const int jlo = P.s1_range[m * 4 + 0], jhi = P.s1_range[m * 4 + 1];
        const double ebvc = P.has_ebv ? P.ebv_coeff[m] : 0.0;

        if constexpr (MODE == MODE_LC) {
            // dense reconstruction in sub-batches of SB samples (gen_detector_lc, model.py:352-404)
            const int SB = L.SB;
            gcf64p VAt = as_global(P.VAt) + (size_t)m * NC * NT;
            for (int sb0 = 0; sb0 < TS; sb0 += SB) {
                for (int t = tid; t < NT; t += NTHR) {
                    const double sp = P.span[m * NT + t], mn = P.mins[m * NT + t];
                    for (int s = 0; s < SB; ++s) {
                        const double* c = cd + (sb0 + s) * NC;
                        double a = VAt[t] * c[0];
                        for (int j = 1; j < NC; ++j) a = fma(VAt[j * NT + t], c[j], a);
                        magb[s * NT + t] = a * sp + mn;
                    }
                }
                __syncthreads();
                for (int idx = tid; idx < SB * NS; idx += NTHR) {
                    const int sl = idx / NS, j = idx - sl * NS;
                    const int s = sb0 + sl;
                    const long b = tile0 + s;
                    if (b >= B) continue;
                    const double zp1 = scal[s * 8 + S_ZP1], tsh = scal[s * 8 + S_TS];
                    const double ebv = scal[s * 8 + S_EBV];
                    const double ext = (ebv != 0.0) ? ebvc * ebv : 0.0;
                    double v = dinf();
                    if (j >= jlo && j <= jhi && jhi > jlo) {
                        const double* magrow = magb + sl * NT;
                        const int i1 = s1i[j];
                        if (identity) {
                            v = magrow[i1];
                        } else {
                            const double y0 = magrow[i1], y1 = magrow[i1 + 1 < NT ? i1 + 1 : NT - 1];
                            const double slope = (y1 - y0) / s1dx[j];
                            v = slope * s1of[j] + y0;
                        }
                        if (ext != 0.0) v = v + ext;
                        v = (v + scal[s * 8 + S_DMOD]) + scal[s * 8 + S_RC];
                    }
                    mag_out[(b * P.M + m) * NS + j] = v;
                    if (m == 0) tobs_out[b * NS + j] = stl[j] * zp1 + tsh;
                }
                __syncthreads();
            }
        }

        // ---- MODE_LOGL: lane groups walk the ragged data of observed filter o
        // (instantiated for <= 10 coefficients, the reference default, and for up to 16)
        auto logl_phase = [&](auto nct_tag) {
            constexpr int NCT = decltype(nct_tag)::value;
            const int G = P.group[o];
            const int gpb = NTHR / G;           // groups per block
            const int g = tid / G, gi = tid - g * G;
            const int d0 = P.doff[o], d1 = P.doff[o + 1];
            const int nf = d1 - d0;
            const int kind = P.sys_kind[o];
            const double lim = P.lim[o];
            const double e_const = P.sys_const[o];
            const bool uniform = P.st_uniform != 0;
            const double st0 = P.st0, inv_dt = P.st_inv_dt;
            const int npass = (TS + gpb - 1) / gpb;
            gcf64p g_dt = as_global(P.dt), g_dm = as_global(P.dm), g_dsig = as_global(P.dsig);
            gcf64p g_sigtot = as_global(P.dsigtot), g_logsig = as_global(P.dlogsig);
            // the lane's first datum is the same for every sample: keep it in registers
            double c_t = 0, c_m = 0, c_sd = 0, c_sig = 0, c_lsig = 0;
            if (gi < nf) {
                const int di = d0 + gi;
                c_t = g_dt[di]; c_m = g_dm[di]; c_sd = g_dsig[di];
                if (kind == NMMA_SYS_CONST) { c_sig = g_sigtot[di]; c_lsig = g_logsig[di]; }
            }
            for (int pass = 0; pass < npass; ++pass) {
                const int sl = pass * gpb + g;
                const bool active = sl < TS;
                const int s = active ? sl : 0;
                const double zp1 = scal[s * 8 + S_ZP1], tsh = scal[s * 8 + S_TS];
                const double dmod = scal[s * 8 + S_DMOD], rc = scal[s * 8 + S_RC];
                const double ebv = scal[s * 8 + S_EBV], izp1 = scal[s * 8 + S_IZP1];
                const double ext = (ebv != 0.0) ? ebvc * ebv : 0.0;
                const double t_lo = stl[jlo] * zp1 + tsh, t_hi = stl[jhi] * zp1 + tsh;
                // this sample's SVD coefficients (shared by the whole lane group)
                // fast path (NC <= 10): coefficients in registers; generic path re-reads LDS
                constexpr int NREG = (NCT <= 10) ? NCT : 1;
                double cc[NREG];
                const double* cl = cd + s * NC;
                if constexpr (NCT <= 10) {
#pragma unroll
                    for (int j = 0; j < NCT; ++j) cc[j] = (j < NC) ? cl[j] : 0.0;
                }

                // absolute magnitude at SVD-grid node i: (VA[i,:] . c) * span[i] + mins[i]
                auto mag_abs = [&](int i) -> double {
                    const double* row = val + i * NC;
                    double a;
                    if constexpr (NCT <= 10) {
                        a = row[0] * cc[0];
#pragma unroll
                        for (int j = 1; j < NCT; ++j)
                            if (j < NC) a = fma(row[j], cc[j], a);
                    } else {
                        a = row[0] * cl[0];
                        for (int j = 1; j < NC; ++j) a = fma(row[j], cl[j], a);
                    }
                    return a * spanl[i] + minsl[i];
                };
                // apparent magnitude at sample node j (stage-1 lerp + model.py:374-404)
                auto app_mag = [&](int j) -> double {
                    const int i1 = s1i[j];
                    double v;
                    if (identity) {
                        v = mag_abs(i1);
                    } else {
                        const double y0 = mag_abs(i1);
                        const double y1 = mag_abs(i1 + 1 < NT ? i1 + 1 : NT - 1);
                        const double slope = (y1 - y0) / s1dx[j];
                        v = slope * s1of[j] + y0;
                    }
                    if (ext != 0.0) v = v + ext;
                    return (v + dmod) + rc;
                };

                double chi = 0.0, gp = 0.0;
                if (active) {
                    for (int dd = gi; dd < nf; dd += G) {
                        const int di = d0 + dd;
                        double t, mobs, sd, sig, lsig;
                        if (dd == gi) { t = c_t; mobs = c_m; sd = c_sd; sig = c_sig; lsig = c_lsig; }
                        else {
                            t = g_dt[di]; mobs = g_dm[di]; sd = g_dsig[di]; sig = 0; lsig = 0;
                            if (kind == NMMA_SYS_CONST) { sig = g_sigtot[di]; lsig = g_logsig[di]; }
                        }
                        // stage-2: np.interp(t, t_obs[jlo..jhi], app, left=right=+inf)
                        double est;
                        if (!(jhi > jlo) || t < t_lo || t > t_hi || t != t) {
                            est = (t != t) ? t : dinf();
                        } else if (t == t_hi) {
                            est = app_mag(jhi);
                        } else {
                            // bracket: t_obs[lo] <= t < t_obs[lo+1]
                            int lo;
                            if (uniform) {
                                lo = (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                                lo = lo < jlo ? jlo : (lo > jhi - 1 ? jhi - 1 : lo);
                                while (lo < jhi - 1 && stl[lo + 1] * zp1 + tsh <= t) ++lo;
                                while (lo > jlo && stl[lo] * zp1 + tsh > t) --lo;
                            } else {
                                lo = jlo;
                                int hi = jhi;
                                while (hi - lo > 1) {
                                    const int mid = (lo + hi) >> 1;
                                    if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                                }
                            }
                            const double x0 = stl[lo] * zp1 + tsh;
                            const double y0 = app_mag(lo);
                            if (x0 == t) {
                                est = y0;
                            } else {
                                const double x1 = stl[lo + 1] * zp1 + tsh;
                                const double y1 = app_mag(lo + 1);
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
                        double e = e_const;
                        if (kind != NMMA_SYS_CONST) {
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
        };
        if constexpr (MODE == MODE_LOGL) {
            if (NC <= 10) logl_phase(std::integral_constant<int, 10>{});
            else logl_phase(std::integral_constant<int, NMMA_MAX_COEFF>{});
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
