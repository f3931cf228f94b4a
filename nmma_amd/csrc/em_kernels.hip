// em_kernels.hip -- gfx950 kernels of the batched EM light-curve log-likelihood other than the hot path (em_logl: em_logl.h,
// instantiated by em_logl_*.hip) -- auxiliary outputs, likelihood from supplied curves, models beyond the surrogate -- and, through
// em_api.inc, the host side of the C ABI.  Kernel overview: em_common.h.
#include "em_host.h"

namespace nmma {

// =======================================================================================
// em_fused: auxiliary outputs (MODE_COEFF: surrogate coefficients; MODE_LC: full
// detector-frame light curves, gen_detector_lc model.py:352-404), one model filter per
// blockIdx.y, all WPB waves on the MLP first.
// =======================================================================================
struct LdsOff {
    int32_t part, cd, praw, scal, stl, s1, mag, total;     // s1: the filter's stage-1 lerp tables [dx NS f64 | off NS f64 | idx NS i32]
    int32_t SB;        // MODE_LC: samples per dense reconstruction sub-batch
};

__host__ inline LdsOff lds_layout(int mode, int R, int NC, int NT, int NS) {
    const int TS = 16 * R;
    LdsOff L{};
    int off = 0;
    // (the partial sums are dead once the coefficients are reduced: the dense magnitude buffer of the light-curve modes
    //  takes their place -- the kernel is latency-bound and every block more per CU counts: 3 -> 4 for BASELINE config 2)
    L.cd = off;   off = align16(off + TS * NC * 8);
    L.praw = off; off = align16(off + TS * 8 * 8);
    L.scal = off; off = align16(off + TS * 8 * 8);
    L.stl = off;  off = align16(off + NS * 8);
    L.s1 = off;   off = align16(off + ((mode == MODE_LC || mode == MODE_LC_ABS) ? NS * 20 : 0));
    L.SB = 0;
    L.part = off;
    L.mag = off;
    int un = NSLICE * TS * PSTR * 4;
    if (mode == MODE_LC || mode == MODE_LC_ABS) {
        int SB = TS;   // dense buffer: as many samples as fit ~32 KiB
        while (SB > 1 && SB * NT * 8 > 32 * 1024) SB >>= 1;
        L.SB = SB;
        un = un > SB * NT * 8 ? un : SB * NT * 8;
    }
    off = align16(off + un);
    L.total = off;
    return L;
}

// Which of a workgroup's units of work (tiles of em_fused, row blocks of em_lc_loglike: unit u = first + i * stride, i = 0 .. 63) hold a
// row flagged in only_rows: bit i of the result, the same in every wave of the workgroup (each wave does the loads itself: no LDS, no
// barrier).  A unit covers `rows` (4 .. 32, a power of two) consecutive rows; only_rows is padded with zeros to whole units.
__device__ __forceinline__ unsigned long long flagged_units(const unsigned char* __restrict__ only_rows, const long first, const long stride,
                                                            const long n_units, const int rows) {
    const long u = first + (long)(threadIdx.x & 63) * stride;
    bool any = false;
    if (u < n_units) {
        const unsigned char* p = only_rows + u * rows;
        if (rows >= 8) {
            unsigned long long acc = 0;
            for (int q = 0; q < rows / 8; ++q) acc |= reinterpret_cast<const unsigned long long*>(p)[q];
            any = acc != 0;
        } else {
            any = *reinterpret_cast<const unsigned*>(p) != 0;
        }
    }
    return __ballot(any);
}

// The kernel proper: tile `bidx` of 16 R parameter vectors, model filter m.
template <int MODE, int R, int WPB, int KP>
__device__ __forceinline__ void em_fused_body(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const LdsOff L,
    float* __restrict__ coeff_out, double* __restrict__ tobs_out, double* __restrict__ mag_out, const unsigned bidx, const int m) {
    constexpr int TS = 16 * R;
    constexpr int NTHR = 64 * WPB;
    constexpr int RECF = rec_floats(KP);
    constexpr int SPW = NSLICE / WPB;

    const EmDev& P = *Pp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* part = reinterpret_cast<float*>(smem + L.part);
    double* cd = reinterpret_cast<double*>(smem + L.cd);
    double* praw = reinterpret_cast<double*>(smem + L.praw);
    double* scal = reinterpret_cast<double*>(smem + L.scal);
    double* stl = reinterpret_cast<double*>(smem + L.stl);
    double* magb = reinterpret_cast<double*>(smem + L.mag);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long tile0 = (long)bidx * TS;
    const int NP = P.NP, NC = P.NC, NT = P.NT, NS = P.NS;

    if constexpr (MODE == MODE_LC) {
        if (tid < TS) {
            long b = tile0 + tid;
            if (b >= B) b = B - 1;
            double chk;
            sample_scalars(P, theta + b * ld, praw + tid * 8, scal + tid * 8, chk);
        }
    } else {
        // coefficients and source-frame curves need the model parameters only: one thread per (sample, parameter) instead of the
        // serial per-sample chain (cosmology-grid search, log10) that the detector-frame outputs go through
        for (int idx = tid; idx < TS * 8; idx += NTHR) {
            const int s = idx >> 3, p = idx & 7;
            long b = tile0 + s;
            if (b >= B) b = B - 1;
            praw[s * 8 + p] = (p < NP) ? apply_slot(P.model_param[p], theta + b * ld) : 0.0;
        }
    }
    for (int j = tid; j < NS; j += NTHR) stl[j] = P.st[j];
    // the filter's stage-1 tables next to the sample times: the write-out loop below then runs on LDS latency (it used to
    // take three L2 round trips per element with one element in flight per thread -- 2/3 of the kernel's time)
    double* s1dx_l = reinterpret_cast<double*>(smem + L.s1);
    double* s1of_l = s1dx_l + NS;
    int* s1i_l = reinterpret_cast<int*>(s1of_l + NS);
    if constexpr (MODE == MODE_LC || MODE == MODE_LC_ABS) {
        for (int j = tid; j < NS; j += NTHR) {
            s1dx_l[j] = P.s1_dx[(size_t)m * NS + j]; s1of_l[j] = P.s1_off[(size_t)m * NS + j]; s1i_l[j] = P.s1_idx[(size_t)m * NS + j];
        }
    }
    __syncthreads();

    // ---- MLP
    {
        float xB[R][KP];
#pragma unroll
        for (int rb = 0; rb < R; ++rb)
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                const int p = 4 * kp + (lane >> 4);
                const int s = rb * 16 + (lane & 15);
                xB[rb][kp] = (p < NP) ? (float)((praw[s * 8 + p] - P.pmin[m * NP + p]) / P.pspan[m * NP + p]) : 0.f;
            }
        const int HBS = P.HB / NSLICE;
        gcf32p rec = as_global(P.wrec) + ((size_t)m * (P.HB + NPAD_REC) + (size_t)wave * SPW * HBS) * RECF;
        mlp_slices<R, KP, 4, SPW>(rec, xB, HBS, lane, part, wave * SPW);
    }
    __syncthreads();

    // slice reduction (fixed order) + bias of the second Dense -> fp64 coefficients
    for (int idx = tid; idx < TS * 16; idx += NTHR) {
        const int rb = idx >> 8, rem = idx & 255, coef = rem >> 4, sidx = rem & 15;
        float c = 0.f;
#pragma unroll
        for (int w = 0; w < NSLICE; ++w) c += part[((w * R + rb) * 16 + sidx) * PSTR + coef];
        c += P.b2[m * 16 + coef];
        if (coef < NC) {
            const int s = rb * 16 + sidx;
            cd[s * NC + coef] = (double)c;
            if (MODE == MODE_COEFF && tile0 + s < B) coeff_out[((tile0 + s) * P.M + m) * NC + coef] = c;
        }
    }
    if constexpr (MODE == MODE_COEFF) return;
    __syncthreads();

    // ---- dense reconstruction in sub-batches of SB samples
    const int jlo = P.s1_range[m * 4 + 0], jhi = P.s1_range[m * 4 + 1];
    const bool identity = P.s1_range[m * 4 + 2] != 0;
    const double ebvc = P.has_ebv ? P.ebv_coeff[m] : 0.0;
    gcf64p VAt = as_global(P.VAt) + (size_t)m * NC * NT;
    const int SB = L.SB;
    // (fallback loop below, taken when fewer than 16 samples fit the dense buffer -- SVD grids of more than 256 nodes -- or with more
    //  than 16 coefficients: a per-thread FMA chain, the block's threads shared by G groups of samples when the grid has fewer nodes than the block has threads)
    const int G = NTHR / NT > 0 ? NTHR / NT : 1;
    for (int sb0 = 0; sb0 < TS; sb0 += SB) {
#ifndef NMMA_FUSED_VALU_RECON
        if (SB >= 16 && NC <= 16) {
            // mag[t][s] = (VA[t, :] . c[s, :]) span[t] + mins[t] for all NT nodes: the one dense product of the path, on the fp64
            // matrix cores -- 16 nodes x 16 samples per v_mfma_f64_16x16x4, K = NC in steps of 4 (zero-padded).  A lane holds
            // A[node lane % 16][k = lane / 16], B[k = lane / 16][sample lane % 16] and D[node 4 r + lane / 16][sample lane % 16]
            // (the fp64 result rows are interleaved, unlike the fp32 16x16x4 variant's 4 (lane / 16) + r: tools/ubench/mfma_f64_layout.hip).
            // (as a per-thread FMA loop this phase took 35 us of the kernel's 80 at 4096 rows: 2/5 of its time)
            typedef double f64x4_t __attribute__((ext_vector_type(4)));
            gcf64p VA = as_global(P.VA) + (size_t)m * NT * NC;
            const int n_tt = (NT + 15) / 16;
            const int n_st = SB / 16;
            // (the basis rows of up to four node tiles are requested before the first product -- one L2 round trip per group of
            //  tiles instead of one per MFMA -- and serve every 16-sample tile of the sub-batch)
            for (int tt0 = wave; tt0 < n_tt; tt0 += 4 * WPB) {
                double av[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t_a = (tt0 + q * WPB) * 16 + (lane & 15);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = kk * 4 + (lane >> 4);
                        av[q][kk] = (tt0 + q * WPB < n_tt && t_a < NT && k < NC) ? VA[(size_t)t_a * NC + k] : 0.0;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tt = tt0 + q * WPB;
                    if (tt >= n_tt) break;
                    for (int st = 0; st < n_st; ++st) {
                        const int s_b = sb0 + st * 16 + (lane & 15);
                        f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            if (kk * 4 >= NC) break;
                            const int k = kk * 4 + (lane >> 4);
                            const double bq = (k < NC) ? cd[s_b * NC + k] : 0.0;
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bq, acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int t = tt * 16 + 4 * r + (lane >> 4);
                            if (t < NT) magb[(st * 16 + (lane & 15)) * NT + t] = acc[r] * P.span[m * NT + t] + P.mins[m * NT + t];
                        }
                    }
                }
            }
        } else
#endif
        for (int idx = tid; idx < NT * G; idx += NTHR) {
            const int g = idx / NT, t = idx - g * NT;
            const double sp = P.span[m * NT + t], mn = P.mins[m * NT + t];
            for (int s = g; s < SB; s += G) {
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
            double zp1 = 1.0, tsh = 0.0, ext = 0.0;
            if constexpr (MODE == MODE_LC) {
                zp1 = scal[s * 8 + S_ZP1]; tsh = scal[s * 8 + S_TS];
                ext = extinction_mag(P.ext_law, ebvc, zp1, scal[s * 8 + S_EBV]);
            }
            double v = dinf();
            if (P.union_grid) {
                // Combined model on a union grid (nmma_em_config::base_times; the rare paths: curve outputs and the re-evaluation launch):
                // the surrogate on its OWN sample node i (stage 1, as below), then np.interp between two of those onto grid node j
                // (autocomplete_data(..., extrapolate=inf), model.py:1440-1448 -- nmma_lc_regrid's arithmetic); tables in global memory
                if (j >= jlo && j <= jhi && jhi > jlo) {
                    const double* magrow = magb + sl * NT;
                    const int NB = P.NB;
                    auto own_node = [&](const int i) {
                        const int a = P.b_idx[(size_t)m * NB + i];
                        const double of = P.b_off[(size_t)m * NB + i];
                        if (of == 0.0 || a + 1 >= NT) return magrow[a];
                        const double y0 = magrow[a], y1 = magrow[a + 1];
                        return ((y1 - y0) / P.b_dx[(size_t)m * NB + i]) * of + y0;
                    };
                    const int i = P.u_idx[(size_t)m * NS + j];
                    const double ya = own_node(i);
                    v = (P.u_off[(size_t)m * NS + j] == 0.0) ? ya : lerp_np(stl[j], P.bt[i], P.bt[i + 1], ya, own_node(i + 1));
                }
            } else if (j >= jlo && j <= jhi && jhi > jlo) {
                const double* magrow = magb + sl * NT;
                const int i1 = s1i_l[j];
                if (identity) {
                    v = magrow[i1];
                } else {
                    const double y0 = magrow[i1], y1 = magrow[i1 + 1 < NT ? i1 + 1 : NT - 1];
                    const double slope = (y1 - y0) / s1dx_l[j];
                    v = slope * s1of_l[j] + y0;
                }
                if constexpr (MODE == MODE_LC) {
                    if (ext != 0.0) v = v + ext;
                    v = (v + scal[s * 8 + S_DMOD]) + scal[s * 8 + S_RC];
                }
            }
            mag_out[(b * P.M + m) * NS + j] = v;
            if (MODE == MODE_LC && m == 0) tobs_out[b * NS + j] = stl[j] * zp1 + tsh;
        }
        __syncthreads();
    }
}

template <int MODE, int R, int WPB, int KP>
__global__ __launch_bounds__(64 * WPB, 2) void em_fused(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const LdsOff L,
    float* __restrict__ coeff_out, double* __restrict__ tobs_out, double* __restrict__ mag_out) {
    em_fused_body<MODE, R, WPB, KP>(Pp, theta, B, ld, L, coeff_out, tobs_out, mag_out, blockIdx.x, (int)blockIdx.y);
}

}  // namespace nmma

#include "em_lc.h"

namespace nmma {

// (the work on one flagged tile, out of line: inlined, its register spills were hoisted to the kernel's entry -- every workgroup of the
//  "nothing flagged" launch wrote its registers to scratch, 14.8 MB per call at config 3's shape, before looking at a single flag)
template <int KP, int G, bool SD>
__device__ __noinline__ void stack2_redo_tile(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const LdsOff Lf, double* __restrict__ kn_ws,
    const LcSets sets, const unsigned char* __restrict__ bad_rows, const int lds_per_sample, const int always_floor,
    double* __restrict__ out, const unsigned char* __restrict__ only_rows, const long tile) {
    constexpr int TS = 32, SPB = 4 * (64 / G);
    const int M = Pp->M;
    // (the surrogate's curves of this tile go to THIS WORKGROUP's 32 rows of the workspace, whatever the tile's position in the batch:
    //  both bodies index by batch row, so the base is moved back by the tile's offset -- the workspace then holds gridDim.x x 32
    //  rows instead of B)
    const long shift = ((long)blockIdx.x - tile) * TS * M * Pp->NS;
    double* const ws = kn_ws + shift;
    LcSets own = sets;
    own.p[0] = ws;
    for (int m = 0; m < M; ++m) {
        em_fused_body<MODE_LC_ABS, 2, 4, KP>(Pp, theta, B, ld, Lf, nullptr, nullptr, ws, (unsigned)tile, m);
        __syncthreads();
    }
    __threadfence();          // the curves this workgroup just wrote are what it reads next
    __syncthreads();
    for (int q = 0; q < TS / SPB; ++q) {
        const long blk = tile * (TS / SPB) + q;
        if (blk * SPB < B)
            em_lc_loglike_body<G, 2, SD, true, true>(Pp, theta, B, ld, own, 2, bad_rows, lds_per_sample, always_floor, out, nullptr, nullptr,
                                                     only_rows, (unsigned)blk);
        __syncthreads();
    }
}

template <int KP, int G, bool SD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void stack2_redo(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const LdsOff Lf, double* __restrict__ kn_ws,
    const LcSets sets, const unsigned char* __restrict__ bad_rows, const int lds_per_sample, const int always_floor,
    double* __restrict__ out, const unsigned char* __restrict__ only_rows) {
    constexpr int TS = 32;
    const long n_tiles = (B + TS - 1) / TS;
    for (long first = blockIdx.x; first < n_tiles; first += 64L * gridDim.x) {
        unsigned long long mask = flagged_units(only_rows, first, (long)gridDim.x, n_tiles, TS);
        while (mask != 0) {
            const int i = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            stack2_redo_tile<KP, G, SD>(Pp, theta, B, ld, Lf, kn_ws, sets, bad_rows, lds_per_sample, always_floor, out, only_rows,
                                        first + (long)i * gridDim.x);
        }
    }
}

// =======================================================================================
// me2017_lc: the Me2017 analytic kilonova (eff_metzger_lc, lightcurve_generation.py:566-652;
// blackbody magnitudes :43-58; flux_to_ABmag utils.py:793-811) -- BASELINE config 1.
// One wave per parameter vector: the 299 mass layers are spread over the lanes, the
// explicit-Euler time loop is sequential, the per-step layer sum and the photosphere
// argmin are wave reductions.  Output: source-frame absolute magnitudes lc[B][M][NS].
// =======================================================================================
namespace me17 {
constexpr double msun = 1.988409870698051e33, c_cgs = 2.99792458e10, h_cgs = 6.62607015e-27, kb = 1.380649e-16;
constexpr double sigSB = 5.6703744191844314e-05, D10pc = 10 * 3.085677581491367e18, day = 86400.0;
constexpr int MPREC = 300, NL = MPREC - 1, LPL = 5;   // layers per lane (5 * 64 >= 299)
}  // namespace me17

// (value, index) of the wave's smallest value, the lowest index among equals (np.argmin), in every lane.  The same DPP steps as
// group_sum -- lexicographic min is idempotent, so the overlapping mirror steps are as good as a butterfly -- instead of 18 dependent
// ds_bpermute round trips per call: the explicit-Euler loop of me2017_lc calls this once per time step with nothing to hide behind.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmin_step(double& v, int& idx) {
    const double ov = dpp_mov_f64<CTRL, ROW_MASK>(v);
    const int oi = __builtin_amdgcn_mov_dpp(idx, CTRL, ROW_MASK, 0xf, false);
    // (rows a row_mask leaves out read undefined values: they never hold the wave's result, which is taken from lane 63)
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
}

__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
    argmin_step<0xB1, 0xf>(v, idx);     // quad_perm [1,0,3,2]
    argmin_step<0x4E, 0xf>(v, idx);     // quad_perm [2,3,0,1]
    argmin_step<0x141, 0xf>(v, idx);    // row_half_mirror
    argmin_step<0x140, 0xf>(v, idx);    // row_mirror: every lane holds its row's result
    argmin_step<0x142, 0xA>(v, idx);    // row_bcast15 into rows 1 and 3
    argmin_step<0x143, 0xC>(v, idx);    // row_bcast31 into rows 2 and 3: lane 63 holds the wave's
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    v = __hiloint2double(hi, lo);
    idx = __builtin_amdgcn_readlane(idx, 63);
}

__global__ __launch_bounds__(256) void me2017_lc(const EmDev* __restrict__ Pp, const double* __restrict__ theta,
                                                 const long B, const long ld, const int lds_per_wave,
                                                 double* __restrict__ lc) {
    using namespace me17;
    const EmDev& P = *Pp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long b = (long)blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int NS = P.NS, M = P.M;
    double* eth = reinterpret_cast<double*>(smem + (size_t)wave * lds_per_wave);
    double* tpw = eth + NS;      // (t/day)^-1.3
    double* lsum = tpw + NS;     // sum over layers of lum[:, j]
    double* rph = lsum + NS;     // photosphere radius
    double* tobs = rph + NS;     // effective temperature
    double* enl = tobs + NS;     // exp(-t / 900 s) per node
    double* tsl = enl + NS;      // node time in seconds
    double* vml = tsl + NS;      // vm per layer [MPREC]
    const double* row = theta + b * ld;

    const double M0 = pow(10.0, apply_slot(P.model_param[0], row)) * msun;
    const double v0 = pow(10.0, apply_slot(P.model_param[1], row)) * c_cgs;
    const double beta = apply_slot(P.model_param[2], row);
    const double kappa_r = pow(10.0, apply_slot(P.model_param[3], row));
    double z = 0.0;
    if (P.redshift_mode == NMMA_Z_SLOT) z = apply_slot(P.redshift, row);
    else if (P.redshift_mode == NMMA_Z_GRID) {
        const double d_eff = apply_slot(P.lumdist, row) * (P.has_h0 ? apply_slot(P.hubble, row) * P.inv_h0_ref : 1.0);
        z = interp_np(d_eff, P.dist_grid, P.z_grid, P.n_cosmo, P.z_grid[0], P.z_grid[P.n_cosmo - 1]);
        if (P.has_h0) z *= d_eff;
    }

    // per-node time factors (thermalisation efficiency, Barnes+16 eq. 34)
    for (int j = lane; j < NS; j += 64) {
        const double td = P.st[j];
        const double f = 2 * 0.17 * pow(td, 0.74);
        eth[j] = 0.36 * (exp(-0.56 * td) + log(1.0 + f) / f);
        tpw[j] = pow((td * day) / day, -1.3);
        lsum[j] = 0.0;
        rph[j] = 0.0;
        tsl[j] = td * day;
        enl[j] = exp(-(td * day) / 900.0);
    }
    // mass layers: m = geomspace(1e-8, M0/msun, 300)
    const double ls = log10(1e-8), le = log10(M0 / msun);
    const double step = (le - ls) / (MPREC - 1);
    auto mlayer = [&](int i) -> double {
        if (i == 0) return 1e-8;
        if (i == MPREC - 1) return M0 / msun;
        return pow(10.0, i * step + ls);
    };
    double mms[LPL], vm[LPL], xn0[LPL], xr[LPL], dmm[LPL], ene[LPL];
    // per-layer factors of the time loop that do not depend on the time step (the loop then divides once per layer and
    // step instead of five times: DESIGN.md section 8):  tdiff = kappa A / t,  tau = kappa Bt / t^2,  t vm / c = t Cv
    double fa[LPL], fb[LPL], fc[LPL], krxr[LPL], omxr[LPL];
#pragma unroll
    for (int q = 0; q < LPL; ++q) {
        const int i = lane + 64 * q;
        const double mi = mlayer(i < MPREC ? i : MPREC - 1);
        const double mn = mlayer(i + 1 < MPREC ? i + 1 : MPREC - 1);
        mms[q] = mi * msun;
        double v = v0 * pow(mi * msun / M0, -1.0 / beta);
        if (v > c_cgs) v = c_cgs;
        vm[q] = v;
        xn0[q] = (1 - 2 * 0.1) * 2 * atan(1e-8 / mi) / kPi;
        xr[q] = 1.0 - xn0[q];
        dmm[q] = (mn - mi) * msun;
        ene[q] = 0.0;
        if (i < MPREC) vml[i] = v;
        fa[q] = 0.08 * mms[q] * 3 / (v * c_cgs * beta);
        fb[q] = mms[q] / (4 * kPi * (v * v));
        fc[q] = v / c_cgs;
        krxr[q] = kappa_r * xr[q];
        omxr[q] = 1.0 - xr[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (int j = 0; j < NS - 1; ++j) {
        const double t = tsl[j], dt = tsl[j + 1] - t;
        const double edotr = 2.1e10 * eth[j] * tpw[j];
        const double en = enl[j];
        const double inv_t = 1.0 / t, inv_t2 = inv_t * inv_t;
        double part = 0.0, best = dinf();
        int besti = NL;
#pragma unroll
        for (int q = 0; q < LPL; ++q) {
            const int i = lane + 64 * q;
            if (i < NL) {
                const double xn = xn0[q] * en;
                const double edot = 3.2e14 * xn + edotr;
                const double kappa = 0.4 * (omxr[q] - xn) + krxr[q];
                const double tdiff = (kappa * fa[q]) * inv_t;
                const double tau = (kappa * fb[q]) * inv_t2;
                const double lum_j = ene[q] / (tdiff + t * fc[q]);
                part += lum_j * dmm[q];
                ene[q] += dt * (edot - (ene[q] * inv_t) - lum_j);
                const double dtau = fabs(tau - 1);
                if (dtau < best) { best = dtau; besti = i; }
            }
        }
        part = wave_sum(part);
        wave_argmin(best, besti);
        if (lane == 0) { lsum[j] = part; rph[j] = vml[besti] * t; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // effective temperature; non-finite entries filled as autocomplete_data(..., "linear") does
    for (int j = lane; j < NS; j += 64) {
        const double ltot = fabs(lsum[j] / 1e20 / 1e20);
        tobs[j] = 1e10 * pow(ltot / (4 * kPi * (rph[j] * rph[j]) * sigSB), 0.25);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int nfin = 0;
    for (int j = lane; j < NS; j += 64) nfin += (tobs[j] - tobs[j] == 0.0) ? 1 : 0;
    nfin = (int)wave_sum((double)nfin);
    for (int j = lane; j < NS; j += 64) {
        double tv = tobs[j];
        if (nfin < 2) {
            tv = dinf();
        } else if (!(tv - tv == 0.0)) {
            int jl = j - 1, jr = j + 1;
            while (jl >= 0 && !(tobs[jl] - tobs[jl] == 0.0)) --jl;
            while (jr < NS && !(tobs[jr] - tobs[jr] == 0.0)) ++jr;
            const double x = P.st[j];
            if (jl >= 0 && jr < NS) {
                tv = lerp_np(x, P.st[jl], P.st[jr], tobs[jl], tobs[jr]);
            } else if (jl < 0) {           // left of the first finite node: slope of the first two
                int j1 = jr + 1;
                while (j1 < NS && !(tobs[j1] - tobs[j1] == 0.0)) ++j1;
                tv = tobs[jr] + (tobs[j1] - tobs[jr]) / (P.st[j1] - P.st[jr]) * (x - P.st[jr]);
            } else {                       // right of the last finite node: slope of the last two
                int j0 = jl - 1;
                while (j0 >= 0 && !(tobs[j0] - tobs[j0] == 0.0)) --j0;
                tv = tobs[jl] + (tobs[jl] - tobs[j0]) / (P.st[jl] - P.st[j0]) * (x - P.st[jl]);
            }
        }
        if (tv <= 0.0) tv = dnan();
        double inv = 1.0 / tv;
        if (!(inv - inv == 0.0)) inv = dinf();
        eth[j] = inv;                      // reuse: 1/T per node
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // blackbody AB magnitudes per filter (host-frame frequency nu_0 * (1 + z))
    for (int f = 0; f < M; ++f) {
        const double nu = P.nu0[f] * (1 + z);
        int npos = 0;
        for (int j = lane; j < NS; j += 64) {
            double ex = h_cgs * nu * eth[j] / kb;
            if (ex > 700) ex = 700;        // np.clip(., None, 700); NaN passes through
            const double F = 2.0 * h_cgs / (c_cgs * c_cgs) * (nu * nu * nu) / expm1(ex) * rph[j] * rph[j] / (D10pc * D10pc);
            tpw[j] = F;
            npos += (F > 0) ? 1 : 0;
        }
        npos = (int)wave_sum((double)npos);
        double* dst = lc + ((size_t)b * M + f) * NS;
        for (int j = lane; j < NS; j += 64) {
            const double F = tpw[j];
            double mag = dinf();
            if (npos < 2) mag = dnan();
            else if (F > 0) mag = -2.5 * log10(F) + (-48.6);
            dst[j] = mag;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Detector-frame transform of supplied source-frame curves (gen_detector_lc, model.py:352-404)
__global__ void lc_to_detector(const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B,
                               const long ld, const double* __restrict__ lc, double* __restrict__ tobs_out,
                               double* __restrict__ mag_out) {
    const EmDev& P = *Pp;
    const long b = blockIdx.x;
    if (b >= B) return;
    __shared__ double praw[8], scal[8];
    __shared__ int nfin_s;
    const int NS = P.NS, M = P.M;
    if (threadIdx.x == 0) { double chk; sample_scalars(P, theta + b * ld, praw, scal, chk); }
    __syncthreads();
    const double ebv = scal[S_EBV];
    for (int m = 0; m < M; ++m) {
        if (threadIdx.x == 0) nfin_s = 0;
        __syncthreads();
        const double* cur = lc + ((size_t)b * M + m) * NS;
        int n = 0;
        for (int j = threadIdx.x; j < NS; j += blockDim.x) { const double v = cur[j]; n += (v - v == 0.0) ? 1 : 0; }
        atomicAdd(&nfin_s, n);
        __syncthreads();
        const double ext = P.has_ebv ? extinction_mag(P.ext_law, P.ebv_coeff[m], scal[S_ZP1], ebv) : 0.0;
        for (int j = threadIdx.x; j < NS; j += blockDim.x) {
            double v = cur[j];
            if (ext != 0.0) v = v + ext;
            v = (v + scal[S_DMOD]) + scal[S_RC];
            mag_out[((size_t)b * M + m) * NS + j] = nfin_s >= 2 ? v : dinf();
            if (m == 0) tobs_out[b * NS + j] = P.st[j] * scal[S_ZP1] + scal[S_TS];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// Pre-pass of the lean task for the Pei-1992 extinction law: ext_mag[b][m] for the whole batch, one thread per parameter
// vector (the law costs ~800 instructions per sample and filter -- six terms with three divisions each, 10^x, log10 -- which
// inside the likelihood kernel would be paid per datum slot).  Same sample_scalars as the kernel's prologue => same z.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ext_prepass_kernel(const EmDev* __restrict__ Pp, const double* __restrict__ theta,
                                                          const long B, const long ld, double* __restrict__ ext_tab) {
    const EmDev& P = *Pp;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (parameter vector, model filter)
    const long b = idx / P.M;
    const int m = (int)(idx - b * P.M);
    if (b >= B) return;
    double praw[NMMA_MAX_PARAMS], scal[8], chk;
    sample_scalars(P, theta + b * ld, praw, scal, chk);
    ext_tab[idx] = P.has_ebv ? extinction_mag(P.ext_law, P.ebv_coeff[m], scal[S_ZP1], scal[S_EBV]) : 0.0;
}

}  // namespace nmma

#include "em_api.inc"
