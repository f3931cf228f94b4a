#pragma once
// em_common.h -- what the EM translation units share on the device side: the MLP on the matrix cores, the role hand-off,
// scalar helpers and the LDS layouts of em_logl.  (Kernel overview of the whole EM path:)
//
//
// em_logl<R, KP, NMW, NVW, FAST>  (the hot path: nmma_em_loglike)
//   One workgroup owns a tile of TS = 16*R parameter vectors and walks the work items
//   (observed filter, source model filter) of the likelihood.  The waves are specialised:
//     NMW "MFMA waves": the surrogate MLP, one continuous stream of weight records over all items
//                x = (theta - pmin)/(pmax - pmin)          lightcurve_generation.py:193-194
//                c = Dense(relu)(x) -> Dense  (fp32)       lightcurve_generation.py:198
//     NVW "likelihood waves": everything downstream of the coefficients (fp64)
//                mag = (VA[:, :NC] @ c)*(maxs-mins)+mins   lightcurve_generation.py:214-216
//                stage-1 lerp onto sample_times, +inf out  lightcurve_generation.py:177
//                t_obs = t*(1+z)+timeshift, app = mag+ext+distmod-2.5log10(1+z)  model.py:374-404
//                stage-2 lerp onto the data epochs         em_likelihood.py:313-335
//                truncated-Gaussian / logsf terms, sum     em_likelihood.py:224-256, :337-352
//   so the f32 MFMA pipe and the f64 VALU pipe of every SIMD work concurrently.  The roles hand
//   items over through LDS counters (no workgroup barrier after the first one): partial sums and,
//   in FAST mode, the item's basis rows travel through a ring of LDS slots; the likelihood waves
//   claim (item, sample group) tasks from a shared counter and the MFMA waves join them when their
//   stream is done.  The final sum over filters and the floor (core/base.py:82, :180) happen in the
//   same launch.  FAST = every item qualifies for the straight-line task (see EmDev::all_fast);
//   otherwise the generic item phase with every reference branch runs in lock-step over the items.
//
//   MLP on the matrix cores: both Dense layers chained without a transpose -- layer 1
//   produces H^T[hidden 16 x sample 16] whose accumulator registers ARE the B operands of
//   layer 2 (C^T[coef 16 x sample 16] += W2^T[coef x 4 hidden] H^T).  Each MFMA wave owns
//   a contiguous run of hidden units and streams its pre-swizzled weight records
//   straight from L2 into a ring of VGPRs PF records deep (no LDS: nothing is shared
//   between waves).  Hidden units are always reduced as NSLICE = 8 partial sums in slice
//   order, so the fp32 result does not depend on the launch geometry.
//
//   Downstream: lane groups walk the ragged data of a filter; every datum brackets its
//   epoch on the redshifted grid and reconstructs ONLY the light-curve nodes it
//   interpolates between (2, or 4 when sample_times differ from the SVD grid) -- the
//   same arithmetic per node as the dense reconstruction.  Filters with so many points that
//   this reconstructs more rows per sample than the sample grid has nodes take the dense task
//   instead (FASTM = 6): the four tasks of (item, 16 samples) reconstruct every node on the fp64
//   matrix cores into an LDS buffer and a datum reads its two node magnitudes.  Small batches
//   are launched one workgroup per (tile, observed band); the band that finishes a tile last
//   adds the bands in the fused epilogue's order (release / acquire at agent scope).
//
// em_fused<MODE, R, WPB, KP>  (auxiliary outputs: coefficients, full light curves for
//   gen_detector_lc) shares the MLP scheme with all waves on the MFMA pipe first.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "em_device.h"
#include "stack2_tab.h"
#include "walk_device.h"
#include "em_math.h"
#include "nmma_common.h"

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

// relu on an MFMA result as ONE integer VALU op: for IEEE-754 bit patterns max_i32(bits, 0)
// is x for x >= +0 and +0 for every negative value (and -0).  A float max costs two ops
// (hipcc canonicalises MFMA outputs first), and in the one-wave-per-SIMD MLP loop every
// filler instruction beyond ~5 per MFMA gap delays the next MFMA issue.  (NaN inputs are
// caught before the MLP: S_BAD.)
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// Opaque identity: stops InstCombine from folding phi(load, load) into load(phi(addr)),
// which would move every prefetched weight load back to its use (no latency hiding).
__device__ __forceinline__ void opaque(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque(double& v) { asm volatile("" : "+v"(v)); }

// Hidden units are always split into NSLICE partial sums added in slice order, so the
// fp32 result does not depend on the launch geometry (R, WPB) chosen for a batch size.
constexpr int NSLICE = 8;
// zero records appended to every model filter's weight stream (deepest prefetch ring + 1)
constexpr int NPAD_REC = 9;
// row stride (floats) of the LDS partial-sum tiles: 16 coefficients + 1 pad (bank spread)
constexpr int PSTR = 17;
// per-model-filter static tables staged in LDS by em_logl (LDS-DMA, 1 KiB per wave-instruction)
constexpr int TAB_MAX_BYTES = 40 * 1024;
// 32-bit words of one work-item descriptor of em_logl (see em_device.h: ItemDesc)
constexpr int ITEM_WORDS = 24;

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }

enum ScalIdx { S_ZP1 = 0, S_TS = 1, S_DMOD = 2, S_RC = 3, S_EBV = 4, S_BAD = 5, S_IZP1 = 6 };

// ---------------------------------------------------------------------------------------
// Surrogate MLP on the f32 MFMA pipe for NSL consecutive hidden slices of one wave.
//   rec   : first weight record of this wave's run (records are contiguous per wave)
//   xB    : layer-1 B operands, lane l holds x[sample rb*16 + (l&15)][param 4*kp + (l>>4)]
//   part  : LDS [NSLICE][R][16 sample][PSTR] partial sums (coef fastest); slices slice0 .. slice0+NSL-1
// PF records are kept in flight in a register ring (loads of record g+PF are issued while
// record g is consumed); HBS (records per slice) must be a multiple of PF.
// ---------------------------------------------------------------------------------------
template <int R, int KP, int PF, int NSL>
__device__ __forceinline__ void mlp_slices(gcf32p rec, const float (&xB)[R][KP], const int HBS, const int lane,
                                           float* __restrict__ part, const int slice0) {
    constexpr int RECF = rec_floats(KP);
    constexpr int RECB = RECF * 4;
    // Buffer loads: the per-lane byte offsets are loop-invariant VGPRs and the record offset
    // is ONE scalar, bumped by SALU -- no VALU address arithmetic competes with the MFMAs.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(uintptr_t)rec, 0, (NSL * HBS + NPAD_REC) * RECB, 0x00020000);
    const int off_a2 = lane * 16;
    const int off_a1 = (256 + lane) * 4;
    const int off_b = (256 + 64 * KP + (lane >> 4) * 4) * 4;
    auto ld4 = [&](int voff, int soff) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
    };
    auto ld1 = [&](int voff, int soff) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    };
    f32x4 ra2[PF], rbias[PF];
    float ra1[PF][KP];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        ra2[u] = ld4(off_a2, u * RECB);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) ra1[u][kp] = ld1(off_a1 + kp * 256, u * RECB);
        rbias[u] = ld4(off_b, u * RECB);
    }
    // layer-1 pre-activations of record 0
    f32x4 d[R];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        d[rb] = rbias[0];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
            d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[0][kp], xB[rb][kp], d[rb], 0, 0, 0);
    }
    int soff = PF * RECB;   // byte offset of the next record to fetch (scalar)
#pragma unroll 1
    for (int sl = 0; sl < NSL; ++sl) {
        f32x4 acc[R][2];
#pragma unroll
        for (int rb = 0; rb < R; ++rb) { acc[rb][0] = f32x4{0, 0, 0, 0}; acc[rb][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll 1
        for (int i0 = 0; i0 < HBS; i0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int nu = (u + 1) % PF;
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
                    d[rb] = rbias[nu];
#pragma unroll
                    for (int kp = 0; kp < KP; ++kp)
                        d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[nu][kp], xB[rb][kp], d[rb], 0, 0, 0);
                }
                const f32x4 a2 = ra2[u];
                // refill slot u with the record PF ahead (NPAD_REC zero records pad every filter)
                ra2[u] = ld4(off_a2, soff);
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) ra1[u][kp] = ld1(off_a1 + kp * 256, soff);
                rbias[u] = ld4(off_b, soff);
                soff += RECB;
                // layer 2: C^T[coef][sample] += W2^T[coef][4 hidden] * H^T[4 hidden][sample]
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int rb = 0; rb < R; ++rb)
                        acc[rb][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[r], h[rb][r], acc[rb][r & 1], 0, 0, 0);
                // order inside the step: relu (VALU) | layer-1 MFMAs | refill loads | layer-2 MFMAs, so the
                // VALU->MFMA wait states are covered by the layer-1 MFMAs instead of s_nops; the fence
                // keeps every refill inside its own step (otherwise the scheduler sinks all PF refills to
                // the end of the unrolled body and the next iteration opens with s_waitcnt vmcnt(0))
                __builtin_amdgcn_sched_group_barrier(0x002, 4 * R, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, R * KP, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2 + KP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * R, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // partial C^T of this hidden slice -> LDS
        const int slice = slice0 + sl;
#pragma unroll
        for (int rb = 0; rb < R; ++rb) {
            const f32x4 s = acc[rb][0] + acc[rb][1];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                part[((slice * R + rb) * 16 + (lane & 15)) * PSTR + (lane >> 4) * 4 + r] = s[r];
        }
    }
}

// ---------------------------------------------------------------------------------------
// Role hand-off of em_logl through three LDS counters instead of workgroup barriers, so the
// two roles never wait for each other unless the data dependency is real:
//   sync[k]         += 1 by every MFMA wave once its partial sums of item k are in LDS;
//   sync[W + 1 + j] += 1 by every likelihood wave after its phase j - 1 (j = 0: prologue, j = k + 1: item k;
//                   fast mode: once per finished task of item k);
//   sync[2W + 2], sync[2W + 3]: prologue staging done / next task to claim;  sync[2W + 4 + k]: rows of item k staged.
// One counter per item/phase (never reset): waves of a role may run ahead of each other, so a
// running total could be reached by early signals of the next item.
// LDS instructions of one wave execute in order, so "data writes, then counter add" by the
// producer and "counter read, then data reads" by the consumer need no further fence.
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) int* lds_ip;
// Debug stamps and watchdog words are written through GLOBAL-address-space pointers: a flat store anywhere in
// the record loop nest makes the compiler guard every ring access with s_waitcnt vmcnt(0).
typedef __attribute__((address_space(1))) int* g_ip;
typedef __attribute__((address_space(1))) long long* g_llp;
// Measurement builds (tools/levers_r04.sh): -DNMMA_SYNC_SLEEP=<n> sets the s_sleep argument of a polling wave (64 n cycles),
// -DNMMA_SYNC_WAKEUP makes every signal wake the workgroup's sleeping waves (s_wakeup), so that long sleeps cost no latency.
#ifndef NMMA_SYNC_SLEEP
#define NMMA_SYNC_SLEEP 6
#endif
__device__ __forceinline__ void sync_signal(int* cnt, const int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((lds_ip)cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef NMMA_SYNC_WAKEUP
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");
#endif
}
// A wait that does not complete within ~2^20 polls (tens of milliseconds; a healthy launch needs
// microseconds) records where it was stuck in the handle's watchdog words and gives up, so that a
// protocol bug surfaces as an error code from the C ABI instead of a hung GPU.
// Set by a wait that gave up (one word of static LDS per workgroup, zeroed before the workgroup's first barrier): the
// epilogue then writes the floor for the whole tile instead of whatever the unfinished hand-off left behind.
__shared__ int g_wd_trip;

template <int SLEEP = NMMA_SYNC_SLEEP>
__device__ __forceinline__ void sync_wait(int* cnt, const int target, int* watchdog_generic = nullptr, const int code = 0) {
    g_ip watchdog = (g_ip)(uintptr_t)watchdog_generic;
    // Every VALU instruction of a polling wave takes an issue slot from the MFMA waves of its SIMD (a poll is
    // v_mov + ds_read + v_cmp): sleep ~400 cycles between polls so that waiting costs next to nothing.
    int spins = 0;
    while (__hip_atomic_load((lds_ip)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > (1 << 18)) {
            if ((threadIdx.x & 63) == 0) g_wd_trip = 1;
            if (watchdog_generic != nullptr && (threadIdx.x & 63) == 0) {
                watchdog[0] = 1; watchdog[1] = code; watchdog[2] = (int)blockIdx.x * 64 + (int)(threadIdx.x >> 6);
                watchdog[3] = __hip_atomic_load((lds_ip)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * 65536 + target;
            }
            break;
        }
    }
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------
// sigma_tot per (datum, sample) of the lean tasks with a sampled systematic: 1 / sigma and ln sigma from s2 = sigma_data^2 + e^2
// without the library's sqrt, division and log (~120 VALU instructions per datum in the task loop, most of them the
// double-double arithmetic of a correctly rounded log): v_rsq_f64 + two Newton steps (<= 2 ulp), and the classic
// argument reduction x = m 2^k, m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m - 1)/(m + 1)) with the degree-14 minimax polynomial
// in s^2 of Sun's fdlibm e_log.c (< 1 ulp; the reference's numpy uses the same family).  Inputs that are not positive and finite
// give garbage that every caller masks (upper limits, non-finite sigma).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double rsqrt_pos(const double s2) {
    double y = __builtin_amdgcn_rsq(s2);
#pragma unroll
    for (int i = 0; i < 2; ++i) { const double c = (s2 * y) * y; y = y * fma(c, -0.5, 1.5); }
    return y;
}
__device__ __forceinline__ double log_pos(const double x) {
    int k = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);                 // x = m 2^k, m in [1/2, 1)
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m; k = lo ? k - 1 : k;
    const double f = m - 1.0, dk = (double)k, d = 2.0 + f;
    double rc = __builtin_amdgcn_rcp(d);
    rc = fma(fma(-d, rc, 1.0), rc, rc);
    rc = fma(fma(-d, rc, 1.0), rc, rc);
    const double sq = f * rc, z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double hfsq = 0.5 * f * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (sq * (hfsq + (t2 + t1)) + dk * 1.90821492927058770002e-10)) - f);
}

// exp(x) for x <= 0 (or -inf): x = k ln2 + r, |r| <= ln2 / 2, Taylor polynomial of degree 13 (remainder 4e-18), ldexp.  For the
// stacking kernel, where the library's exp and log were the whole cost of a node.
__device__ __forceinline__ double exp_neg(const double x) {
    if (!(x > -745.2)) return 0.0;
    const double kf = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, x);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                         // 1 / 13!
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.3888888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)kf);
}

// ---------------------------------------------------------------------------------------
// MFMA role of em_logl: ONE continuous stream of weight records over all work items.
// Wave `wave` of NMW owns NSL = NSLICE/NMW hidden slices of every item; its records of
// consecutive items are chained into a single prefetch ring (the refills issued during the
// last PF records of an item already fetch the first PF records of the next one), so the
// L2 latency is paid once per launch instead of once per item.  Layer-1 pre-activations
// run one record ahead and therefore switch to the next item's normalised inputs on the
// last record of an item.  Partial sums of item k go to buffer k % NBUF of `part`; the role
// only waits for the likelihood role when that buffer still holds item k - NBUF.  The role issues
// nothing but buffer loads into VGPRs: an LDS-DMA (or any flat load) inside this loop makes the compiler
// guard every ring access with s_waitcnt vmcnt(0), i.e. one exposed L2 round trip per 8 records.
// ---------------------------------------------------------------------------------------
// (the ring depth NBUF is a launch parameter, LdsW::nbuf: what fits in LDS, at most 4)
// prologue staging of em_logl: theta columns per row and cosmology-grid nodes kept in LDS
constexpr int STAGE_COLS = 24, STAGE_COSMO = 256;
// (the fused MCMC step parks [tot 16 | 5 x 2 rounds x 64 | 2 x 16 doubles | 6 x 16 ints | prior table] in the staging area of a 16-sample tile)
static_assert((16 * STAGE_COLS + 2 * STAGE_COSMO) * 8 >= (16 + 5 * 2 * 64 + 2 * 16 + 3 * 16) * 8 + 8 * 40, "parked walk state (8 lanes per chain; 16: lds_layout_logl_try)");
// fast mode: most (item, sample group) tasks of one tile whose index -> (item, chunk) map is kept in LDS
constexpr int TMAP_MAX = 512;
constexpr int DENSE_NBUF = 2, DENSE_STRIDE = 17;   // dense lean task: node-magnitude buffers of 16 samples, row stride in doubles (odd: bank spread)
constexpr int SPLIT_COUNTER_BYTES = 64 * 1024;    // split launch: one arrival counter per tile, in front of the band workspace
// fast mode: most photometry points (all filters) staged in LDS as [t | m | 1/sigma | log sigma]
constexpr int DAT_MAX = 2560;
// most points of one filter the lean task takes (passes of 32 per group of 16 lanes; beyond this the extended task's wider
// groups win)
#ifndef NMMA_LEAN_NF_MAX
#define NMMA_LEAN_NF_MAX 2560
#endif
constexpr int LEAN_NF_MAX = NMMA_LEAN_NF_MAX;

// SKIPNULL (the combined-model flavours): the first EmDev::n_items_null work items are NULL filters -- bands the surrogate has no network for
// (nmma_em_config::null_filters; their basis rows give +inf whatever the coefficients).  The record stream does not walk them: this role
// publishes zero partial sums for them before it starts, so their tasks run beside the stream of the real items.
template <int R, int KP, int PF, int NMW, int NVW, bool FAST, bool SKIPNULL, class LateX>
__device__ __forceinline__ void mfma_role(const EmDev& P, const double (&xraw)[R][KP], double* xnl, const int wave, const int lane,
                                          float* __restrict__ part, const int NBUF, int* sync,
                                          long long* __restrict__ dbg_generic, LateX&& late_xraw) {
    g_llp dbg = (g_llp)(uintptr_t)dbg_generic;
    constexpr int RECF = rec_floats(KP);
    constexpr int RECB = RECF * 4;
    constexpr int NSL = NSLICE / NMW;
    constexpr int TS = 16 * R;
    const int W = P.n_items, NP = P.NP;
    const int HBS = P.HB / NSLICE;
    const int CPS = HBS / PF;                      // chunks per slice
    gci32p items = as_global(P.items);
    gci32p idesc = as_global(reinterpret_cast<const int*>(P.item_desc));   // words 22, 23 of a descriptor = ntask[R - 1]
    gcf64p pmin = as_global(P.pmin), pinv = as_global(P.pinv);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(uintptr_t)P.wrec, 0, P.wrec_bytes, 0x00020000);
    const int off_a2 = lane * 16;
    const int off_a1 = (256 + lane) * 4;
    const int off_b = (256 + 64 * KP + (lane >> 4) * 4) * 4;
    auto ld4 = [&](int voff, int soff) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
    };
    auto ld1 = [&](int voff, int soff) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    };
    // byte offset (from wrec) of this wave's first record of item k
    auto item_base = [&](int k) -> int {
        const int m = items[4 * k + 2];
        return __builtin_amdgcn_readfirstlane((m * (P.HB + NPAD_REC) + wave * NSL * HBS) * RECB);
    };
    // Normalisation constants (pmin, 1/pspan) of every model filter go to LDS once: every MFMA wave
    // writes the same values and reads them back after its own writes (in-order LDS), so switching
    // items costs LDS latency instead of an L2 round trip in the middle of the record stream.
    for (int j = lane; j < P.M * NP; j += 64) { xnl[2 * j] = pmin[j]; xnl[2 * j + 1] = pinv[j]; }
    // normalised layer-1 inputs of item k: lane l holds x[sample rb*16 + (l&15)][param 4*kp + (l>>4)]
    auto load_x = [&](int k, float (&x)[R][KP]) {
        const int m = items[4 * k + 2];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const int p = 4 * kp + (lane >> 4);
            const double mn = (p < NP) ? xnl[2 * (m * NP + p)] : 0.0, iv = (p < NP) ? xnl[2 * (m * NP + p) + 1] : 0.0;
#pragma unroll
            for (int rb = 0; rb < R; ++rb) x[rb][kp] = (float)((xraw[rb][kp] - mn) * iv);
        }
    };

    const int Wn = SKIPNULL ? P.n_items_null : 0;        // (null items come first: nmma_em_create orders them so)
    if constexpr (SKIPNULL) {
        for (int k = 0; k < Wn; ++k) {
            float* pk = part + (k % NBUF) * (NSLICE * TS * PSTR);
            if (k >= NBUF) sync_wait(sync + W + 1 + (k - NBUF + 1), FAST ? idesc[(k - NBUF) * ITEM_WORDS + 22 + (R - 1)] : NVW, P.watchdog, 100 + k);
#pragma unroll 1
            for (int sl = 0; sl < NSL; ++sl) {
                const int slice = wave * NSL + sl;
#pragma unroll
                for (int rb = 0; rb < R; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[((slice * R + rb) * 16 + (lane & 15)) * PSTR + (lane >> 4) * 4 + r] = 0.f;
            }
            sync_signal(sync + k, lane);      // item k published: zero coefficients
        }
    }
    int base = item_base(Wn);
    f32x4 ra2[PF], rbias[PF];
    float ra1[PF][KP];
    // The ring holds records g .. g+PF-2 when record g is consumed; the step that consumes slot g % PF
    // refills the slot consumed ONE STEP EARLIER with record g+PF-1.  Every reader of that slot has been
    // issued before the load, so the load writes the slot's own registers (loading into the slot being
    // consumed makes hipcc double-buffer the whole ring: 16 v_mov_b64 and an s_waitcnt vmcnt(0) per chunk).
    ra2[PF - 1] = f32x4{0, 0, 0, 0}; rbias[PF - 1] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) ra1[PF - 1][kp] = 0.f;
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
        ra2[u] = ld4(off_a2, base + u * RECB);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) ra1[u][kp] = ld1(off_a1 + kp * 256, base + u * RECB);
        rbias[u] = ld4(off_b, base + u * RECB);
    }
    late_xraw();          // (measurement build -DNMMA_DBG_PRELOAD_FIRST: theta is read only now, behind the ring's first loads)
    float xB[R][KP], xN[R][KP];
    load_x(Wn, xB);
    f32x4 d[R];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        d[rb] = rbias[0];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
            d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[0][kp], xB[rb][kp], d[rb], 0, 0, 0);
    }

#pragma unroll 1
    for (int k = Wn; k < W; ++k) {
        if (dbg && blockIdx.x == 0 && wave == 0 && lane == 0) dbg[2 * k] = clock64();
        const int nbase = (k + 1 < W) ? item_base(k + 1) : base;     // last item: harmless re-read
        if (k + 1 < W) load_x(k + 1, xN);
        else {
#pragma unroll
            for (int rb = 0; rb < R; ++rb)
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) xN[rb][kp] = xB[rb][kp];
        }
        float* pk = part + (k % NBUF) * (NSLICE * TS * PSTR);
        // ring slot k % NBUF is reused: item k - NBUF must be consumed before the first write into it
        // (partial sums or staged rows, whichever comes first); one signal per wave, or per task in fast mode
        bool slot_free = k < NBUF;
        auto wait_slot = [&]() {
            if (!slot_free) {
                sync_wait(sync + W + 1 + (k - NBUF + 1), FAST ? idesc[(k - NBUF) * ITEM_WORDS + 22 + (R - 1)] : NVW, P.watchdog, 100 + k);
                slot_free = true;
            }
        };
        int soff = base + (PF - 1) * RECB;        // record fetched by the next refill (PF-1 of this item are in the ring)
#pragma unroll 1
        for (int sl = 0; sl < NSL; ++sl) {
            f32x4 acc[R][2];
#pragma unroll
            for (int rb = 0; rb < R; ++rb) { acc[rb][0] = f32x4{0, 0, 0, 0}; acc[rb][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll 1
            for (int c = 0; c < CPS; ++c) {
                const bool last_chunk = (sl == NSL - 1) && (c == CPS - 1);
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int nu = (u + 1) % PF;
                    f32x4 h[R];
#pragma unroll
                    for (int rb = 0; rb < R; ++rb) {
#ifdef NMMA_DBG_NORELU
                        h[rb] = d[rb];
#else
                        h[rb][0] = relu1(d[rb][0]); h[rb][1] = relu1(d[rb][1]);
                        h[rb][2] = relu1(d[rb][2]); h[rb][3] = relu1(d[rb][3]);
#endif
                    }
                    // layer 1 of the NEXT record; the record after the last one of an item is the next item's
#pragma unroll
                    for (int rb = 0; rb < R; ++rb) {
                        d[rb] = rbias[nu];
#pragma unroll
                        for (int kp = 0; kp < KP; ++kp) {
                            const float xv = (u == PF - 1 && last_chunk) ? xN[rb][kp] : xB[rb][kp];
                            d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[nu][kp], xv, d[rb], 0, 0, 0);
                        }
                    }
                    const f32x4 a2 = ra2[u];
                    // refill the slot consumed one step earlier; in the last chunk of an item step 0 still
                    // fetches the item's last record, steps 1.. fetch the first PF-1 records of the next item
                    const int pu = (u + PF - 1) % PF;
                    if (u == 1 && last_chunk) soff = nbase;
#ifndef NMMA_DBG_NOLOAD
                    ra2[pu] = ld4(off_a2, soff);
#pragma unroll
                    for (int kp = 0; kp < KP; ++kp) ra1[pu][kp] = ld1(off_a1 + kp * 256, soff);
                    rbias[pu] = ld4(off_b, soff);
#endif
                    soff += RECB;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int rb = 0; rb < R; ++rb)
                            acc[rb][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[r], h[rb][r], acc[rb][r & 1], 0, 0, 0);
#ifndef NMMA_DBG_NORELU
                    __builtin_amdgcn_sched_group_barrier(0x002, 4 * R, 0);
#endif
                    __builtin_amdgcn_sched_group_barrier(0x008, R * KP, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * R, 0);
#ifndef NMMA_DBG_NOLOAD
                    __builtin_amdgcn_sched_group_barrier(0x020, 2 + KP, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int slice = wave * NSL + sl;
            wait_slot();
#pragma unroll
            for (int rb = 0; rb < R; ++rb) {
                const f32x4 s = acc[rb][0] + acc[rb][1];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pk[((slice * R + rb) * 16 + (lane & 15)) * PSTR + (lane >> 4) * 4 + r] = s[r];
            }
        }
        sync_signal(sync + k, lane);      // item k published
        base = nbase;
#pragma unroll
        for (int rb = 0; rb < R; ++rb)
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) xB[rb][kp] = xN[rb][kp];
        if (dbg && blockIdx.x == 0 && wave == 0 && lane == 0) dbg[2 * k + 1] = clock64();
    }
}

// ---------------------------------------------------------------------------------------
// Sum of a double over lane groups of G = 16, 32 or 64 lanes with DPP moves (VALU only).
// The total lands in every lane of the group's LAST 16-lane row (lanes G-16 .. G-1).
// Fixed addition order => deterministic.
// ---------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (no "old" value: rows a row_mask leaves out are undefined -- they never hold the group's total, see group_sum --
    //  so no v_mov is needed to initialise the destination)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double group_sum(double v, const int G) {
    v += dpp_mov_f64<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141, 0xf>(v);    // row_half_mirror
    v += dpp_mov_f64<0x140, 0xf>(v);    // row_mirror: every lane holds its row's sum
    if (G >= 32) v += dpp_mov_f64<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    if (G >= 64) v += dpp_mov_f64<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return v;
}

// Per-sample scalars of em_parameter_setup (model.py:288-303) + conversions, for the
// sample whose theta row is `row`; written to scal[8] / praw[8] of that sample.
__device__ __forceinline__ void sample_scalars(const EmDev& P, const double* row, double* praw, double* scal,
                                               double& chk, const double* dist_grid = nullptr,
                                               const double* z_grid = nullptr) {
    if (dist_grid == nullptr) { dist_grid = P.dist_grid; z_grid = P.z_grid; }
    for (int p = 0; p < NMMA_MAX_PARAMS; ++p) praw[p] = (p < P.NP) ? apply_slot(P.model_param[p], row) : 0.0;
    const double d_l = apply_slot(P.lumdist, row);
    double z = 0.0;
    if (P.redshift_mode == NMMA_Z_SLOT) {
        z = apply_slot(P.redshift, row);
    } else if (P.redshift_mode == NMMA_Z_GRID) {
        // (a sampled Hubble constant: the grid belongs to the reference H0, distances scale as 1 / H0)
        // (with has_h0 the table holds z / d_L, a nearly constant function: linear interpolation of it is exact to ~1e-11
        //  where interpolating z itself on 256 nodes is off by 1e-7 -- log L moves by 1e4 per unit redshift)
        const double d_eff = P.has_h0 ? d_l * apply_slot(P.hubble, row) * P.inv_h0_ref : d_l;
        z = interp_np(d_eff, dist_grid, z_grid, P.n_cosmo, z_grid[0], z_grid[P.n_cosmo - 1]);
        if (P.has_h0) z *= d_eff;
    }
    scal[S_ZP1] = 1 + z;
    scal[S_IZP1] = 1.0 / (1 + z);   // only seeds the bracket guess (exactly re-checked)
    scal[S_TS] = apply_slot(P.timeshift, row);
    scal[S_DMOD] = distance_modulus(d_l);
    scal[S_RC] = redshift_correction(z);
    scal[S_EBV] = P.has_ebv ? apply_slot(P.ebv, row) : 0.0;
    chk = d_l + z + scal[S_TS] + scal[S_EBV];
    for (int p = 0; p < P.NP; ++p) chk += praw[p];
}

// =======================================================================================
// em_logl: the hot path
// =======================================================================================
// workgroup of em_logl: 4 MFMA-role waves + NVW VALU-role waves
constexpr int logl_threads(int NMW, int NVW) { return 64 * (NMW + NVW); }

struct LdsW {
    int32_t praw, scal, stl, part, chi, gp, bad, cdl, itab, est, tab, sync, xn, stage, tmap, dat, epar, exttab, total;
    int32_t nodes;      // dense lean task (em_logl<.., 6>): DENSE_NBUF buffers of [dense_rows][DENSE_STRIDE] fp64 node magnitudes of 16 samples
    int32_t nf_max;
    int32_t nbuf;       // depth of the partial-sum ring (items the MFMA role may run ahead)
    // (combined-model flavour, em_logl<.., 7>: `nodes` is the offset of the two-model flux-sum table, stack2_tab.h, STACK2_LDS_BYTES --
    //  the flavour is never dense, and a field of its own would move the kernel arguments behind this struct for every flavour)
};

// Per-call operands of em_logl's combined-model flavour (FASTM 7), passed by value behind the other kernel arguments:
// the second transient's source-frame curves [B][M][NS] on the handle's sample_times and model filters, and the rows for which
// a sub-model delivered no light curve (or NULL).
// gap_rows[B]: written by the kernel for every row -- 1 = the row met an interior non-finite node of lc2 and must be re-evaluated
// by the materialising kernels (nmma_em_loglike_stack2 launches them restricted to those rows), 0 = out[b] is final.
// completed != 0 (NMMA_STACK2_COMPLETED): lc2 came out of nmma_lc_regrid -- its non-finite nodes are leading / trailing only and mean "no flux".
struct EmAux { const double* lc2; const unsigned char* bad_rows; unsigned char* gap_rows; int completed; };
struct EmNoAux {};      // (what every other flavour takes in that place: their kernel arguments stay as they were)
template <int FASTM> struct em_aux_of { typedef EmNoAux type; };
template <> struct em_aux_of<7> { typedef EmAux type; };
template <> struct em_aux_of<8> { typedef EmAux type; };

// One candidate layout: `nbuf` ring slots, photometry staged or not.
// Dynamic LDS a launch may ask for: the 160 KiB of a CU minus the kernel's static words (g_wd_trip), rounded down to the
// 1-KiB granule the layouts use -- a layout of exactly 160 KiB is refused by hipFuncSetAttribute.
constexpr int LDS_DYNAMIC_MAX = 159 * 1024;

__host__ inline LdsW lds_layout_logl_try(int R, int NS, int nf_avg_max, int tab_bytes, int tab_fast_bytes, int n_items, int M,
                                         int NP, int all_fast, int n_data, int n_sys_slots, int nbuf, bool stage_dat, int ext_rows = 0,
                                         int dat_point_bytes = 32, int dense_rows = 0, int stack_bytes = 0, int walk_lanes = 0) {
    const int TS = 16 * R;
    const bool bracket_lookup = NS < 0;        // (NS < 0: unequally spaced sample_times -- the lean tasks' lookup table sits behind the grid)
    NS = NS < 0 ? -NS : NS;
    LdsW L{};
    int off = 0;
    L.nbuf = nbuf;
    L.praw = off; off = align16(off + TS * 8 * 8);
    L.scal = off; off = align16(off + TS * 8 * 8);
    L.stl = off;  off = align16(off + 2 * NS * 8 + (bracket_lookup ? (BG_CELLS + 1) * 4 : 0));   // sample times | 1 / (t[j+1] - t[j]) | bracket lookup
    L.part = off; off = align16(off + L.nbuf * NSLICE * TS * PSTR * 4);
    L.chi = off;  off = align16(off + n_items * TS * 8);             // per item: [TS] minus-chi-square sums
    L.gp = off;   off = align16(off + n_items * TS * 8);
    L.sync = off; off = align16(off + (3 * n_items + 4 + 4 * n_items + 2) * 4);      // (+ produced / consumed counters per (item, 16 samples): dense task; + 2: fused MCMC step)
    // (prologue staging; afterwards the fused MCMC step parks its state there: 16 lanes per chain -- four rounds per tile -- need more)
    {
        const int stage = (TS * STAGE_COLS + 2 * STAGE_COSMO) * 8;
        const int parked = walk_lanes ? (TS + 5 * (TS / (64 / walk_lanes)) * 64 + 2 * TS + 3 * TS) * 8 + walk_lanes * 40 : 0;
        L.stage = off; off = align16(off + (stage > parked ? stage : parked));
    }
    L.tmap = off;  off = align16(off + (all_fast ? TMAP_MAX * 4 : 0));   // fast mode: task index -> (item << 8 | chunk)
    L.dat = (all_fast && stage_dat && n_data <= DAT_MAX) ? off : -1;     // fast mode: photometry [t | m | 1/sigma | log sigma]
    if (L.dat >= 0) off = align16(off + n_data * dat_point_bytes);     // (8: the epochs only -- item-staged photometry, EmDev::dat_in_tab)
    L.epar = off;  off = align16(off + (all_fast ? n_sys_slots * TS * 8 : 0));        // fast modes: sysv[slot][sample]
    L.exttab = off; off = align16(off + ext_rows * TS * 8);          // lean task with extinction: ext_mag[item][sample]
    L.xn = off;   off = align16(off + M * NP * 2 * 8);               // (pmin, 1/pspan) per model filter and parameter
    L.bad = off;  off = align16(off + (5 + (stack_bytes ? 1 : 0)) * TS * 4);   // bad[TS] (NaN terms) | badp[4][TS] (prologue parts) | combined model: gap[TS]
    L.cdl = off;  off = align16(off + (dense_rows ? 0 : 16 * 2 * 4 * 16 * 8));      // per wave (any role): 2 x 4 slots x 16 coefficients (the dense task has none)
    L.nodes = off; off = align16(off + DENSE_NBUF * dense_rows * DENSE_STRIDE * 8 + stack_bytes);      // (never both)
    L.itab = off; off = align16(off + n_items * ITEM_WORDS * 4);     // per-item descriptors
    L.nf_max = nf_avg_max;
    L.est = off;  off = align16(off + TS * nf_avg_max * 8);
    off = (off + 1023) / 1024 * 1024;
    L.tab = off;  off = align16(off + (all_fast ? L.nbuf * tab_fast_bytes : 2 * tab_bytes));   // fast: ring of [rows | b2]; generic: double buffer
    L.total = off;
    return L;
}

// Ring depth (items the MFMA role may run ahead): fast mode rings {partial sums, staged basis rows} per item and
// takes as many slots (at most 4) as fit the 160 KiB of LDS, giving up the photometry staging before the last
// slots; the generic path keeps 3 partial-sum buffers next to its double-buffered tables.
// (ring_max: NMMA_EM_RING=<n>, read at nmma_em_create -- an upper bound on the ring depth: a shallower ring leaves LDS to kernels
//  that share the CUs, e.g. RCCL's while a collective overlaps the likelihood)
__host__ inline LdsW lds_layout_logl(int R, int NS, int nf_avg_max, int tab_bytes, int tab_fast_bytes, int n_items, int M, int NP,
                                     int all_fast, int n_data, int n_sys_slots, int ext_rows = 0, int ring_max = 4, int dat_point_bytes = 32,
                                     int dense_rows = 0, int stack_bytes = 0, int walk_lanes = 0) {
    constexpr int LDS_MAX = LDS_DYNAMIC_MAX;
    int want = n_items < 1 ? 1 : (n_items < (all_fast ? 4 : 3) ? n_items : (all_fast ? 4 : 3));
    if (want > ring_max) want = ring_max < 1 ? 1 : ring_max;
    LdsW L{};
    for (int pass = 0; pass < 2; ++pass)
        for (int nbuf = want; nbuf >= (pass == 0 ? (want < 3 ? want : 3) : 1); --nbuf) {
            // (all_fast == 1, the lean task, reads the photometry from LDS only: never give the staging up)
            L = lds_layout_logl_try(R, NS, nf_avg_max, tab_bytes, tab_fast_bytes, n_items, M, NP, all_fast, n_data, n_sys_slots, nbuf, pass == 0 || all_fast == 1, ext_rows, dat_point_bytes, dense_rows, stack_bytes, walk_lanes);
            if (L.total <= LDS_MAX) return L;
        }
    return L;     // does not fit: the launch fails with an explicit error
}

// ---------------------------------------------------------------------------------------
// Flux sum of light-curve sets (stack_magnitudes, model.py:1486-1510): shared by the stacking / likelihood-from-curves kernels
// (em_kernels.hip) and by em_logl's combined-model flavour, which stacks a second transient's curves onto the kilonova's bracket nodes.
// ---------------------------------------------------------------------------------------
// The curve sets of one stacking call, passed by value (no device-side pointer table).
struct LcSets { const double* p[8]; };       // the curve sets of one stacking call, passed by value (no device-side pointer table)

// The two-model table into LDS (STACK2_LDS_BYTES at tab_lds); the caller synchronises the workgroup before lc_stack_node reads it.
__device__ __forceinline__ void stack2_stage(double* tab_lds, const int tid, const int n_threads) {
    for (int j = tid; j < STACK2_NINT * STACK2_ROW; j += n_threads) tab_lds[j] = kStack2Tab[j];
}

// Two models (kilonova + afterglow, the reference's combined models), both finite at this node -- all but a few nodes:
// mag = min(m0, m1) - g(|m0 - m1|), g(D) = 2.5 log10(1 + 10^(-0.4 D)) from a table of degree-10 polynomials on 64 intervals of
// [0, 40) mag (stack2_tab.h, tools/gen_softplus_table.py: 1.7e-15 mag from the direct formula; beyond 40 mag g < 3e-16).
// tab2: the table staged in LDS by the caller (stack2_stage) -- every lane reads its own row, six 16-byte reads; from global
// memory those gathers cost as much as the exp and the log in fp64 they replace (~150 vector instructions a node).
// false: a non-finite value -- the node takes lc_stack_node's general path (gap filling).
__device__ __forceinline__ bool stack2_fast(const double v0, const double v1, const double* tab2, double& r) {
    if (!((v0 - v0 == 0.0) && (v1 - v1 == 0.0))) return false;
    const double lo = v0 < v1 ? v0 : v1, D = fabs(v0 - v1);
    r = lo;
    if (D >= STACK2_DMAX) return true;
    const double sc = D * STACK2_INV_H;
    int idx = (int)sc;
    idx = idx > STACK2_NINT - 1 ? STACK2_NINT - 1 : idx;
    const double t = 2.0 * (sc - (double)idx) - 1.0;
    const double2* cf = reinterpret_cast<const double2*>(tab2) + idx * (STACK2_ROW / 2);
    const double2 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5];
    double p = fma(c0.x, t, c0.y);
    p = fma(p, t, c1.x); p = fma(p, t, c1.y);
    p = fma(p, t, c2.x); p = fma(p, t, c2.y);
    p = fma(p, t, c3.x); p = fma(p, t, c3.y);
    p = fma(p, t, c4.x); p = fma(p, t, c4.y);
    p = fma(p, t, c5.x);
    r = lo - p;
    return true;
}

// One node: vv[k] = model k's value at node g (already loaded).
template <int KM>
__device__ __forceinline__ double lc_stack_node(const EmDev& P, const LcSets& sets, const int n_models, const long g, const double* vv,
                                                const double* tab2 = nullptr) {
    const int NS = P.NS;
    if constexpr (KM == 2) {
        double r;
        if (n_models == 2 && tab2 != nullptr && stack2_fast(vv[0], vv[1], tab2, r)) return r;
    }
    const double ln10 = 2.302585092994046;
    double amax = -HUGE_VAL, terms[KM];
    bool any_nan = false;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        terms[k] = -HUGE_VAL;
        if (k >= n_models) continue;
        double v = vv[k];                 // (the curves of a set are contiguous: node g of the set)
        if (!(v - v == 0.0)) {            // non-finite node: interpolate between finite neighbours
            // (only here is the node's place in its curve needed: the 64-bit division stays off the common path)
            const long cidx = g / NS;
            const int j = (int)(g - cidx * NS);
            const double* cur = sets.p[k] + (size_t)cidx * NS;
            int jl = j - 1, jr = j + 1;
            while (jl >= 0 && !(cur[jl] - cur[jl] == 0.0)) --jl;
            while (jr < NS && !(cur[jr] - cur[jr] == 0.0)) ++jr;
            v = (jl >= 0 && jr < NS) ? lerp_np(P.st[j], P.st[jl], P.st[jr], cur[jl], cur[jr]) : HUGE_VAL;
        }
        const double a = -2.0 / 5.0 * ln10 * v;
        terms[k] = a;
        if (a != a) any_nan = true;
        if (a > amax) amax = a;
    }
    double res;
    if (any_nan) res = HUGE_VAL - HUGE_VAL;
    else if (!(amax - amax == 0.0)) res = amax;          // every model -inf (no flux) or +inf
    else {
        double sacc = 0.0;
        // (exp(0) is exactly 1: the largest term needs no exponential -- half of them for two models.  Own exp for arguments <= 0
        //  and log for [1, 8] -- exp_neg / log_pos, ~1 ulp: the library's two calls were a third of a node's instructions)
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < n_models) sacc += (terms[k] == amax) ? 1.0 : exp_neg(terms[k] - amax);
        res = log_pos(sacc) + amax;
    }
    // (x (1 / ln 10) instead of / ln 10: one rounding more than the reference's expression -- 1 ulp of a magnitude -- for a
    //  division's ~35 instructions less per node)
    return (-5.0 / 2.0 * res) * 0.43429448190325176;
}

// em_logl's combined-model flavour (FASTM 7): stack2_fast without its early returns -- the same operations in the same order for a
// finite pair (bit-identical result), garbage for a non-finite one (the caller sorts those out) -- so that the lean task stays one
// straight-line scheduling region.
__device__ __forceinline__ double stack2_node(const double v0, const double v1, const double* tab2) {
    const double lo = v0 < v1 ? v0 : v1, D = fabs(v0 - v1);
    const double sc = D * STACK2_INV_H;
    int idx = (int)sc;
    idx = idx > STACK2_NINT - 1 ? STACK2_NINT - 1 : idx;
    idx = idx < 0 ? 0 : idx;
    const double t = 2.0 * (sc - (double)idx) - 1.0;
    const double2* cf = reinterpret_cast<const double2*>(tab2) + idx * (STACK2_ROW / 2);
    const double2 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5];
    double p = fma(c0.x, t, c0.y);
    p = fma(p, t, c1.x); p = fma(p, t, c1.y);
    p = fma(p, t, c2.x); p = fma(p, t, c2.y);
    p = fma(p, t, c3.x); p = fma(p, t, c3.y);
    p = fma(p, t, c4.x); p = fma(p, t, c4.y);
    p = fma(p, t, c5.x);
    return (D >= STACK2_DMAX) ? lo : lo - p;
}
// lc_stack_node's general form for (kn, +inf): the second model contributes no flux -- terms {a, -inf}, sum of exponentials exactly 1,
// log exactly 0 -- operation for operation (NaN for a non-finite kn).
__device__ __forceinline__ double stack2_no_flux(const double kn) {
    const double ln10 = 2.302585092994046;
    const double a = -2.0 / 5.0 * ln10 * kn;
    return (-5.0 / 2.0 * a) * 0.43429448190325176;
}

}  // namespace nmma
