// em_kernels.hip -- gfx950 kernels of the batched EM light-curve log-likelihood.
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

__device__ __forceinline__ void sync_wait(int* cnt, const int target, int* watchdog_generic = nullptr, const int code = 0) {
    g_ip watchdog = (g_ip)(uintptr_t)watchdog_generic;
    // Every VALU instruction of a polling wave takes an issue slot from the MFMA waves of its SIMD (a poll is
    // v_mov + ds_read + v_cmp): sleep ~400 cycles between polls so that waiting costs next to nothing.
    int spins = 0;
    while (__hip_atomic_load((lds_ip)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(NMMA_SYNC_SLEEP);
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
static_assert((16 * STAGE_COLS + 2 * STAGE_COSMO) * 8 >= (16 + 5 * 2 * 64 + 2 * 16 + 3 * 16) * 8 + 8 * 40, "parked walk state");
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

template <int R, int KP, int PF, int NMW, int NVW, bool FAST, class LateX>
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

    int base = item_base(0);
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
    load_x(0, xB);
    f32x4 d[R];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        d[rb] = rbias[0];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
            d[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[0][kp], xB[rb][kp], d[rb], 0, 0, 0);
    }

#pragma unroll 1
    for (int k = 0; k < W; ++k) {
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
};

// One candidate layout: `nbuf` ring slots, photometry staged or not.
// Dynamic LDS a launch may ask for: the 160 KiB of a CU minus the kernel's static words (g_wd_trip), rounded down to the
// 1-KiB granule the layouts use -- a layout of exactly 160 KiB is refused by hipFuncSetAttribute.
constexpr int LDS_DYNAMIC_MAX = 159 * 1024;

__host__ inline LdsW lds_layout_logl_try(int R, int NS, int nf_avg_max, int tab_bytes, int tab_fast_bytes, int n_items, int M,
                                         int NP, int all_fast, int n_data, int n_sys_slots, int nbuf, bool stage_dat, int ext_rows = 0,
                                         int dat_point_bytes = 32, int dense_rows = 0) {
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
    L.sync = off; off = align16(off + (3 * n_items + 4 + 4 * n_items + 1) * 4);      // (+ produced / consumed counters per (item, 16 samples): dense task)
    L.stage = off; off = align16(off + (TS * STAGE_COLS + 2 * STAGE_COSMO) * 8);
    L.tmap = off;  off = align16(off + (all_fast ? TMAP_MAX * 4 : 0));   // fast mode: task index -> (item << 8 | chunk)
    L.dat = (all_fast && stage_dat && n_data <= DAT_MAX) ? off : -1;     // fast mode: photometry [t | m | 1/sigma | log sigma]
    if (L.dat >= 0) off = align16(off + n_data * dat_point_bytes);     // (8: the epochs only -- item-staged photometry, EmDev::dat_in_tab)
    L.epar = off;  off = align16(off + (all_fast ? n_sys_slots * TS * 8 : 0));        // fast modes: sysv[slot][sample]
    L.exttab = off; off = align16(off + ext_rows * TS * 8);          // lean task with extinction: ext_mag[item][sample]
    L.xn = off;   off = align16(off + M * NP * 2 * 8);               // (pmin, 1/pspan) per model filter and parameter
    L.bad = off;  off = align16(off + 5 * TS * 4);                   // bad[TS] (NaN terms) | badp[4][TS] (prologue parts)
    L.cdl = off;  off = align16(off + (dense_rows ? 0 : 16 * 2 * 4 * 16 * 8));      // per wave (any role): 2 x 4 slots x 16 coefficients (the dense task has none)
    L.nodes = off; off = align16(off + DENSE_NBUF * dense_rows * DENSE_STRIDE * 8);
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
                                     int dense_rows = 0) {
    constexpr int LDS_MAX = LDS_DYNAMIC_MAX;
    int want = n_items < 1 ? 1 : (n_items < (all_fast ? 4 : 3) ? n_items : (all_fast ? 4 : 3));
    if (want > ring_max) want = ring_max < 1 ? 1 : ring_max;
    LdsW L{};
    for (int pass = 0; pass < 2; ++pass)
        for (int nbuf = want; nbuf >= (pass == 0 ? (want < 3 ? want : 3) : 1); --nbuf) {
            // (all_fast == 1, the lean task, reads the photometry from LDS only: never give the staging up)
            L = lds_layout_logl_try(R, NS, nf_avg_max, tab_bytes, tab_fast_bytes, n_items, M, NP, all_fast, n_data, n_sys_slots, nbuf, pass == 0 || all_fast == 1, ext_rows, dat_point_bytes, dense_rows);
            if (L.total <= LDS_MAX) return L;
        }
    return L;     // does not fit: the launch fails with an explicit error
}

// FASTM = 0: generic item phase; 1: every work item qualifies for the basic fast task (constant systematics,
// at most 2 G points per filter); 2: extended fast task (sampled systematics per datum, any number of points) --
// separate instantiations so that the extensions cost the basic configuration nothing.
// Fast modes: every work item qualifies for the fast path (EmDev::all_fast) -- the generic item phase and its
// LDS table staging are not compiled in, which keeps the register budget small enough for 16-wave workgroups.
// WALKF: the MCMC step fused in (nmma_em_loglike_walk) -- the first likelihood wave, which sums the tile's log L, also runs the accept
// of the walk's step `wstep` and the proposal of the next one for the tile's chains (walk_device.h: the device functions of
// walk_step_kernel, same arithmetic) and writes the tile's theta rows for the next launch.  Its own instantiations: the walk code
// must not touch the register allocation of the tuned flavours.
template <int R, int KP, int NMW, int NVW, int FASTM, int WALKF = 0>
__global__ __launch_bounds__(logl_threads(NMW, NVW), (NMW + NVW) / 4) void em_logl(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const LdsW L,
    const int always_floor, double* __restrict__ out, double* __restrict__ chi_parts, double* __restrict__ gp_parts,
    long long* __restrict__ dbg, const nmma_walk_fuse* __restrict__ wf = nullptr, const unsigned long long wstep = 0, const int wlast = 0) {
    constexpr int TS = 16 * R;
    constexpr int PF = (R == 1) ? 8 : 4;
    constexpr bool FAST = FASTM != 0, EXT = FASTM == 2;
    // FASTM == 3: the lean task with its extras compiled in (filters with more than 32 points, a sampled em_syserr); the
    // plain lean kernel (FASTM == 1, BASELINE config 2 and the CLI grid) does not carry them: they cost it 2 % when present
    constexpr bool SPLITTABLE = R == 1 && FASTM != 0 && FASTM != 2;    // small batches: one band per workgroup (launch_logl_one)
    constexpr bool DENSE = FASTM == 6;       // lean task that reconstructs ALL nodes of (item, 16 samples) on the fp64 matrix cores (many points per filter)
    constexpr bool LEANX = FASTM >= 3;       // 3: equally spaced sample_times, 4: unequally spaced (fewer inlined variants per kernel)
    constexpr int NV = 64 * NVW;      // VALU-role threads

    // (blockIdx.y > 0 only in the split launch of small batches: one copy of the configuration per observed band)
    const EmDev& P = Pp[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* praw = reinterpret_cast<double*>(smem + L.praw);
    double* scal = reinterpret_cast<double*>(smem + L.scal);
    double* stl = reinterpret_cast<double*>(smem + L.stl);
    float* part = reinterpret_cast<float*>(smem + L.part);
    double* chi_tot = reinterpret_cast<double*>(smem + L.chi);
    double* gp_tot = reinterpret_cast<double*>(smem + L.gp);
    int* bad = reinterpret_cast<int*>(smem + L.bad);
    double* estb = reinterpret_cast<double*>(smem + L.est);
    unsigned char* tabl = smem + L.tab;
    double* cdl = reinterpret_cast<double*>(smem + L.cdl);
    const ItemDesc* itab = reinterpret_cast<const ItemDesc*>(smem + L.itab);
    int* sync = reinterpret_cast<int*>(smem + L.sync);
    double* xnl = reinterpret_cast<double*>(smem + L.xn);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NBUF = L.nbuf;
    for (int j = tid; j < 7 * P.n_items + 5; j += logl_threads(NMW, NVW)) sync[j] = 0;      // (the last one: fused MCMC step, first phase done)
    if (tid == 0) g_wd_trip = 0;
    __syncthreads();             // the only workgroup barrier: counters zeroed
    const long tile0 = (long)blockIdx.x * TS;
    const int NP = P.NP, NC = P.NC, NT = P.NT, NS = P.NS;
    const int W = P.n_items;
    gci32p items = as_global(P.items);
    // (WALKF = 8: the fused MCMC step keeps a group of that many lanes per chain -- one per sampled dimension -- and 16-sample tiles)
    constexpr int WT = WALKF ? WALKF : 8;             // lanes per chain (8: up to 8 sampled dimensions; the parked state below is sized for it)
    constexpr int WCR = 64 / WT;                      // chains per round of one wave
    constexpr int WNR = WALKF ? TS / WCR : 1;
    // (staging area after the prologue: tot[TS] | parked walk state 5 x WNR x 64 + 2 TS doubles + 6 TS ints | prior table)
    nmma_walk_prior* wspl = reinterpret_cast<nmma_walk_prior*>(reinterpret_cast<double*>(smem + L.stage) + TS + 5 * WNR * 64 + 2 * TS + 3 * TS);

    if (wave < NMW) {
        // ============================ MFMA role ============================
        // layer-1 operands straight from theta (no dependency on the other role's prologue)
        double xraw[R][KP];
        auto fill_xraw = [&]() {
#pragma unroll
            for (int rb = 0; rb < R; ++rb) {
                long b = tile0 + rb * 16 + (lane & 15);
                if (b >= B) b = B - 1;
                const double* row = theta + b * ld;
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) {
                    const int p = 4 * kp + (lane >> 4);
                    xraw[rb][kp] = (p < NP) ? apply_slot(P.model_param[p], row) : 0.0;
                }
            }
        };
#ifndef NMMA_DBG_PRELOAD_FIRST
        fill_xraw();
#endif
        switch (P.prio_mfma) {
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 3: __builtin_amdgcn_s_setprio(3); break;
            default: break;
        }
#ifdef NMMA_DBG_PRELOAD_FIRST
        mfma_role<R, KP, PF, NMW, NVW, FAST>(P, xraw, xnl, wave, lane, part, L.nbuf, sync, dbg, fill_xraw);
#else
        mfma_role<R, KP, PF, NMW, NVW, FAST>(P, xraw, xnl, wave, lane, part, L.nbuf, sync, dbg, [] {});
#endif
        if (!FAST || !P.helpers) return;
        // fast mode: the record stream is done -- join the likelihood workers for the remaining tasks
    }

    // ================================ VALU role ================================
    // Few instructions, long dependency chains: give them the issue slot whenever they are
    // ready; the MFMA waves (lower priority) soak up every other cycle of the SIMD.
    switch (P.prio_valu) {
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    const bool helper = wave < NMW;              // an MFMA-role wave that finished its stream
    const int vt = tid - 64 * NMW;               // negative for helpers (they skip the prologue)
    const int vwave = wave - NMW;
    // Lean task with extinction: ext_mag[item][sample] of this tile in LDS, filled by the prologue lane that owns the sample's
    // E(B-V) -- coefficient x E(B-V) for the linear law, the pre-pass kernel's value for the Pei-1992 law -- so that a task reads
    // ONE LDS word per slot (anything more inside the task tips hipcc into spilling: see the register-budget test)
    auto fill_ext = [&](const int s_l, const double ebv) {
        double* et = reinterpret_cast<double*>(smem + L.exttab);
        if (!P.has_ebv) { et[s_l] = 0.0; return; }        // one row of zeros that every item reads
        long bb = tile0 + s_l;
        if (bb >= B) bb = B - 1;
        for (int kk = 0; kk < W; ++kk) {
            const int m_k = P.item_desc[kk].m;
            et[kk * TS + s_l] = P.p92_tab ? P.ext_tab[bb * P.M + m_k] : ((ebv != 0.0) ? P.item_desc[kk].ebvc * ebv : 0.0);
        }
    };
    if (!helper) {
        if (dbg && blockIdx.x == 0 && vt == 0) dbg[64] = clock64();
        // ---- prologue: per-sample scalars, accumulators, sample-time grid.
        // The tile's theta rows and the cosmology grid are first staged in LDS by all likelihood waves
        // (ONE memory round trip), so the serial slot/interpolation chain of sample_scalars runs on
        // LDS latency instead of ~40 dependent L2/HBM round trips.
        const bool staged = (ld <= STAGE_COLS) && (P.redshift_mode != NMMA_Z_GRID || P.n_cosmo <= STAGE_COSMO);
        double* thl = reinterpret_cast<double*>(smem + L.stage);
        double* dgl = thl + TS * STAGE_COLS;
        double* zgl = dgl + STAGE_COSMO;
        if (staged) {
            const int ncol = (int)ld;
            for (int idx = vt; idx < TS * ncol; idx += NV) {
                const int sidx = idx / ncol, cidx = idx - sidx * ncol;
                long b = tile0 + sidx;
                if (b >= B) b = B - 1;
                thl[sidx * ncol + cidx] = theta[b * ld + cidx];
            }
            if (P.redshift_mode == NMMA_Z_GRID)
                for (int j = vt; j < P.n_cosmo; j += NV) { dgl[j] = P.dist_grid[j]; zgl[j] = P.z_grid[j]; }
            sync_signal(sync + 2 * W + 2, lane);
        }
        int* badp = bad + TS;        // [4][TS]: non-finite input seen by prologue part 0..3 for sample s
        if (staged) {
            // four waves share the per-sample chain (a divergent split inside one wave would execute all parts
            // one after the other): 0 = redshift branch (grid interpolation, log10), 1 = distance modulus, time
            // shift, E(B-V), 2 = model parameters, 3 = systematics parameters.  Lane = sample.
            if (vwave < 4 && lane < TS) {
                sync_wait(sync + 2 * W + 2, NVW, P.watchdog, 200);
                // (LDS-address-space pointers: ds_read instead of flat_load for the staged rows and grids)
                typedef const __attribute__((address_space(3))) double* lds_cdp;
                const lds_cdp row = (lds_cdp)(thl + lane * (int)ld);
                const lds_cdp dgl_l = (lds_cdp)dgl, zgl_l = (lds_cdp)zgl;
                double* sc = scal + lane * 8;
                double chk = 0.0;
#ifdef NMMA_DBG_NOCHAINS
                // measurement build: what a pre-pass kernel for the per-sample scalars could save AT MOST -- the chains are gone,
                // plausible constants stand in (the tasks run the same instruction stream on wrong numbers)
                if (vwave == 0) { sc[S_ZP1] = 1.0093; sc[S_IZP1] = 1.0 / 1.0093; sc[S_RC] = -0.01; }
                else if (vwave == 1) { sc[S_DMOD] = 33.0; sc[S_TS] = -0.3; sc[S_EBV] = 0.0; if constexpr (LEANX) fill_ext(lane, 0.0); }
                else if (vwave == 3) bad[lane] = 0;
                (void)row; (void)dgl_l; (void)zgl_l;
#else
                if (vwave == 0) {
                    const double d_l = apply_slot(P.lumdist, row);
                    double z = 0.0;
                    if (P.redshift_mode == NMMA_Z_SLOT) z = apply_slot(P.redshift, row);
                    else if (P.redshift_mode == NMMA_Z_GRID) {
                        const double d_eff = P.has_h0 ? d_l * apply_slot(P.hubble, row) * P.inv_h0_ref : d_l;
                        z = interp_np(d_eff, dgl_l, zgl_l, P.n_cosmo, zgl_l[0], zgl_l[P.n_cosmo - 1]);
                        if (P.has_h0) z *= d_eff;
                    }
                    sc[S_ZP1] = 1 + z;
                    sc[S_IZP1] = 1.0 / (1 + z);
                    sc[S_RC] = redshift_correction(z);
                    chk = d_l + z;
                } else if (vwave == 1) {
                    const double d_l = apply_slot(P.lumdist, row);
                    sc[S_DMOD] = distance_modulus(d_l);
                    sc[S_TS] = apply_slot(P.timeshift, row);
                    sc[S_EBV] = P.has_ebv ? apply_slot(P.ebv, row) : 0.0;
                    chk = sc[S_TS] + sc[S_EBV];
                    if constexpr (LEANX) fill_ext(lane, sc[S_EBV]);
                } else if (vwave == 2) {
                    for (int p = 0; p < P.NP; ++p) chk += apply_slot(P.model_param[p], row);
                } else {
                    // (sampled time nodes are not "inputs that must be finite": autocomplete_data masks them, em/utils.py:634-645)
                    for (int o = 0; o < P.O; ++o) {
                        const int q0 = P.sys_off[o], q1 = P.sys_off[o + 1];
                        const bool nodes = P.sys_kind[o] == NMMA_SYS_NODES;
                        bool odd = false;
                        for (int q = q0; q < q1; ++q) {
                            const double v = apply_slot(P.sys_slots[q], row);
                            if (nodes) odd = odd || !(v - v == 0.0); else chk += v;
                            if constexpr (FAST) reinterpret_cast<double*>(smem + L.epar)[q * TS + lane] = v;   // sysv[slot][sample]
                        }
                        if constexpr (FAST)
                            if (nodes && odd) repair_nodes(reinterpret_cast<double*>(smem + L.epar) + q0 * TS + lane, TS, P.sys_node_t + q0, q1 - q0);
                    }
                    bad[lane] = 0;

                }
#endif
                badp[vwave * TS + lane] = (chk - chk == 0.0) ? 0 : 1;
            }
        } else if (vt < TS) {
            long b = tile0 + vt;
            if (b >= B) b = B - 1;
            const double* row = theta + b * ld;
            double chk;
            sample_scalars(P, row, praw + vt * 8, scal + vt * 8, chk);
            for (int o = 0; o < P.O; ++o) {
                const int q0 = P.sys_off[o], q1 = P.sys_off[o + 1];
                const bool nodes = P.sys_kind[o] == NMMA_SYS_NODES;
                bool odd = false;
                for (int q = q0; q < q1; ++q) {
                    const double v = apply_slot(P.sys_slots[q], row);
                    if (nodes) odd = odd || !(v - v == 0.0); else chk += v;
                    if constexpr (FAST) reinterpret_cast<double*>(smem + L.epar)[q * TS + vt] = v;
                }
                if constexpr (FAST)
                    if (nodes && odd) repair_nodes(reinterpret_cast<double*>(smem + L.epar) + q0 * TS + vt, TS, P.sys_node_t + q0, q1 - q0);
            }
            if constexpr (LEANX) fill_ext(vt, scal[vt * 8 + S_EBV]);
            badp[vt] = (chk - chk == 0.0) ? 0 : 1;
            badp[TS + vt] = 0; badp[2 * TS + vt] = 0; badp[3 * TS + vt] = 0;
            bad[vt] = 0;
        }
        // The table copies below do not depend on the per-sample chains above: with the staged prologue the chains occupy
        // likelihood waves 0-3 only, so waves 4-7 take ALL the copies and the two run side by side (a band's workgroup of the
        // split launch has nothing to hide its prologue behind, so the prologue costs the longer of the two instead of their sum)
        const bool two_lane_prologue = staged && NVW == 8;
        const int cvt = two_lane_prologue ? vt - 256 : vt;
        const int cnv = two_lane_prologue ? NV - 256 : NV;
        if (cvt >= 0) {
        // (slots: one per work item; some task flavours file a band's sums under its observed-filter index instead, which in a
        //  band's own workgroup of the split launch can lie beyond its item count -- the layout always holds P.O slots)
        for (int j = cvt; j < (W > P.O ? W : P.O) * TS; j += cnv) { chi_tot[j] = 0.0; gp_tot[j] = 0.0; }
#ifdef NMMA_DBG_NODIV     // measurement build: the table of reciprocal spacings without its division (a create-time table would hold them)
        for (int j = cvt; j < NS; j += cnv) { stl[j] = P.st[j]; stl[NS + j] = (j + 1 < NS) ? P.st_inv_dt : 0.0; }
#else
        for (int j = cvt; j < NS; j += cnv) { stl[j] = P.st[j]; stl[NS + j] = (j + 1 < NS) ? 1.0 / (P.st[j + 1] - P.st[j]) : 0.0; }
#endif
        if (!P.st_uniform && P.bguess != nullptr) {
            int* bgl = reinterpret_cast<int*>(stl + 2 * NS);
            gci32p bgs = as_global(P.bguess);
            for (int j = cvt; j <= BG_CELLS; j += cnv) bgl[j] = bgs[j];
        }
        if constexpr (FAST) {
            int* tmap = reinterpret_cast<int*>(smem + L.tmap);
            gci32p src = as_global(P.task_map[R - 1]);
            for (int j = cvt; j < P.n_tasks[R - 1] && j < TMAP_MAX; j += cnv) tmap[j] = src[j];
            if (L.dat >= 0) {
                double* dat = reinterpret_cast<double*>(smem + L.dat);
                const int nd = P.n_data;
                if constexpr (!EXT) {      // lean task: one {t, m, 1/sigma, ln sigma} record per datum
                    if (LEANX && P.dat_in_tab) {      // item-staged photometry: the epochs only, the records come with the item's rows
                        gcf64p sdt = as_global(P.dt);
                        for (int j = cvt; j < nd; j += cnv) dat[j] = sdt[j];
                    } else {
                        gcf64p src4 = as_global(P.dat4);
                        for (int j = cvt; j < 4 * nd; j += cnv) dat[j] = src4[j];
                    }
                } else {
                    gcf64p sdt = as_global(P.dt), sdm = as_global(P.dm), sis = as_global(P.dinvsig), sls = as_global(P.dlogsig);
                    for (int j = cvt; j < nd; j += cnv) { dat[j] = sdt[j]; dat[nd + j] = sdm[j]; dat[2 * nd + j] = sis[j]; dat[3 * nd + j] = sls[j]; }
                }
            }
        }
        {
            gci32p src = as_global(reinterpret_cast<const int*>(P.item_desc));
            int* dst = reinterpret_cast<int*>(smem + L.itab);
            for (int j = cvt; j < W * ITEM_WORDS; j += cnv) dst[j] = src[j];
        }
        }

        // Static tables of a model filter (basis rows, span, mins, stage-1 lerp tables) are
        // copied global -> LDS by LDS-DMA one item ahead, into the other half of a double buffer:
        // no registers, and the L2 latency hides behind the phase of the previous item.
    }
    const int tab_bytes = P.tab_bytes;                 // multiple of 1 KiB (one wave-instruction)
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
    auto tab_dma = [&](int k) {
        const int m = items[4 * k + 2];
        gbyte_p src = (gbyte_p)(uintptr_t)(P.tab + (size_t)m * tab_bytes);
        lds_p dst = (lds_p)(tabl + (k & 1) * tab_bytes);
        for (int c = vwave; c * 1024 < tab_bytes; c += NVW)
            __builtin_amdgcn_global_load_lds(src + c * 1024 + lane * 16, dst + c * 1024, 16, 0, 0);
    };
    constexpr bool all_fast = FAST;
    if (!helper) {
        if constexpr (!FAST) tab_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dbg && blockIdx.x == 0 && vt == 0) dbg[65] = clock64();
        sync_signal(sync + W + 1, lane);     // phase "prologue" of this wave done
    }

    auto sample_bad = [&](const int s_) -> bool {      // a non-finite input of sample s_ (any prologue part)
        const int* badp = bad + TS;
        return (badp[s_] | badp[TS + s_] | badp[2 * TS + s_] | badp[3 * TS + s_]) != 0;
    };
    const bool uniform = P.st_uniform != 0;
    const double st0 = P.st0, inv_dt = P.st_inv_dt;
    gcf64p g_dt = as_global(P.dt), g_dm = as_global(P.dm), g_dsig = as_global(P.dsig);
    gcf64p g_sigtot = as_global(P.dsigtot), g_logsig = as_global(P.dlogsig), g_invsig = as_global(P.dinvsig);

    auto item_phase = [&](auto nct_tag, const int k) {
        constexpr int NCT = decltype(nct_tag)::value;
        const ItemDesc& it = itab[k];
        const int o = it.o, ks = it.ks, nsrc = it.nsrc;
        const float* pbuf = part + (k % NBUF) * (NSLICE * TS * PSTR);
        const int jlo = it.jlo, jhi = it.jhi;
        const bool identity = it.identity != 0;
        const double ebvc = it.ebvc;
        const int G = it.G;                       // lanes per sample, power of two in [16, 64]
        const int gpb = NV / G;
        const int g = vt / G, gi = vt - g * G;
        const int d0 = it.d0;
        const int nf = it.nf;
        const int kind = it.kind;
        const double lim = it.lim;
        const double e_const = it.e_const;
        const int npass = (TS + gpb - 1) / gpb;
        const unsigned char* tb = tabl + (k & 1) * tab_bytes;
        const double* rows_m = reinterpret_cast<const double*>(tb);   // [NT][RS]: VA row | span | mins
        const int RS = P.tab_row_stride;
        const double* s1dx = reinterpret_cast<const double*>(tb + P.tab_off_s1dx);
        const double* s1of = reinterpret_cast<const double*>(tb + P.tab_off_s1of);
        const int* s1i = reinterpret_cast<const int*>(tb + P.tab_off_s1i);
        const float* b2l = reinterpret_cast<const float*>(tb + P.tab_off_b2);
        const bool dbg_on = dbg && blockIdx.x == 0 && vt == 0 && k == W - 1;
        if (dbg_on) dbg[96] = clock64();
        // the lane's first datum is the same for every sample: keep it in registers
        double c_t = 0, c_m = 0, c_sd = 0, c_sig = 0, c_lsig = 0;
        if (gi < nf) {
            const int di = d0 + gi;
            c_t = g_dt[di]; c_m = g_dm[di]; c_sd = g_dsig[di];
            if (kind == NMMA_SYS_CONST) { c_sig = g_sigtot[di]; c_lsig = g_logsig[di]; }
        }
        // (ordinary loads retired before the DMA is issued: a later wait on them would
        //  otherwise drain the DMA queue as well)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(c_t), "+v"(c_m), "+v"(c_sd), "+v"(c_sig), "+v"(c_lsig)::"memory");
        if (k + 1 < W) tab_dma(k + 1);            // next item's tables: in flight during this phase
        if (dbg_on) dbg[97] = clock64();

        for (int pass = 0; pass < npass; ++pass) {
            const int sl = pass * gpb + g;
            const bool active = sl < TS;
            const int s = active ? sl : 0;
            // waves whose groups are all beyond the tile have nothing to do in this pass (uniform)
            if (pass * gpb + (vwave * 64) / G >= TS) continue;
            // slice reduction (fixed order) + bias of the second Dense: lane gi < 16 owns
            // coefficient gi and hands it to its group through LDS (same wave: LDS is in order)
            if (gi < 16) {
                const int rb = s >> 4, sidx = s & 15;
                float cmine = 0.f;
#pragma unroll
                for (int w = 0; w < NSLICE; ++w) cmine += pbuf[((w * R + rb) * 16 + sidx) * PSTR + gi];
                cmine += b2l[gi];
                cdl[(vwave * (64 / 16) + (lane >> 4)) * 16 + gi] = (double)cmine;   // row: 16-lane slot of this wave
                // a model filter nobody observed (nf == 0): its light curve still has to be a usable one -- a non-finite
                // coefficient makes every node non-finite, the reference's sanity_check then floors the sample
                // (em_likelihood.py:305-311; with data the NaN reaches the sum through the terms themselves)
                if (nf == 0 && active && gi < NC && !(cmine - cmine == 0.f)) bad[s] = 1;
            }
            const double* crow = cdl + (vwave * 4 + ((lane & ~(G - 1)) >> 4)) * 16;     // the group's first slot
            // NCT > 0: exactly NCT coefficients, kept in registers; NCT == 0: any NC, re-read from LDS
            constexpr int NREG = NCT > 0 ? NCT : 1;
            double cc[NREG];
            if constexpr (NCT > 0) {
#pragma unroll
                for (int j = 0; j < NCT; ++j) cc[j] = crow[j];
            }

            if (dbg_on && pass == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dbg[98] = clock64(); }
            const double zp1 = scal[s * 8 + S_ZP1], tsh = scal[s * 8 + S_TS];
            const double dmod = scal[s * 8 + S_DMOD], rc = scal[s * 8 + S_RC];
            const double ebv = scal[s * 8 + S_EBV], izp1 = scal[s * 8 + S_IZP1];
            const double ext = extinction_mag(P.ext_law, ebvc, zp1, ebv);
            const double t_lo = stl[jlo] * zp1 + tsh, t_hi = stl[jhi] * zp1 + tsh;
            long brow = tile0 + s;
            if (brow >= B) brow = B - 1;
            const double* row = theta + brow * ld;

            // absolute magnitude at SVD-grid node i: (VA[i,:] . c) * span[i] + mins[i]
            auto mag_abs = [&](int i) -> double {
                const double* vr = rows_m + i * RS;
                double a;
                if constexpr (NCT > 0) {
                    a = vr[0] * cc[0];
#pragma unroll
                    for (int j = 1; j < NCT; ++j) a = fma(vr[j], cc[j], a);
                } else {
                    a = vr[0] * crow[0];
                    for (int j = 1; j < NC; ++j) a = fma(vr[j], crow[j], a);
                }
                return a * vr[NC] + vr[NC + 1];
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
                    // stage-2: np.interp(t, t_obs[jlo..jhi], app, left=right=+inf); straight-line:
                    // bracket (clamped), both nodes reconstructed, result selected at the end
                    const bool inside = (jhi > jlo) && t >= t_lo && t <= t_hi;
                    int lo;   // t_obs[lo] <= t < t_obs[lo+1]  (lo = jhi-1 when t == t_obs[jhi])
                    if (uniform) {
                        lo = (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                        lo = lo < jlo ? jlo : (lo > jhi - 1 ? jhi - 1 : lo);
                    } else {
                        lo = jlo;
                        int hi = jhi;
                        while (inside && hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                        }
                    }
                    if (lo > jhi - 1) lo = jhi - 1;
                    if (lo < 0) lo = 0;
                    int hi1 = lo + 1 < NS ? lo + 1 : lo;
                    double x0 = stl[lo] * zp1 + tsh, x1 = stl[hi1] * zp1 + tsh;
                    // exact re-check of the guessed bracket (normally no iteration)
                    while (inside && ((x0 > t && lo > jlo) || (x1 <= t && lo < jhi - 1))) {
                        lo += (x0 > t) ? -1 : 1;
                        hi1 = lo + 1;
                        x0 = stl[lo] * zp1 + tsh; x1 = stl[hi1] * zp1 + tsh;
                    }
                    const double y0 = app_mag(lo), y1 = app_mag(hi1);
                    double est = lerp_np(t, x0, x1, y0, y1);
                    if (x0 == t) est = y0;
                    if (x1 == t) est = y1;          // also the exact right edge (np.interp: fp[-1])
                    if (!inside) est = (t != t) ? t : dinf();
                    if (dbg_on && pass == 0) { asm volatile("" : "+v"(est)); dbg[99] = clock64(); }
                    if (nsrc > 1) {  // averaged band: (a + b [+ c]) / n  (utils.py:566-584)
                        double acc_e = est;
                        if (ks > 0) acc_e = estb[s * L.nf_max + dd] + est;
                        if (ks < nsrc - 1) { estb[s * L.nf_max + dd] = acc_e; continue; }
                        est = acc_e / (double)nsrc;
                    }
                    // systematics (systematics.py:279-296) and combined sigma (em_likelihood.py:341)
                    double e = e_const;
                    if (kind != NMMA_SYS_CONST) {
                        const nmma_slot* sv = P.sys_slots + P.sys_off[o];
                        if (kind == NMMA_SYS_PARAM) {
                            e = apply_slot(sv[0], row);
                        } else {
                            const int K = P.sys_nn[o];
                            const int ni = P.sys_nidx[di];
                            if (ni < 0) e = apply_slot(sv[0], row);
                            else if (ni >= K - 1) e = apply_slot(sv[K - 1], row);
                            else {
                                const double v0 = apply_slot(sv[ni], row), v1 = apply_slot(sv[ni + 1], row);
                                const double sl2 = (v1 - v0) / P.sys_ndx[di];
                                e = sl2 * P.sys_noff[di] + v0;
                            }
                            // (a non-finite node value: autocomplete_data's finite mask, em/utils.py:634-645)
                            if (!(e - e == 0.0)) e = masked_nodes_at(sv, P.sys_node_t + P.sys_off[o], K, P.dt[di], row);
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
            if (dbg_on && pass == 0) { asm volatile("" : "+v"(chi), "+v"(gp)); dbg[100] = clock64(); }
            if (ks < nsrc - 1) continue;      // uniform: more sources of this band to come
            // group reduction by DPP (no LDS round trips); the sum lands in the group's LAST row
            chi = group_sum(chi, G);
            gp = group_sum(gp, G);
            if (active && gi == G - 16) {
                // running sums over observed filters, in filter order (em_likelihood.py:337-352)
                chi_tot[k * TS + s] = chi;
                gp_tot[k * TS + s] = gp;
                if (chi != chi) bad[s] = 1;
                if (chi_parts != nullptr && tile0 + s < B) {
                    chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                    gp_parts[(long)o * B + tile0 + s] = gp;
                }
            }
            if (dbg_on && pass == 0) dbg[101] = clock64();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of the next item's tables has landed
        if (dbg_on) dbg[102] = clock64();
    };

    // ---------------------------------------------------------------------------------
    // Fast path (every item qualifies; flag computed at create): NC == 10, one source per band,
    // constant systematics, no detection limit, sample_times equal to an equally spaced SVD grid,
    // no extinction.  The likelihood waves are independent workers: a task is one wave-load of
    // (item k, 64/G consecutive samples), G = 16/32/64 lanes per sample with two data per lane;
    // tasks are dealt round-robin to the waves.  A task has two stages:
    //   P (needs only the prologue): bracket every datum on the sample's observer-frame grid and
    //     fetch its two basis rows [VA | span | mins] from L2 into registers;
    //   Q (needs the coefficients of item k): reduce the 8 slice partial sums, 2 x 10 FMAs per datum,
    //     lerp, likelihood term, DPP group sum.
    // Only Q is on the critical path behind the MFMA role, and it contains no global-memory latency.
    // ---------------------------------------------------------------------------------
    const int nd_l = P.n_data;
    const double* dat_l = (FAST && L.dat >= 0) ? reinterpret_cast<const double*>(smem + L.dat) : nullptr;
    auto fast_task = [&](auto kind_tag, const int k, const int c) {
        // systematics of the item: 0 = constant (1/sigma_tot precomputed), 1 = one sampled parameter, 2 = sampled time nodes
        constexpr int SK = decltype(kind_tag)::value;
        constexpr bool par = SK != 0;
        const bool dbt = dbg && blockIdx.x == 0 && lane == 0 && c == 0 && k == W - 1;
        if (dbt) dbg[96] = clock64();
        const ItemDesc& it = itab[k];
        const int o = it.o;
        if (c == 0) {
            // The wave that owns the item's first task stages its basis rows [VA | span | mins | b2] into ring slot
            // k % NBUF by LDS-DMA (no registers).  Tasks are claimed well before their item is published, so the
            // copy lands while this wave does stage P and waits for the MLP.  (Sharing the copy among the item's
            // tasks would make them wait for each other: with more tasks per item than free waves that deadlocks.)
            if (k >= NBUF) sync_wait(sync + W + 1 + (k - NBUF + 1), itab[k - NBUF].ntask[R - 1], P.watchdog, 800 + k);
            typedef __attribute__((address_space(3))) unsigned char* lds_bp;
            typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
            gbyte_p src = (gbyte_p)(uintptr_t)(P.tab + (size_t)it.m * P.tab_bytes);
            lds_bp dst = (lds_bp)(tabl + (k % NBUF) * P.tab_fast_bytes);
            for (int q = 0; q * 1024 < P.tab_fast_bytes; ++q)
                __builtin_amdgcn_global_load_lds(src + q * 1024 + lane * 16, dst + q * 1024, 16, 0, 0);
        }
        const float* pbuf = part + (k % NBUF) * (NSLICE * TS * PSTR);
        const int jlo = it.jlo, jhi = it.jhi;
        const int G = it.G, d0 = it.d0, nf = it.nf;         // G = 16, 32 or 64 (a power of two: shifts, no division)
        const int lgG = (G == 16) ? 4 : (G == 32 ? 5 : 6);
        const int g = lane >> lgG, gi = lane & (G - 1);
        const int s = (c << (6 - lgG)) + g;                 // < TS: TS * G is a multiple of 64
        const double st0 = P.st0, inv_dt = P.st_inv_dt;
        const double* sc = scal + s * 8;
        const double zp1 = sc[S_ZP1], tsh = sc[S_TS], izp1 = sc[S_IZP1];
        const double dmrc = sc[S_DMOD] + sc[S_RC];
        const double izdt = izp1 * inv_dt;
        const double t_lo = stl[jlo] * zp1 + tsh, t_hi = stl[jhi] * zp1 + tsh;
        const double* sysv = reinterpret_cast<const double*>(smem + L.epar) + s;        // sysv[slot * TS]: this sample's values
        const int sv0 = par ? P.sys_off[o] : 0;                                            // first slot of the filter's group
        double e_sys = (SK == 1) ? sysv[sv0 * TS] : it.e_const;
        const bool lim_finite = EXT && (it.lim - it.lim == 0.0);
        const bool two = EXT && !(it.identity != 0 && it.same_grid != 0);
        const double ext = (EXT && sc[S_EBV] != 0.0) ? it.ebvc * sc[S_EBV] : 0.0;

        // ---- stage P: this lane's data and their brackets on the sample's observer-frame grid.
        // A lane owns data gi, gi + G, gi + 2G, ...; they are processed in pairs (two slots in registers):
        // the first pair before the item is published, further pairs (more than 2 G points per filter) after it.
        constexpr int NDL = 2;
        double c_t[NDL], c_m[NDL], c_is[NDL], c_ls[NDL], x0[NDL], x1[NDL];
        bool inside[NDL], hit1[NDL];
        int lo_[NDL];
        auto stage_p = [&](const int u0) {
#pragma unroll
            for (int u = 0; u < NDL; ++u) {
                const int dd = gi + (u0 + u) * G;
                const int di = d0 + (dd < nf ? dd : 0);
                if (dat_l != nullptr) { c_t[u] = dat_l[di]; c_m[u] = dat_l[nd_l + di]; c_is[u] = dat_l[2 * nd_l + di]; c_ls[u] = dat_l[3 * nd_l + di]; }
                else { c_t[u] = g_dt[di]; c_m[u] = g_dm[di]; c_is[u] = g_invsig[di]; c_ls[u] = g_logsig[di]; }
            }
#pragma unroll
            for (int u = 0; u < NDL; ++u) {
                if ((u0 + u) * G >= nf) { inside[u] = false; hit1[u] = false; x0[u] = 0; x1[u] = 0; lo_[u] = 0; continue; }   // uniform: slot unused by this item
                const double t = c_t[u];
                inside[u] = (jhi > jlo) && t >= t_lo && t <= t_hi;
                int lo;
                if (!EXT || uniform) {
                    lo = (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                    lo = lo < jlo ? jlo : (lo > jhi - 1 ? jhi - 1 : lo);
                } else {                     // sample_times not equally spaced (e.g. the CLI's log-spaced grid): bisection
                    lo = jlo;
                    int hi = jhi;
                    while (inside[u] && hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                    }
                    if (lo > jhi - 1) lo = jhi - 1;
                }
                if (lo < 0) lo = 0;
                double a = stl[lo] * zp1 + tsh, b = stl[lo + 1] * zp1 + tsh;
                for (int it2 = 0; it2 < 4 && inside[u] && ((a > t && lo > jlo) || (b <= t && lo < jhi - 1)); ++it2) {   // exact re-check (the guess is off by at most one)
                    lo += (a > t) ? -1 : 1;
                    a = stl[lo] * zp1 + tsh; b = stl[lo + 1] * zp1 + tsh;
                }
                x0[u] = a; x1[u] = b; hit1[u] = (b == t); lo_[u] = lo;
            }
        };
        stage_p(0);

        if (dbt) { asm volatile("" : "+v"(x0[0]), "+v"(lo_[0])); dbg[97] = clock64(); }
        // ---- stage Q
        if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sync_signal(sync + 2 * W + 4 + k, lane);     // rows of item k staged
        }
        sync_wait(sync + 2 * W + 4 + k, 1, P.watchdog, 350 + k);
        sync_wait(sync + k, NMW, P.watchdog, 300 + k);   // coefficients of item k published
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k] = clock64();
        if (dbt) dbg[98] = clock64();
        const unsigned char* tbl = tabl + (k % NBUF) * P.tab_fast_bytes;
        const double* rows_l = reinterpret_cast<const double*>(tbl);          // [NT][12]: VA row | span | mins
        const float* b2l = reinterpret_cast<const float*>(tbl + P.tab_off_b2);
        const double* s1dx_l = reinterpret_cast<const double*>(tbl + P.tab_off_s1dx);   // (staged only when some item needs them)
        const double* s1of_l = reinterpret_cast<const double*>(tbl + P.tab_off_s1of);
        const int* s1i_l = reinterpret_cast<const int*>(tbl + P.tab_off_s1i);
        if (gi < 16) {
            const int rb = s >> 4, sidx = s & 15;
            const float* pp = pbuf + (rb * 16 + sidx) * PSTR + gi;
            float cmine = 0.f;
#pragma unroll
            for (int w = 0; w < NSLICE; ++w) cmine += pp[w * (R * 16 * PSTR)];
            cmine += b2l[gi];
            cdl[(wave * 4 + (lane >> 4)) * 16 + gi] = (double)cmine;
        }
        const double* crow = cdl + (wave * 4 + ((lane & ~(G - 1)) >> 4)) * 16;
        double cc[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) cc[j] = crow[j];
        if (dbt) { asm volatile("" : "+v"(cc[0]), "+v"(cc[9])); dbg[99] = clock64(); }
        double chi = 0.0, gp = 0.0;
        auto stage_q = [&](const int u0) {
#pragma unroll
            for (int u = 0; u < NDL; ++u) {
                if ((u0 + u) * G >= nf) continue;
                double y0, y1;
                if (EXT && two) {
                    // sample_times differ from the SVD grid (uniform per item): each of the two sample nodes is itself a
                    // lerp between two SVD nodes (stage 1, lightcurve_generation.py:177) -- up to four basis rows
                    auto mag_row = [&](const int i) -> double {
                        const double* r = rows_l + i * 12;
                        double a = r[0] * cc[0];
#pragma unroll
                        for (int j = 1; j < 10; ++j) a = fma(r[j], cc[j], a);
                        return a * r[10] + r[11];
                    };
                    auto stage1 = [&](const int j) -> double {
                        const int i1 = s1i_l[j];
                        const double ya = mag_row(i1);
                        if (it.identity) return ya;
                        const double yb = mag_row(i1 + 1 < NT ? i1 + 1 : NT - 1);
                        return ((yb - ya) / s1dx_l[j]) * s1of_l[j] + ya;
                    };
                    y0 = stage1(lo_[u]);
                    y1 = stage1(lo_[u] + 1);
                } else {
                    const double* r0 = rows_l + lo_[u] * 12;
                    const double* r1 = r0 + 12;
                    double a0 = r0[0] * cc[0], a1 = r1[0] * cc[0];
#pragma unroll
                    for (int j = 1; j < 10; ++j) { a0 = fma(r0[j], cc[j], a0); a1 = fma(r1[j], cc[j], a1); }
                    y0 = a0 * r0[10] + r0[11]; y1 = a1 * r1[10] + r1[11];
                }
                if constexpr (EXT) { if (ext != 0.0) { y0 = y0 + ext; y1 = y1 + ext; } }     // uniform per sample group
                y0 = y0 + dmrc; y1 = y1 + dmrc;
                const double t = c_t[u];
                double est = ((y1 - y0) * izdt) * (t - x0[u]) + y0;
                if (EXT && !uniform) est = ((y1 - y0) / (x1[u] - x0[u])) * (t - x0[u]) + y0;
                if (hit1[u]) est = y1;
                if (!inside[u]) est = (t != t) ? t : dinf();
                if (gi + (u0 + u) * G < nf) {
                    double isig = c_is[u], lsig = c_ls[u];
                    bool sig_bad = false;
                    if constexpr (par) {             // sampled systematic, combined per datum
                        if constexpr (SK == 2) {     // time nodes: constant outside, linear in between (systematics.py:288-291)
                            const int di = d0 + gi + (u0 + u) * G;
                            const int K = P.sys_nn[o], ni = P.sys_nidx[di];
                            if (ni < 0) e_sys = sysv[sv0 * TS];
                            else if (ni >= K - 1) e_sys = sysv[(sv0 + K - 1) * TS];
                            else {
                                const double v0 = sysv[(sv0 + ni) * TS], v1 = sysv[(sv0 + ni + 1) * TS];
                                const double sl2 = (v1 - v0) / P.sys_ndx[di];
                                e_sys = sl2 * P.sys_noff[di] + v0;
                            }
                        }
                        const double sd = c_is[u];   // (the slot carries sigma_data for these filters)
                        const double sig = sqrt(sd * sd + e_sys * e_sys);
                        if (sig - sig == 0.0) { isig = 1.0 / sig; lsig = log(sig); sig_bad = !(sig > 0); }
                        else isig = 0.0;             // infinite data error: upper limit
                        if (sig != sig) sig_bad = true;
                    }
                    // (two explicit accumulations: written as "chi += v" in one branch and "gp += ..." in the other, hipcc merges
                    //  them into ONE add on a two-element private array indexed by the branch -- scratch memory and an
                    //  s_waitcnt vmcnt(0) per datum)
                    double add_chi = 0.0, add_gp = 0.0;
                    if (isig != 0.0 || sig_bad) {
                        double v;
                        if (EXT && lim_finite) {         // uniform: truncated Gaussian with a finite detection limit
                            v = sig_bad ? dnan() : detection_term(c_m[u], est, 1.0 / isig, lsig, it.lim);
                        } else {
                            const double x = (c_m[u] - est) * isig;
                            v = (-(x * x) / 2.0 - kNormPdfLogC) - lsig;
                            if (!(est < dinf()) || sig_bad) v = dnan();
                        }
                        add_chi = v;
                    } else {
                        add_gp = upper_limit_term(c_m[u], est, e_sys);
                    }
                    opaque(add_chi); opaque(add_gp);
                    chi += add_chi; gp += add_gp;
                }
            }
        };
        stage_q(0);
        if constexpr (EXT)
            for (int u0 = NDL; u0 * G < nf; u0 += NDL) { stage_p(u0); stage_q(u0); }    // uniform: only beyond 2 G points
        if (dbt) { asm volatile("" : "+v"(chi)); dbg[100] = clock64(); }
        chi = group_sum(chi, G);
        if (it.has_ul) gp = group_sum(gp, G);
        if (dbt) { asm volatile("" : "+v"(chi)); dbg[101] = clock64(); }
        if (gi == G - 16) {
            chi_tot[o * TS + s] = chi;       // slot of the observed filter (= item order of the reference sum)
            gp_tot[o * TS + s] = gp;
            if (chi != chi) bad[s] = 1;
            if (chi_parts != nullptr && tile0 + s < B) {
                chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                gp_parts[(long)o * B + tile0 + s] = gp;
            }
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k + 1] = clock64();
        sync_signal(sync + W + 2 + k, lane);     // one signal per task
    };

    // ---------------------------------------------------------------------------------
    // Lean task (FASTM == 1: every item has constant systematics, no detection limit, no extinction, sample_times =
    // the equally spaced SVD grid, at most 32 points per filter, photometry staged in LDS as {t, m, 1/sigma, ln sigma}).
    // Straight-line code, two independent slots per lane whose instruction streams the compiler interleaves:
    //   TYPEB = false (17..32 points): a task = 4 samples x 16 lanes, slot u = datum gi + 16 u of the lane's sample;
    //   TYPEB = true  (<= 16 points) : a task = 8 samples, slot u = datum gi of sample 8 c + 4 u + g.
    // Every VALU instruction of a likelihood wave takes issue time from the f32 MFMA stream of its SIMD (f32 MFMA and
    // VALU share the SIMD's vector pipe: tools/ubench/valu_mix2.hip), and a chain of dependent instructions advances one
    // instruction per MFMA issued in between -- hence few instructions and two chains per wave.
    // The bracket of a datum on the sample's observer-frame grid is taken from the index guess without the exact
    // re-check of the extended task: the guess can differ from np.interp's bracket only when the epoch lies within
    // ~1e-13 of a grid node, where both brackets give the same value to rounding (linear interpolation is continuous).
    // ---------------------------------------------------------------------------------
    auto lean_task = [&](auto typeb_tag, auto two_tag, auto sys_tag, auto nonuni_tag, const int k, const int c) {
        // NONUNI: sample_times not equally spaced (the CLI's default log-spaced grid): branch-free bisection instead of the
        // index guess, and the node spacing from a table
        constexpr bool NONUNI = decltype(nonuni_tag)::value;
        constexpr bool TYPEB = decltype(typeb_tag)::value;
        // SYS: one sampled systematic per filter or shared (em_syserr): sigma_tot = sqrt(sigma_data^2 + e^2) per datum and sample,
        // with the extended task's expressions (the photometry record then carries sigma_data instead of 1 / sigma_tot)
        constexpr bool SYS = decltype(sys_tag)::value;
        // TWO: sample_times differ from the SVD grid -- each of a datum's two sample nodes is a stage-1 lerp between two
        // SVD rows (lightcurve_generation.py:177), evaluated in two passes of the same four FMA chains
        constexpr bool TWO = decltype(two_tag)::value;
        const ItemDesc& it = itab[k];
        const int o = it.o;
#ifdef NMMA_DBG_TASKSTAMPS      // diagnostic build: per-task stage stamps, dbg[128 + 8 (4 k + c) + j]
#define NM_TS(j) do { if (dbg && blockIdx.x == 0 && lane == 0 && k < 6 && c < 4) dbg[128 + 8 * (4 * k + c) + (j)] = clock64(); } while (0)
#else
#define NM_TS(j) do { } while (0)
#endif
        NM_TS(0);
#ifdef NMMA_DBG_TASKSTAMPS
        if (dbg && blockIdx.x == 0 && lane == 0 && k < 6 && c < 4) dbg[128 + 8 * (4 * k + c) + 7] = wave;
#endif
        if (c == 0) {      // this wave stages the item's basis rows (see fast_task)
            if (k >= NBUF) sync_wait(sync + W + 1 + (k - NBUF + 1), itab[k - NBUF].ntask[R - 1], P.watchdog, 800 + k);
            typedef __attribute__((address_space(3))) unsigned char* lds_bp;
            typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
            // (item-staged photometry: the item's own table, its records behind the filter's rows)
            // (dense: [b2 | records] only -- the basis rows are the A operands of the reconstruction, read from global memory)
            gbyte_p src = (LEANX && P.dat_in_tab) ? (gbyte_p)(uintptr_t)(P.tabi + (size_t)it.tabi * P.tabi_bytes + (DENSE ? P.tab_off_b2 : 0))
                                                  : (gbyte_p)(uintptr_t)(P.tab + (size_t)it.m * P.tab_bytes);
            lds_bp dst = (lds_bp)(tabl + (k % NBUF) * P.tab_fast_bytes);
            for (int q = 0; q * 1024 < P.tab_fast_bytes; ++q)
                __builtin_amdgcn_global_load_lds(src + q * 1024 + lane * 16, dst + q * 1024, 16, 0, 0);
        }
        NM_TS(1);
        typedef const __attribute__((address_space(3))) double* lds_cdp;
        typedef const __attribute__((address_space(3))) float* lds_cfp;
        typedef __attribute__((address_space(3))) double* lds_dp;
        const lds_cfp pbuf = (lds_cfp)(part + (k % NBUF) * (NSLICE * TS * PSTR));
        // (uniform descriptor words as scalars: comparisons on them are SALU work)
        const int jlo = __builtin_amdgcn_readfirstlane(it.jlo), jhi = __builtin_amdgcn_readfirstlane(it.jhi);
        const int d0 = __builtin_amdgcn_readfirstlane(it.d0), nf = __builtin_amdgcn_readfirstlane(it.nf);
        const int g = lane >> 4, gi = lane & 15;
        const double st0 = P.st0, inv_dt = P.st_inv_dt;
        const lds_cdp stl_l = (lds_cdp)stl;
        typedef __attribute__((ext_vector_type(2))) double f64x2;
        typedef const __attribute__((address_space(3))) f64x2* lds_c2p;
        // photometry: records {t, m | 1/sigma, ln sigma} of all points in LDS -- or, item-staged (EmDev::dat_in_tab), the epochs of
        // all points there (stage P) and the item's records in its ring slot (stage Q)
        const bool item_dat = LEANX && __builtin_amdgcn_readfirstlane(P.dat_in_tab) != 0;
        const lds_cdp tdat = (lds_cdp)(smem + L.dat);
        const int tstride = item_dat ? 1 : 4;                                   // doubles between the epochs of consecutive points
        const lds_c2p dat4 = item_dat ? (lds_c2p)(tabl + (k % NBUF) * P.tab_fast_bytes + P.tab_off_dat - (DENSE ? P.tab_off_b2 : 0))
                                      : (lds_c2p)(smem + L.dat);
        const int dbase = item_dat ? 0 : d0;
        const double st_lo = stl_l[jlo], st_hi = stl_l[jhi];
        const int nbis = NONUNI ? __builtin_amdgcn_readfirstlane(P.bg_nbis) : 0;
        const double bg_inv_h = NONUNI ? P.bg_inv_h : 0.0;
        const bool range_ok = jhi > jlo;
        constexpr int NSL = 2;
        int s_[NSL];
        s_[0] = TYPEB ? 8 * c + g : 4 * c + g;
        s_[1] = TYPEB ? s_[0] + 4 : s_[0];
        // ---- stage P (needs only the prologue)
        // (only what depends on the bracket stays in registers across the wait for the MLP: the photometry record and
        //  the sample scalars are read again from LDS in stage Q -- LDS reads cost the MFMA stream nothing, registers
        //  are what limits the workgroup to 16 waves)
        double dtx_[NSL];
        bool inside_[NSL], valid_[NSL];
        int lo_[NSL];
        lds_c2p D_[NSL];
        // The kernels that take further passes over a filter with more than 32 points (both slots of a lane then belong to ONE
        // sample) keep that sample's scalars and window in registers across the passes: 12 fewer VALU instructions per pass.
        constexpr bool HOIST = LEANX && !TYPEB;
        double h_zp1 = 0.0, h_tsh = 0.0, h_izp1 = 0.0, h_tlo = 0.0, h_thi = 0.0, h_dmrc = 0.0, h_izdt = 0.0, h_guess0 = 0.0;
        if constexpr (HOIST) {
            const lds_cdp sc = (lds_cdp)(scal + s_[0] * 8);
            h_zp1 = sc[S_ZP1]; h_tsh = sc[S_TS]; h_izp1 = sc[S_IZP1];
            h_tlo = st_lo * h_zp1 + h_tsh; h_thi = st_hi * h_zp1 + h_tsh;
            h_dmrc = sc[S_DMOD] + sc[S_RC];
            h_izdt = h_izp1 * inv_dt;
            h_guess0 = -((h_tsh * h_izp1 + st0) * inv_dt);
        }
        // (filters with 17 .. 32 points take one pass over the lane's two slots; more points further passes of 32: `pp`)
        auto stage_p = [&](const int pp) {
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int dd = TYPEB ? gi : gi + 16 * u + 32 * pp;
            valid_[u] = dd < nf;
            const int dix = valid_[u] ? dd : 0;
            D_[u] = dat4 + 2 * (dbase + dix);
            const double t = LEANX ? tdat[(d0 + dix) * tstride] : D_[u][0][0];
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double zp1 = HOIST ? h_zp1 : sc[S_ZP1], tsh = HOIST ? h_tsh : sc[S_TS], izp1 = HOIST ? h_izp1 : sc[S_IZP1];
            const double t_lo = HOIST ? h_tlo : st_lo * zp1 + tsh, t_hi = HOIST ? h_thi : st_hi * zp1 + tsh;
            inside_[u] = range_ok & (t >= t_lo) & (t <= t_hi);
            int lo;
            if constexpr (NONUNI) {
                // largest node index in [jlo, jhi - 1] whose observer-frame time is <= t (np.interp's bracket)
                // (narrowed first by the lookup over equal cells of the source-frame grid, one node of slack on either side for the
                //  rounding of the source-frame time: bg_nbis steps instead of ceil(log2 NS))
                typedef const __attribute__((address_space(3))) int* lds_cip_bg;
                const lds_cip_bg bgl = (lds_cip_bg)(stl_l + 2 * NS);
                int cq = (int)(((t - tsh) * izp1 - st0) * bg_inv_h);
                cq = cq < 0 ? 0 : (cq > BG_CELLS - 1 ? BG_CELLS - 1 : cq);
                lo = bgl[cq] - 1;
                int hi = bgl[cq + 1] + 2;
                lo = lo < jlo ? jlo : lo;
                hi = hi > jhi ? jhi : hi;
                for (int itb = 0; itb < nbis; ++itb) {         // uniform trip count
                    const int mid = (lo + hi) >> 1;
                    const bool le = (stl_l[mid] * zp1 + tsh) <= t;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid;
                }
                lo = lo > jhi - 1 ? jhi - 1 : lo;
            } else {
                // (HOIST: the same guess from one FMA on per-sample constants -- it may differ from the unfused form only for an
                //  epoch within rounding of a node, where both brackets give the same value)
                lo = HOIST ? (int)floor(fma(t, h_izdt, h_guess0)) : (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                lo = lo > jhi - 1 ? jhi - 1 : lo;
                lo = lo < jlo ? jlo : lo;
            }
            dtx_[u] = t - (stl_l[lo] * zp1 + tsh);            // t - x0
            lo_[u] = lo;
            asm volatile("" : "+v"(dtx_[u]), "+v"(lo_[u]));   // (evaluated here, before the wait for the MLP)
        }
        };
        stage_p(0);
        // dense: the A operands of this task's node tiles -- rows of [VA o span | mins | 0] on the sample grid (the stage-1 lerp
        // between SVD nodes folded in), pre-swizzled per filter at create
        // (EmDev::dva, one coalesced 512-byte load per MFMA) -- requested before the waits for the surrogate and the node buffer
        const int dn_tt = (NS + 15) >> 4;          // node tiles of the SAMPLE grid (<= 16: four per task, checked at create)
        double dav[DENSE ? 4 : 1][3];
        if constexpr (DENSE) {
            gcf64p dva = as_global(P.dva) + (size_t)it.m * dn_tt * 3 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = (c & 3) + 4 * q;
#pragma unroll
                for (int step = 0; step < 3; ++step) dav[q][step] = tt < dn_tt ? dva[(tt * 3 + step) * 64] : 0.0;
            }
        }
        NM_TS(2);
        // ---- stage Q (needs the coefficients of item k)
        if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            NM_TS(3);
            sync_signal(sync + 2 * W + 4 + k, lane);     // rows of item k staged
        }
        sync_wait(sync + 2 * W + 4 + k, 1, P.watchdog, 350 + k);
        NM_TS(4);
        sync_wait(sync + k, NMW, P.watchdog, 300 + k);   // coefficients of item k published
        NM_TS(5);
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k] = clock64();
        const unsigned char* tbl = tabl + (k % NBUF) * P.tab_fast_bytes;
        const lds_cfp b2l = (lds_cfp)(tbl + (DENSE ? 0 : P.tab_off_b2));
        const float b2v = b2l[gi];
        constexpr int NCC = TYPEB ? 2 : 1;
        lds_c2p cc_[NCC];          // the sample's 10 coefficients (fp64) in this wave's LDS slots: [wave][q][g][16]
        // ---- dense: the four tasks of (item k, 16 samples) reconstruct ALL nodes of those samples together,
        //      mag[node][sample] = (VA[node, :] . c[sample, :]) span[node] + mins[node], 16 nodes x 16 samples per
        //      v_mfma_f64_16x16x4_f64 (K = NC in three steps, zero-padded; operand layout as in em_fused) into one of
        //      DENSE_NBUF LDS buffers; a datum then READS its two node magnitudes instead of reconstructing two rows.
        lds_cdp nodes_l = nullptr;
        int* unit_done = nullptr;
        if constexpr (DENSE) {
            const int h = c >> 2;                                 // 4 samples per task: tasks 4h .. 4h + 3 share the 16 samples of half h
            const int unit = R * k + h;
            int* const unit_prod = sync + 3 * W + 4 + unit;
            unit_done = sync + 3 * W + 4 + R * W + unit;
            const lds_dp nb = (lds_dp)(smem + L.nodes) + (unit % DENSE_NBUF) * (((NS + 15) & ~15) * DENSE_STRIDE);
            nodes_l = (lds_cdp)nb;
            // B operands: coefficient 4 step + lane / 16 of sample 16 h + lane % 16 (slice sums in the fixed order, + b2, as fp64);
            // "coefficient" NC is the constant 1 that multiplies the mins column of the A table
            const int sj = 16 * h + (lane & 15), kq = lane >> 4;
            double bq[3];
#pragma unroll
            for (int step = 0; step < 3; ++step) {
                const int kc = 4 * step + kq;
                const lds_cfp pp = pbuf + ((sj >> 4) * 16 + (sj & 15)) * PSTR + (kc < 16 ? kc : 0);
                float cm = pp[0];
#pragma unroll
                for (int w = 1; w < NSLICE; ++w) cm += pp[w * (R * 16 * PSTR)];
                cm += b2l[kc < 16 ? kc : 0];
                bq[step] = kc < NC ? (double)cm : (kc == NC ? 1.0 : 0.0);
            }
            // the buffer's previous unit has been consumed by all four of its tasks
            if (unit >= DENSE_NBUF) sync_wait(unit_done - DENSE_NBUF, 4, P.watchdog, 360 + k);
            typedef double f64x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = (c & 3) + 4 * q;                  // this task's node tiles (uniform)
                if (tt >= dn_tt) break;
                f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int step = 0; step < 3; ++step) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dav[q][step], bq[step], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) nb[(16 * tt + 4 * r + kq) * DENSE_STRIDE + (lane & 15)] = acc[r];
            }
            sync_signal(unit_prod, lane);
            sync_wait(unit_prod, 4, P.watchdog, 370 + k);        // all node tiles of the unit are in LDS
        }
#pragma unroll
        for (int q = 0; q < (DENSE ? 0 : NCC); ++q) {
            // slice reduction (fixed order) + bias of the second Dense: lane gi owns coefficient gi of its sample
            const int s = s_[q];
            const lds_cfp pp = pbuf + ((s >> 4) * 16 + (s & 15)) * PSTR + gi;
            float cmine = pp[0];
#pragma unroll
            for (int w = 1; w < NSLICE; ++w) cmine += pp[w * (R * 16 * PSTR)];
            cmine += b2v;
            const lds_dp cslot = (lds_dp)(cdl + ((wave * 2 + q) * 4 + g) * 16);
            cslot[gi] = (double)cmine;
            cc_[q] = (lds_c2p)cslot;
        }
        // the two basis rows of every slot, read as 16-byte pairs [VA[2j], VA[2j+1]] (pair 5 = [span, mins]); the four FMA
        // chains (2 slots x 2 rows) advance together, one pair per step, so that no instruction waits for its predecessor.
        // TWO: pass 0 reconstructs the SVD rows around sample node lo, pass 1 those around node lo + 1.
        const lds_c2p rows2 = (lds_c2p)(tbl);
        const lds_cdp s1of_l = (lds_cdp)(tbl + P.tab_off_s1of), s1inv_l = (lds_cdp)(tbl + P.tab_off_s1inv);
        typedef const __attribute__((address_space(3))) int* lds_cip;
        const lds_cip s1i_l = (lds_cip)(tbl + P.tab_off_s1i);
        double v_[NSL], gp_[NSL], esys_[NSL] = {0.0, 0.0};
        const int sv0 = SYS ? __builtin_amdgcn_readfirstlane(P.sys_off[o]) : 0;      // first slot of the filter's parameter
        const lds_cdp ext_l = (lds_cdp)(smem + L.exttab) + (P.has_ebv ? k : 0) * TS;
        auto stage_q = [&]() {
        double ynode_[2][NSL];            // magnitudes at the two sample nodes of every slot
        if constexpr (DENSE) {
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const lds_cdp nd = nodes_l + lo_[u] * DENSE_STRIDE + (s_[0] & 15);
                ynode_[0][u] = nd[0]; ynode_[1][u] = nd[DENSE_STRIDE];
            }
        }
#pragma unroll
        for (int pass = 0; pass < (DENSE ? 0 : (TWO ? 2 : 1)); ++pass) {
            lds_c2p ra_[NSL], rb_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    int i1 = s1i_l[j];
                    i1 = i1 < 0 ? 0 : i1;                         // (nodes outside the SVD grid lie outside [jlo, jhi]: never bracketed)
                    const int i2 = i1 + 1 < NT ? i1 + 1 : NT - 1;
                    ra_[u] = rows2 + i1 * 6; rb_[u] = rows2 + i2 * 6;
                } else {
                    ra_[u] = rows2 + lo_[u] * 6; rb_[u] = ra_[u] + 6;
                }
            }
            double a0_[NSL], a1_[NSL];
            f64x2 p0_[NSL], p1_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) { p0_[u] = ra_[u][0]; p1_[u] = rb_[u][0]; }
            f64x2 cq_[NCC];
#pragma unroll
            for (int q = 0; q < NCC; ++q) cq_[q] = cc_[q][0];
#pragma unroll
            for (int jp = 0; jp < 5; ++jp) {
                f64x2 n0_[NSL], n1_[NSL], nq_[NCC];
#pragma unroll
                for (int u = 0; u < NSL; ++u) { n0_[u] = ra_[u][jp + 1]; n1_[u] = rb_[u][jp + 1]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) nq_[q] = cc_[q][jp + 1];
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    if (jp == 0) { a0_[u] = p0_[u][0] * cq[0]; a1_[u] = p1_[u][0] * cq[0]; }
                    else { a0_[u] = fma(p0_[u][0], cq[0], a0_[u]); a1_[u] = fma(p1_[u][0], cq[0], a1_[u]); }
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    a0_[u] = fma(p0_[u][1], cq[1], a0_[u]); a1_[u] = fma(p1_[u][1], cq[1], a1_[u]);
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) { p0_[u] = n0_[u]; p1_[u] = n1_[u]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) cq_[q] = nq_[q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double ya = a0_[u] * p0_[u][0] + p0_[u][1], yb = a1_[u] * p1_[u][0] + p1_[u][1];   // (VA[i,:].c) span[i] + mins[i]
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    // stage 1: ((yb - ya) / dx) * off + ya with the reciprocal of dx from the table
                    ynode_[pass][u] = ((yb - ya) * s1inv_l[j]) * s1of_l[j] + ya;
                } else {
                    ynode_[0][u] = ya; ynode_[1][u] = yb;
                }
            }
        }
        double est_[NSL], m_[NSL];
        bool ul_[NSL];
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const f64x2 tm = D_[u][0], sl = D_[u][1];       // {t, m}, {1/sigma, ln sigma}
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double dmrc = HOIST ? h_dmrc : sc[S_DMOD] + sc[S_RC];
            const double izdt = (HOIST && !NONUNI) ? h_izdt : (HOIST ? h_izp1 : sc[S_IZP1]) * (NONUNI ? stl_l[NS + lo_[u]] : inv_dt);
            double y0 = ynode_[0][u], y1 = ynode_[1][u];
            double est;
            if constexpr (HOIST) {
                // the sample's offsets (extinction of this item + distance modulus + K-correction) added once, to the left node:
                // the slope is the difference of the node magnitudes themselves
                const double yl = (y0 + ext_l[s_[0]]) + dmrc;
                est = ((y1 - y0) * izdt) * dtx_[u] + yl;
            } else {
                if constexpr (LEANX) {       // extinction magnitude of this sample and item (filled by the prologue, model.py:323-342)
                    const double ext = ext_l[s_[u]];       // (no branch here: 0 when there is no extinction)
                    y0 = y0 + ext; y1 = y1 + ext;
                }
                y0 = y0 + dmrc; y1 = y1 + dmrc;
                est = ((y1 - y0) * izdt) * dtx_[u] + y0;
            }
            double isig = sl[0], lsig = sl[1];
            bool sig_bad = false;
            if constexpr (SYS) {
                const double sd = sl[0];                      // sigma_data
                const double e_sys = ((lds_cdp)(smem + L.epar))[sv0 * TS + s_[u]];
                esys_[u] = e_sys;
                // (1 / sigma_tot and ln sigma_tot without the library's sqrt, division and log: rsqrt_pos / log_pos above.  The exact
                //  shape of these five lines matters to hipcc: taking the log of sigma^2 instead, or the finite test from sqrt(s2),
                //  made the lean kernels spill ~770 registers)
                const double s2 = sd * sd + e_sys * e_sys;    // sigma_tot^2
                const double rs = rsqrt_pos(s2);
                const double sig = s2 * rs;
                const bool fin = (s2 - s2 == 0.0);
                isig = fin ? rs : 0.0;                        // infinite data error: upper limit
                lsig = log_pos(sig);
                sig_bad = (fin & !(s2 > 0)) | (s2 != s2);
            }
            const double x = (tm[1] - est) * isig;
            double v = (-(x * x) / 2.0 - kNormPdfLogC) - lsig;
            opaque(v);                                        // (computed on every lane: no exec-masked region around the chain)
            // outside the model window est = +inf: truncnorm.logpdf(loc = inf) = NaN (em_likelihood.py:252-256)
            v = (inside_[u] & !sig_bad) ? v : dnan();
            ul_[u] = valid_[u] & (isig == 0.0) & !sig_bad;    // infinite data error: an upper limit
            v_[u] = (valid_[u] & !ul_[u]) ? v : 0.0;
            est_[u] = est; m_[u] = tm[1];
        }
        gp_[0] = 0.0; gp_[1] = 0.0;
        if (it.has_ul) {                                    // uniform; the term itself only on the lanes that hold a limit
#pragma unroll
            for (int u = 0; u < NSL; ++u)
                if (ul_[u]) gp_[u] = upper_limit_term(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const);
        }
        };
        stage_q();
        lds_dp chi_l = (lds_dp)chi_tot;
        lds_dp gp_l = (lds_dp)gp_tot;
        if constexpr (TYPEB) {
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double chi = group_sum(v_[u], 16);
                double gp = 0.0;
                if (it.has_ul) gp = group_sum(gp_[u], 16);
                if (gi == 0) {
                    const int s = s_[u];
                    chi_l[o * TS + s] = chi;
                    gp_l[o * TS + s] = gp;
                    if (chi != chi) bad[s] = 1;
                    if (chi_parts != nullptr && tile0 + s < B) {
                        chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                        gp_parts[(long)o * B + tile0 + s] = gp;
                    }
                }
            }
        } else {
            double vacc = v_[0] + v_[1], gacc = gp_[0] + gp_[1];
            if constexpr (LEANX) {
                for (int pp = 1; pp * 32 < nf; ++pp) {      // uniform: only filters with more than 32 points
                    stage_p(pp);
                    stage_q();
                    vacc += v_[0] + v_[1]; gacc += gp_[0] + gp_[1];
                }
            }
            const double chi = group_sum(vacc, 16);
            double gp = 0.0;
            if (it.has_ul) gp = group_sum(gacc, 16);
            if (gi == 0) {
                const int s = s_[0];
                chi_l[o * TS + s] = chi;
                gp_l[o * TS + s] = gp;
                if (chi != chi) bad[s] = 1;
                if (chi_parts != nullptr && tile0 + s < B) {
                    chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                    gp_parts[(long)o * B + tile0 + s] = gp;
                }
            }
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k + 1] = clock64();
        NM_TS(6);
        if constexpr (DENSE) sync_signal(unit_done, lane);      // this task no longer reads the unit's node buffer
        sync_signal(sync + W + 2 + k, lane);     // one signal per task
#undef NM_TS
    };

    // ---------------------------------------------------------------------------------
    // lean_gen_task (FASTM = 5): the general lean task -- lean_task's extras (passes of 32 points, extinction table, sampled
    // systematics, unequally spaced grids) plus
    //  * averaged bands: an observed filter whose magnitude is the mean of several model filters (utils.py:566-584).  k is the
    //    LAST source item of the band (items k - nsrc + 1 .. k, one surrogate each; nsrc = 1 for an ordinary band); stage Q
    //    walks the sources -- wait for the source's coefficients, node magnitudes with its basis rows, + extinction of its
    //    filter + distance modulus -- sums them in source order and divides the interpolated sum by nsrc (the generic item
    //    phase's order).  The other source items own no tasks; every task of the band signals each of them, which releases
    //    their ring slots.
    //  * time-node systematics (systematics.py:288-291): see SYS below.
    // A copy of lean_task rather than a variant of it: any change to that lambda, even a semantically neutral one, moves
    // hipcc's register allocation off its optimum in every instantiation.
    // ---------------------------------------------------------------------------------
    auto lean_gen_task = [&](auto typeb_tag, auto two_tag, auto sys_tag, auto nonuni_tag, const int k, const int c) {
        // NONUNI: sample_times not equally spaced (the CLI's default log-spaced grid): branch-free bisection instead of the
        // index guess, and the node spacing from a table
        constexpr bool NONUNI = decltype(nonuni_tag)::value;
        constexpr bool TYPEB = decltype(typeb_tag)::value;
        // SYS: sampled systematics -- one parameter per filter or shared (em_syserr), or parameters at time nodes, constant
        // outside and linear in between (systematics.py:288-291): sigma_tot = sqrt(sigma_data^2 + e^2) per datum and sample.
        // The photometry record then carries [t | m | sigma_data | q], q = node index + fraction of the datum's node interval
        // (0 for a single parameter; written by nmma_em_create), and e = v[i] + (v[i + 1] - v[i]) * fraction
        constexpr bool SYS = decltype(sys_tag)::value;
        // TWO: sample_times differ from the SVD grid -- each of a datum's two sample nodes is a stage-1 lerp between two
        // SVD rows (lightcurve_generation.py:177), evaluated in two passes of the same four FMA chains
        constexpr bool TWO = decltype(two_tag)::value;
        const ItemDesc& it = itab[k];
        const int o = it.o;
        const int nsrc = __builtin_amdgcn_readfirstlane(it.nsrc);
        const int k0 = k - (nsrc - 1);
        if (c == 0) {      // this wave stages the basis rows of every source (the host guarantees NBUF >= nsrc)
            typedef __attribute__((address_space(3))) unsigned char* lds_bp;
            typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
            for (int kk = k0; kk <= k; ++kk) {
                if (kk >= NBUF) sync_wait(sync + W + 1 + (kk - NBUF + 1), itab[kk - NBUF].ntask[R - 1], P.watchdog, 800 + kk);
                gbyte_p src = (gbyte_p)(uintptr_t)(P.tab + (size_t)itab[kk].m * P.tab_bytes);
                lds_bp dst = (lds_bp)(tabl + (kk % NBUF) * P.tab_fast_bytes);
                for (int q = 0; q * 1024 < P.tab_fast_bytes; ++q)
                    __builtin_amdgcn_global_load_lds(src + q * 1024 + lane * 16, dst + q * 1024, 16, 0, 0);
            }
        }
        typedef const __attribute__((address_space(3))) double* lds_cdp;
        typedef const __attribute__((address_space(3))) float* lds_cfp;
        typedef __attribute__((address_space(3))) double* lds_dp;
        // (uniform descriptor words as scalars: comparisons on them are SALU work)
        const int jlo = __builtin_amdgcn_readfirstlane(it.jlo), jhi = __builtin_amdgcn_readfirstlane(it.jhi);
        const int d0 = __builtin_amdgcn_readfirstlane(it.d0), nf = __builtin_amdgcn_readfirstlane(it.nf);
        const int g = lane >> 4, gi = lane & 15;
        const double st0 = P.st0, inv_dt = P.st_inv_dt;
        const lds_cdp stl_l = (lds_cdp)stl;
        typedef __attribute__((ext_vector_type(2))) double f64x2;
        typedef const __attribute__((address_space(3))) f64x2* lds_c2p;
        const lds_c2p dat4 = (lds_c2p)(smem + L.dat);
        const double st_lo = stl_l[jlo], st_hi = stl_l[jhi];
        const int nbis = NONUNI ? __builtin_amdgcn_readfirstlane(P.bg_nbis) : 0;
        const double bg_inv_h = NONUNI ? P.bg_inv_h : 0.0;
        const bool range_ok = jhi > jlo;
        constexpr int NSL = 2;
        int s_[NSL];
        s_[0] = TYPEB ? 8 * c + g : 4 * c + g;
        s_[1] = TYPEB ? s_[0] + 4 : s_[0];
        // ---- stage P (needs only the prologue)
        // (only what depends on the bracket stays in registers across the wait for the MLP: the photometry record and
        //  the sample scalars are read again from LDS in stage Q -- LDS reads cost the MFMA stream nothing, registers
        //  are what limits the workgroup to 16 waves)
        double dtx_[NSL];
        bool inside_[NSL], valid_[NSL];
        int lo_[NSL];
        lds_c2p D_[NSL];
        // (filters with 17 .. 32 points take one pass over the lane's two slots; more points further passes of 32: `pp`)
        auto stage_p = [&](const int pp) {
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int dd = TYPEB ? gi : gi + 16 * u + 32 * pp;
            valid_[u] = dd < nf;
            D_[u] = dat4 + 2 * (d0 + (valid_[u] ? dd : 0));
            const double t = D_[u][0][0];
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double zp1 = sc[S_ZP1], tsh = sc[S_TS], izp1 = sc[S_IZP1];
            const double t_lo = st_lo * zp1 + tsh, t_hi = st_hi * zp1 + tsh;
            inside_[u] = range_ok & (t >= t_lo) & (t <= t_hi);
            int lo;
            if constexpr (NONUNI) {
                // largest node index in [jlo, jhi - 1] whose observer-frame time is <= t (np.interp's bracket)
                // (narrowed first by the lookup over equal cells of the source-frame grid, one node of slack on either side for the
                //  rounding of the source-frame time: bg_nbis steps instead of ceil(log2 NS))
                typedef const __attribute__((address_space(3))) int* lds_cip_bg;
                const lds_cip_bg bgl = (lds_cip_bg)(stl_l + 2 * NS);
                int cq = (int)(((t - tsh) * izp1 - st0) * bg_inv_h);
                cq = cq < 0 ? 0 : (cq > BG_CELLS - 1 ? BG_CELLS - 1 : cq);
                lo = bgl[cq] - 1;
                int hi = bgl[cq + 1] + 2;
                lo = lo < jlo ? jlo : lo;
                hi = hi > jhi ? jhi : hi;
                for (int itb = 0; itb < nbis; ++itb) {         // uniform trip count
                    const int mid = (lo + hi) >> 1;
                    const bool le = (stl_l[mid] * zp1 + tsh) <= t;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid;
                }
                lo = lo > jhi - 1 ? jhi - 1 : lo;
            } else {
                lo = (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                lo = lo > jhi - 1 ? jhi - 1 : lo;
                lo = lo < jlo ? jlo : lo;
            }
            dtx_[u] = t - (stl_l[lo] * zp1 + tsh);            // t - x0
            lo_[u] = lo;
            asm volatile("" : "+v"(dtx_[u]), "+v"(lo_[u]));   // (evaluated here, before the wait for the MLP)
        }
        };
        stage_p(0);
        // ---- stage Q (needs the coefficients of item k)
        if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int kk = k0; kk <= k; ++kk) sync_signal(sync + 2 * W + 4 + kk, lane);     // rows of the band's sources staged
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k] = clock64();
        constexpr int NCC = TYPEB ? 2 : 1;
        // the two basis rows of every slot, read as 16-byte pairs [VA[2j], VA[2j+1]] (pair 5 = [span, mins]); the four FMA
        // chains (2 slots x 2 rows) advance together, one pair per step, so that no instruction waits for its predecessor.
        // TWO: pass 0 reconstructs the SVD rows around sample node lo, pass 1 those around node lo + 1.
        typedef const __attribute__((address_space(3))) int* lds_cip;
        double v_[NSL], gp_[NSL], esys_[NSL] = {0.0, 0.0};
        // a finite detection limit (uniform per band): the truncated Gaussian of em_likelihood.py:252-256 through detection_term,
        // evaluated after the straight-line term like the upper limits -- a call per datum, on the lanes that hold a detection
        const bool lim_fin = (it.lim - it.lim == 0.0);
        double isig_[NSL], lsig_[NSL];
        bool sbad_[NSL];
        const int sv0 = SYS ? __builtin_amdgcn_readfirstlane(P.sys_off[o]) : 0;      // first slot of the filter's parameter(s)
        const int svl = SYS ? __builtin_amdgcn_readfirstlane(P.sys_nn[o]) - 1 : 0;   // last node
        auto stage_q = [&]() {
        double ys_[2][NSL] = {{0.0, 0.0}, {0.0, 0.0}};     // sums over the sources of the node magnitudes
        for (int kk = k0; kk <= k; ++kk) {               // uniform trip count
        sync_wait(sync + 2 * W + 4 + kk, 1, P.watchdog, 350 + kk);
        sync_wait(sync + kk, NMW, P.watchdog, 300 + kk);   // coefficients of source item kk published
        const unsigned char* tbl = tabl + (kk % NBUF) * P.tab_fast_bytes;
        const lds_cfp b2l = (lds_cfp)(tbl + P.tab_off_b2);
        const float b2v = b2l[gi];
        lds_c2p cc_[NCC];          // the sample's 10 coefficients (fp64) in this wave's LDS slots: [wave][q][g][16]
#pragma unroll
        for (int q = 0; q < NCC; ++q) {
            // slice reduction (fixed order) + bias of the second Dense: lane gi owns coefficient gi of its sample
            const int s = s_[q];
            const lds_cfp pp = (lds_cfp)(part + (kk % NBUF) * (NSLICE * TS * PSTR)) + ((s >> 4) * 16 + (s & 15)) * PSTR + gi;
            float cmine = pp[0];
#pragma unroll
            for (int w = 1; w < NSLICE; ++w) cmine += pp[w * (R * 16 * PSTR)];
            cmine += b2v;
            const lds_dp cslot = (lds_dp)(cdl + ((wave * 2 + q) * 4 + g) * 16);
            cslot[gi] = (double)cmine;
            cc_[q] = (lds_c2p)cslot;
        }
        const lds_c2p rows2 = (lds_c2p)(tbl);
        const lds_cdp s1of_l = (lds_cdp)(tbl + P.tab_off_s1of), s1inv_l = (lds_cdp)(tbl + P.tab_off_s1inv);
        const lds_cip s1i_l = (lds_cip)(tbl + P.tab_off_s1i);
        const lds_cdp ext_l = (lds_cdp)(smem + L.exttab) + (P.has_ebv ? kk : 0) * TS;
        double ynode_[2][NSL];            // magnitudes at the two sample nodes of every slot
#pragma unroll
        for (int pass = 0; pass < (TWO ? 2 : 1); ++pass) {
            lds_c2p ra_[NSL], rb_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    int i1 = s1i_l[j];
                    i1 = i1 < 0 ? 0 : i1;                         // (nodes outside the SVD grid lie outside [jlo, jhi]: never bracketed)
                    const int i2 = i1 + 1 < NT ? i1 + 1 : NT - 1;
                    ra_[u] = rows2 + i1 * 6; rb_[u] = rows2 + i2 * 6;
                } else {
                    ra_[u] = rows2 + lo_[u] * 6; rb_[u] = ra_[u] + 6;
                }
            }
            double a0_[NSL], a1_[NSL];
            f64x2 p0_[NSL], p1_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) { p0_[u] = ra_[u][0]; p1_[u] = rb_[u][0]; }
            f64x2 cq_[NCC];
#pragma unroll
            for (int q = 0; q < NCC; ++q) cq_[q] = cc_[q][0];
#pragma unroll
            for (int jp = 0; jp < 5; ++jp) {
                f64x2 n0_[NSL], n1_[NSL], nq_[NCC];
#pragma unroll
                for (int u = 0; u < NSL; ++u) { n0_[u] = ra_[u][jp + 1]; n1_[u] = rb_[u][jp + 1]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) nq_[q] = cc_[q][jp + 1];
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    if (jp == 0) { a0_[u] = p0_[u][0] * cq[0]; a1_[u] = p1_[u][0] * cq[0]; }
                    else { a0_[u] = fma(p0_[u][0], cq[0], a0_[u]); a1_[u] = fma(p1_[u][0], cq[0], a1_[u]); }
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    a0_[u] = fma(p0_[u][1], cq[1], a0_[u]); a1_[u] = fma(p1_[u][1], cq[1], a1_[u]);
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) { p0_[u] = n0_[u]; p1_[u] = n1_[u]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) cq_[q] = nq_[q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double ya = a0_[u] * p0_[u][0] + p0_[u][1], yb = a1_[u] * p1_[u][0] + p1_[u][1];   // (VA[i,:].c) span[i] + mins[i]
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    // stage 1: ((yb - ya) / dx) * off + ya with the reciprocal of dx from the table
                    ynode_[pass][u] = ((yb - ya) * s1inv_l[j]) * s1of_l[j] + ya;
                } else {
                    ynode_[0][u] = ya; ynode_[1][u] = yb;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NSL; ++u) {    // + extinction of this source's filter (0 without) + distance modulus (model.py:323-342)
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double dmrc = sc[S_DMOD] + sc[S_RC], ext = ext_l[s_[u]];
            ys_[0][u] += (ynode_[0][u] + ext) + dmrc; ys_[1][u] += (ynode_[1][u] + ext) + dmrc;
        }
        }
        double est_[NSL], m_[NSL];
        bool ul_[NSL];
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const f64x2 tm = D_[u][0], sl = D_[u][1];       // {t, m}, {1/sigma, ln sigma}
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double izdt = sc[S_IZP1] * (NONUNI ? stl_l[NS + lo_[u]] : inv_dt);
            const double y0 = ys_[0][u], y1 = ys_[1][u];
            const double est = (((y1 - y0) * izdt) * dtx_[u] + y0) / (double)nsrc;     // (a + b [+ c]) / n  (utils.py:566-584)
            double isig = sl[0], lsig = sl[1];
            bool sig_bad = false;
            if constexpr (SYS) {
                const double sd = sl[0];                      // sigma_data
                const double qf = floor(sl[1]), fr = sl[1] - qf;
                const int i0 = (int)qf, i1 = i0 < svl ? i0 + 1 : svl;
                const lds_cdp ep = (lds_cdp)(smem + L.epar) + sv0 * TS + s_[u];
                const double e0 = ep[i0 * TS], e1 = ep[i1 * TS];
                const double e_sys = (e1 - e0) * fr + e0;     // (a single parameter: i0 = i1 = 0, fr = 0 -- exactly e0)
                esys_[u] = e_sys;
                // (1 / sigma_tot and ln sigma_tot without the library's sqrt, division and log: rsqrt_pos / log_pos above.  The exact
                //  shape of these five lines matters to hipcc: taking the log of sigma^2 instead, or the finite test from sqrt(s2),
                //  made the lean kernels spill ~770 registers)
                const double s2 = sd * sd + e_sys * e_sys;    // sigma_tot^2
                const double rs = rsqrt_pos(s2);
                const double sig = s2 * rs;
                const bool fin = (s2 - s2 == 0.0);
                isig = fin ? rs : 0.0;                        // infinite data error: upper limit
                lsig = log_pos(sig);
                sig_bad = (fin & !(s2 > 0)) | (s2 != s2);
            }
            const double x = (tm[1] - est) * isig;
            double v = (-(x * x) / 2.0 - kNormPdfLogC) - lsig;
            opaque(v);                                        // (computed on every lane: no exec-masked region around the chain)
            // outside the model window est = +inf: truncnorm.logpdf(loc = inf) = NaN (em_likelihood.py:252-256)
            v = (inside_[u] & !sig_bad) ? v : dnan();
            ul_[u] = valid_[u] & (isig == 0.0) & !sig_bad;    // infinite data error: an upper limit
            v_[u] = (valid_[u] & !ul_[u]) ? v : 0.0;
            est_[u] = est; m_[u] = tm[1];
            isig_[u] = isig; lsig_[u] = lsig; sbad_[u] = sig_bad;
        }
        if (lim_fin) {
#pragma unroll
            for (int u = 0; u < NSL; ++u)
                if (valid_[u] & !ul_[u])
                    v_[u] = sbad_[u] ? dnan() : detection_term(m_[u], inside_[u] ? est_[u] : dinf(), 1.0 / isig_[u], lsig_[u], it.lim);
        }
        gp_[0] = 0.0; gp_[1] = 0.0;
        if (it.has_ul) {                                    // uniform; the term itself only on the lanes that hold a limit
#pragma unroll
            for (int u = 0; u < NSL; ++u)
                if (ul_[u]) gp_[u] = upper_limit_term(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const);
        }
        };
        stage_q();
        lds_dp chi_l = (lds_dp)chi_tot;
        lds_dp gp_l = (lds_dp)gp_tot;
        if constexpr (TYPEB) {
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double chi = group_sum(v_[u], 16);
                double gp = 0.0;
                if (it.has_ul) gp = group_sum(gp_[u], 16);
                if (gi == 0) {
                    const int s = s_[u];
                    chi_l[k * TS + s] = chi;
                    gp_l[k * TS + s] = gp;
                    if (chi != chi) bad[s] = 1;
                    if (chi_parts != nullptr && tile0 + s < B) {
                        chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                        gp_parts[(long)o * B + tile0 + s] = gp;
                    }
                }
            }
        } else {
            double vacc = v_[0] + v_[1], gacc = gp_[0] + gp_[1];
            if constexpr (LEANX) {
                for (int pp = 1; pp * 32 < nf; ++pp) {      // uniform: only filters with more than 32 points
                    stage_p(pp);
                    stage_q();
                    vacc += v_[0] + v_[1]; gacc += gp_[0] + gp_[1];
                }
            }
            const double chi = group_sum(vacc, 16);
            double gp = 0.0;
            if (it.has_ul) gp = group_sum(gacc, 16);
            if (gi == 0) {
                const int s = s_[0];
                chi_l[k * TS + s] = chi;
                gp_l[k * TS + s] = gp;
                if (chi != chi) bad[s] = 1;
                if (chi_parts != nullptr && tile0 + s < B) {
                    chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                    gp_parts[(long)o * B + tile0 + s] = gp;
                }
            }
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k + 1] = clock64();
        for (int kk = k0; kk <= k; ++kk) sync_signal(sync + W + 2 + kk, lane);     // one signal per task and source item
    };

    if constexpr (FAST) {
        sync_wait(sync + W + 1, NVW, P.watchdog, 400);   // prologue data of every likelihood wave in LDS
        if constexpr (WALKF) {
            // ---- the fused MCMC step, first phase (the LAST likelihood wave, before it claims tasks): everything that does not
            // depend on log L -- the chains' uniforms, the two live points of each move, the chains' state -- is loaded NOW and
            // parked in the prologue's staging area (free from here on), so that the epilogue finds it in LDS instead of waiting for
            // a chain of dependent L2 round trips after the tile's last task.  At priority 0: it has all of the launch to finish
            // and must not take issue slots from the MFMA stream.
            if (!helper && vwave == NVW - 1) {
                __builtin_amdgcn_s_setprio(0);
                double* wl = reinterpret_cast<double*>(smem + L.stage);
                double* pl = wl + TS;                         // [5][WNR * 64]: live_j - live_i | u | v | proposal | theta, per (round, lane)
                double* pcd = pl + 5 * WNR * 64;              // [2][TS]: gamma | bound, per chain
                int* pci = reinterpret_cast<int*>(pcd + 2 * TS);      // [6][TS]: inside | active | counts[4]
                {
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(&wf->priors[0]);
                    for (int j = lane; j < wf->ndim * 10; j += 64) reinterpret_cast<uint32_t*>(wspl)[j] = src[j];
                }
#ifndef NMMA_DBG_WALK_NOPRE
                constexpr int WPR = WNR < 2 ? WNR : 2;       // rounds in flight together
#pragma unroll
                for (int r0 = 0; r0 < WNR; r0 += WPR) {
                    WalkPre wq[WPR];
                    WalkPreKey wkey[WPR];
                    double wr[WPR][7];
#pragma unroll
                    for (int rr = 0; rr < WPR; ++rr) {   // the pair's state loads first ...
                        const long c = tile0 + (r0 + rr) * WCR + lane / WT;
                        walk_step_pre_a(wf->ndim, c < B ? c : B - 1, lane & (WT - 1), wf->key, wf->u, wf->v, wf->prop, theta, wf->inside, wf->loglstar, wf->counts,
                                        wf->n_steps, (uint64_t)wstep, wq[rr], wkey[rr]);
                    }
#pragma unroll
                    for (int rr = 0; rr < WPR; ++rr)     // ... then the hashes and the live points they address ...
                        walk_step_pre_b(wf->ndim, lane & (WT - 1), wf->live, (long)wf->n_live, wf->first_step + (uint64_t)wstep, wkey[rr], wq[rr], wr[rr]);
#pragma unroll
                    for (int rr = 0; rr < WPR; ++rr) {   // ... then the move's scale; park everything
                        const int r = r0 + rr;
                        walk_step_pre_c(wf->ndim, wr[rr], wq[rr]);
                        const int e = r * 64 + lane;
                        pl[e] = wq[rr].lj - wq[rr].li; pl[WNR * 64 + e] = wq[rr].uu; pl[2 * WNR * 64 + e] = wq[rr].vv; pl[3 * WNR * 64 + e] = wq[rr].pp;
                        pl[4 * WNR * 64 + e] = wq[rr].th;
                        if ((lane & (WT - 1)) == 0) {
                            const int cl = r * WCR + lane / WT;
                            pcd[cl] = wq[rr].gamma; pcd[TS + cl] = wq[rr].lstar;
                            pci[cl] = wq[rr].in0; pci[TS + cl] = wq[rr].active; pci[2 * TS + cl] = wq[rr].cnt0; pci[3 * TS + cl] = wq[rr].cnt1;
                            pci[4 * TS + cl] = wq[rr].cnt2; pci[5 * TS + cl] = wq[rr].cnt3;
                        }
                    }
                }
#endif
                sync_signal(sync + 7 * W + 4, lane);
                switch (P.prio_valu) {
                    case 0: break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
        }
        // Tasks (item-major) are claimed from one LDS counter: likelihood waves from the start, MFMA-role
        // waves once their record stream is finished.  Any wave may compute any task (results go to
        // per-(item, sample) slots), so the claim order does not affect the values.
        // (task counts are made explicitly wave-uniform: the claim loop must not be compiled as a divergent loop)
        const int ntot = P.n_tasks[R - 1];
        const int* tmap = reinterpret_cast<const int*>(smem + L.tmap);
        int tstat = vwave; (void)tstat;
        int claims = 0;
        for (;;) {
#ifdef NMMA_DBG_STATIC
            int t = tstat; tstat += NVW;
#else
            // every lane issues the LDS add (lane 0 adds 1, the others 0): no divergent control flow around the claim
            int tv;
            {
                const unsigned addr = (unsigned)(uintptr_t)(lds_ip)(sync + 2 * W + 3);
                const int inc = (lane == 0) ? 1 : 0;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(tv) : "v"(addr), "v"(inc) : "memory");
            }
            int t = __builtin_amdgcn_readfirstlane(tv);
#endif
            if (dbg && blockIdx.x == 0 && lane == 0 && t < 24) { dbg[16 + t] = clock64(); dbg[40 + t] = wave; }
            if (t >= ntot) break;
            const int t_claim = t;
            if (++claims > ntot + 64) {          // cannot happen; fail loudly instead of spinning
                if (lane == 0) { g_ip wd = (g_ip)(uintptr_t)P.watchdog; wd[0] = 1; wd[1] = 900; wd[2] = (int)blockIdx.x * 64 + wave; wd[3] = t; }
                break;
            }
            int k = 0;
            if (ntot <= TMAP_MAX) {              // one LDS read instead of a serial scan over the item list
                const int e = __builtin_amdgcn_readfirstlane(tmap[t]);
                k = e >> 8; t = e & 255;
            } else {
                for (;; ++k) { const int n = __builtin_amdgcn_readfirstlane(itab[k].ntask[R - 1]); if (t < n) break; t -= n; }
            }
#ifndef NMMA_DBG_NOVALU
            if constexpr (EXT) {
                const int kind = itab[k].kind;
                if (kind == NMMA_SYS_PARAM) fast_task(std::integral_constant<int, 1>{}, k, t);
                else if (kind == NMMA_SYS_NODES) fast_task(std::integral_constant<int, 2>{}, k, t);
                else fast_task(std::integral_constant<int, 0>{}, k, t);
            } else {
                const bool two = !(itab[k].identity != 0 && itab[k].same_grid != 0);       // uniform per item
                const bool sysp = LEANX && (itab[k].kind == NMMA_SYS_PARAM || (FASTM == 5 && itab[k].kind == NMMA_SYS_NODES));
                auto run = [&](auto tb) {
                    using T = std::true_type; using F = std::false_type;
                    if constexpr (FASTM == 4) {          // an unequally spaced grid never coincides with the SVD grid: always two-stage
                        if (sysp) lean_task(tb, T{}, T{}, T{}, k, t); else lean_task(tb, T{}, F{}, T{}, k, t);
                    } else if constexpr (FASTM == 5) {
                        if (!P.st_uniform) { if (sysp) lean_gen_task(tb, T{}, T{}, T{}, k, t); else lean_gen_task(tb, T{}, F{}, T{}, k, t); }
                        else if (sysp) { if (two) lean_gen_task(tb, T{}, T{}, F{}, k, t); else lean_gen_task(tb, F{}, T{}, F{}, k, t); }
                        else { if (two) lean_gen_task(tb, T{}, F{}, F{}, k, t); else lean_gen_task(tb, F{}, F{}, F{}, k, t); }
                    } else if constexpr (FASTM == 3) {
                        if (sysp) { if (two) lean_task(tb, T{}, T{}, F{}, k, t); else lean_task(tb, F{}, T{}, F{}, k, t); }
                        else { if (two) lean_task(tb, T{}, F{}, F{}, k, t); else lean_task(tb, F{}, F{}, F{}, k, t); }
                    } else if constexpr (FASTM == 6) {   // dense: more than 16 points in every filter (host); any sample grid -- the
                        // stage-1 lerp lives in the A operands, an unequally spaced grid only changes how a datum finds its bracket
                        if (P.st_uniform) { if (sysp) lean_task(F{}, F{}, T{}, F{}, k, t); else lean_task(F{}, F{}, F{}, F{}, k, t); }
                        else { if (sysp) lean_task(F{}, F{}, T{}, T{}, k, t); else lean_task(F{}, F{}, F{}, T{}, k, t); }
                    } else {
                        if (two) lean_task(tb, T{}, F{}, F{}, k, t); else lean_task(tb, F{}, F{}, F{}, k, t);
                    }
                };
                if constexpr (FASTM == 6) run(std::false_type{});
                else if (itab[k].nf <= 16) run(std::true_type{}); else run(std::false_type{});
            }
#else
            sync_wait(sync + k, NMW); sync_signal(sync + W + 2 + k, lane);
#endif
            if (dbg && blockIdx.x == 0 && lane == 0 && t_claim < 24) dbg[104 + t_claim] = clock64();     // (after the task's done-signal)
        }
    } else {
        for (int k = 0; k < W; ++k) {
            // every likelihood wave finished its previous phase (prologue data; LDS table buffer free)
            sync_wait(sync + W + 1 + k, NVW, P.watchdog, 500 + k);
            sync_wait(sync + k, NMW, P.watchdog, 600 + k);   // coefficients of item k published
            if (dbg && blockIdx.x == 0 && vt == 0) dbg[66 + 2 * k] = clock64();
            if (NC == 10) item_phase(std::integral_constant<int, 10>{}, k);   // the reference default
            else item_phase(std::integral_constant<int, 0>{}, k);
            if (dbg && blockIdx.x == 0 && vt == 0) dbg[66 + 2 * k + 1] = clock64();
            sync_signal(sync + W + 2 + k, lane);
        }
    }
    // ---- sum over filters + floor (core/base.py:178-182)
    WalkPre wpre[WNR < 2 ? WNR : 2];
    if (vwave == 0) {            // the first likelihood wave (helpers have vwave < 0)
        for (int k = 0; k < W; ++k) sync_wait(sync + W + 2 + k, all_fast ? itab[k].ntask[R - 1] : NVW, P.watchdog, 700 + k);
        const int nb = SPLITTABLE ? P.n_bands : 1;
        bool own_totals = !SPLITTABLE || nb <= 1;       // (wave-uniform)
        if (vt < TS && tile0 + vt < B) {
            const bool isbad = always_floor != 0 || bad[vt] != 0 || sample_bad(vt) || g_wd_trip != 0;
            if (!SPLITTABLE || nb <= 1) {
                double c = 0.0, g = 0.0;             // running sums in item (= observed-filter) order
                for (int k = 0; k < W; ++k) { c += chi_tot[k * TS + vt]; g += gp_tot[k * TS + vt]; }
                double tot = c + g;
                if (isbad || !(tot - tot == 0.0)) tot = NMMA_LOGL_FLOOR;
                out[tile0 + vt] = tot;
                if constexpr (WALKF) reinterpret_cast<double*>(smem + L.stage)[vt] = tot;      // (the prologue's staging area is free by now)
            } else {
                // split launch: this workgroup owns a GROUP of one to three adjacent bands.  A band's sums sit in the slot of its
                // observed filter (lean_task) or of its last work item (lean_gen_task, FASTM 5); they are parked per observed filter
                // in the workspace [2][P.O][B] that gp_parts points at; NaN marks a bad sample
                constexpr bool SLOT_O = FASTM != 5;
                for (int k = 0; k < W; ++k) {
                    if (itab[k].ks != itab[k].nsrc - 1) continue;
                    const int o = itab[k].o;
                    const int slot = (SLOT_O ? o : k) * TS + vt;
                    gp_parts[(long)o * B + tile0 + vt] = isbad ? dnan() : chi_tot[slot];
                    gp_parts[((long)P.O + o) * B + tile0 + vt] = gp_tot[slot];
                }
            }
        }
        if (SPLITTABLE && nb > 1) {
            // The group that arrives LAST at its tile's counter adds the bands in band order -- the running sums of the fused
            // epilogue above, bit for bit -- and re-arms the counter for the next launch.  (Counters: the 64 KiB in front of the
            // workspace, zeroed when it is allocated.  Release / acquire at agent scope: the bands of a tile run on any XCD.)
            unsigned* cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(gp_parts) - SPLIT_COUNTER_BYTES) + blockIdx.x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            unsigned prev = 0;
            if (vt == 0) prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (prev == (unsigned)(nb - 1)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (vt == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (vt < TS && tile0 + vt < B) {
                    double c = 0.0, g = 0.0;
                    const int nO = P.O;
                    for (int y = 0; y < nO; ++y) {
                        c += gp_parts[(long)y * B + tile0 + vt];
                        g += gp_parts[((long)nO + y) * B + tile0 + vt];
                    }
                    double tot = c + g;
                    if (always_floor != 0 || !(tot - tot == 0.0)) tot = NMMA_LOGL_FLOOR;
                    out[tile0 + vt] = tot;
                    if constexpr (WALKF) reinterpret_cast<double*>(smem + L.stage)[vt] = tot;
                }
                own_totals = true;       // (split launch: the group that added the bands owns the tile's MCMC step)
            }
        }
        if (WALKF && own_totals) {
            // ---- the MCMC step, second phase: decide, move, propose, leave the tile's theta rows ready for the next launch
            double* totl = reinterpret_cast<double*>(smem + L.stage);
            sync_wait(sync + 7 * W + 4, 1, P.watchdog, 900);          // (the first phase finished long ago)
            const double* pl = totl + TS;
            const double* pcd = pl + 5 * WNR * 64;
            const int* pci = reinterpret_cast<const int*>(pcd + 2 * TS);
            constexpr int WPR = WNR < 2 ? WNR : 2;           // (rounds handled together, as in the first phase)
#pragma unroll
            for (int r0 = 0; r0 < WNR; r0 += WPR) {
#pragma unroll
                for (int rr = 0; rr < WPR; ++rr) {
                    const int r = r0 + rr;
                    const int e = r * 64 + vt, cl = r * WCR + vt / WT;
                    WalkPre& w = wpre[rr];
                    w.li = 0.0; w.lj = pl[e]; w.uu = pl[WNR * 64 + e]; w.vv = pl[2 * WNR * 64 + e]; w.pp = pl[3 * WNR * 64 + e]; w.th = pl[4 * WNR * 64 + e];
                    w.gamma = pcd[cl]; w.lstar = pcd[TS + cl];
                    w.in0 = pci[cl]; w.active = pci[TS + cl]; w.cnt0 = pci[2 * TS + cl]; w.cnt1 = pci[3 * TS + cl]; w.cnt2 = pci[4 * TS + cl];
                    w.cnt3 = pci[5 * TS + cl];
                }
#ifndef NMMA_DBG_WALK_NOPOST
#pragma unroll
                for (int rr = 0; rr < WPR; ++rr) {
                    const int cl = (r0 + rr) * WCR + vt / WT;
                    const long c = tile0 + cl;
                    if (c < B)
                        walk_step_post<false>(wspl, wf->ndim, WT, c, vt & (WT - 1), totl[cl], wpre[rr], wf->u, wf->v, wf->logl, wf->counts, wf->prop,
                                              const_cast<double*>(theta), wf->inside, wf->con_ops, wf->n_con_ops, !wlast);
                }
#endif
            }
#ifdef NMMA_DBG_WALK_NOPOST
            if (vt == 0) wf->counts[0] = wpre[0].cnt0 + (int)wpre[0].gamma;      // (keep the first phase alive)
#endif
        }
    }
}

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

template <int MODE, int R, int WPB, int KP>
__global__ __launch_bounds__(64 * WPB, 2) void em_fused(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const LdsOff L,
    float* __restrict__ coeff_out, double* __restrict__ tobs_out, double* __restrict__ mag_out) {
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
    const long tile0 = (long)blockIdx.x * TS;
    const int m = blockIdx.y;  // model filter
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
            if (j >= jlo && j <= jhi && jhi > jlo) {
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

// =======================================================================================
// em_lc_loglike: likelihood from SUPPLIED source-frame light curves lc[B][M][NS] (absolute
// magnitudes on the handle's sample_times, +inf / NaN where the model has no value) --
// the generic tail of the reference path for models whose light curve is produced by
// another kernel (Me2017) or by the caller (GRB afterglow, combined models):
//   combine_detector_data (model.py:381-404), sanity_check (em_likelihood.py:305-311),
//   autocomplete_data with its finite mask (utils.py:626-645), band_log_likelihood (:337-352).
// One wave per parameter vector; lanes stride over the data of each observed filter.
// HBM-bound by design (the curves are read once: B x M x NS doubles): each wave first puts ALL of its sample's curves in
// flight (coalesced, one LDS slab per wave) together with the block's copy of the sample-time and cosmology grids, so that
// the serial parts that follow -- per-sample scalars, bracket searches, finite-node walks -- run on LDS latency
// (`stage_all`; a configuration whose curves do not fit keeps the per-filter copy from global memory).
// =======================================================================================
__device__ __forceinline__ double wave_sum(double v) {
    v = group_sum(v, 64);
    // total sits in the last row; broadcast lane 63
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

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


// em_lc_loglike<G, NM, SD, SA>: the generic tail of the reference path from detector-frame curves on -- sanity_check, autocomplete_data's
// dynamic finite mask, systematics, Gaussian / truncated / upper-limit terms, floor -- for curves another kernel produced or the caller
// supplies (Me2017, combined models).  G lanes per parameter vector (64: a wave per sample; 32 / 16: two / four samples per wave), 256
// threads per workgroup; a sample's terms are added in one order whatever G (group_total_canon).  One wave of the workgroup runs the
// per-sample scalar chains while the other three stage all the workgroup's curves into LDS.  At 8192 rows every workgroup of the launch
// is resident at once, so the kernel's time is a workgroup's chain of phases (HBM-bound staging, then latency and fp64 issue), not a
// throughput: DESIGN 3.3, profiles/r04_config3_tail.md.
// NM: how the sample's curves come about -- 1: set 0 as it is; 2 / 0: the flux sum of two / of n_sets (<= 8) sets, node by node
// (stack_magnitudes, model.py:1486-1510, with its per-model gap filling: lc_stack_node) WHILE the curves are staged into LDS, so
// that a combined model's stacked set is never written to memory and read back (72.5 MB -> 48.6 MB per call at config 3's shape,
// one launch instead of two).  SD / SA: photometry / curves staged in LDS (compile-time, so that the pointers are LDS pointers to
// the compiler).  bad_rows (or NULL): rows whose sub-model delivered no light curve (floor).
#ifdef NMMA_DBG_LC_STAMPS      // measurement builds: cycle stamps of workgroup 300's four waves at the phase boundaries
__device__ unsigned long long g_lc_stamps[4 * 16];
#define LC_STAMP(i) do { if (blockIdx.x == 300 && lane == 0) g_lc_stamps[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LC_STAMP(i) do { } while (0)
#endif
// A sample's sum in ONE order whatever the grouping: virtual lane v = (index of the term) mod 64 adds its terms in increasing
// index order; a group of G lanes carries 64 / G virtual lanes per lane (a[q]: v = lane-in-group + G q); the 64 partial sums are
// added as four DPP rows of 16, then (R0 + R1) + (R2 + R3).  16, 32 or 64 lanes per sample then give the same bits, so the grouping
// can follow the batch size without a row's value depending on the size of its batch.
template <int G>
__device__ __forceinline__ double group_total_canon(const double (&a)[64 / G]) {
    double R0, R1, R2, R3;
    if constexpr (G == 16) {
        R0 = group_sum(a[0], 16); R1 = group_sum(a[1], 16); R2 = group_sum(a[2], 16); R3 = group_sum(a[3], 16);
    } else if constexpr (G == 32) {
        const double s0 = group_sum(a[0], 16), s1 = group_sum(a[1], 16);
        const int base = (int)(threadIdx.x & 32);
        R0 = __shfl(s0, base, 64); R1 = __shfl(s0, base + 16, 64); R2 = __shfl(s1, base, 64); R3 = __shfl(s1, base + 16, 64);
    } else {
        const double s0 = group_sum(a[0], 16);
        R0 = __shfl(s0, 0, 64); R1 = __shfl(s0, 16, 64); R2 = __shfl(s0, 32, 64); R3 = __shfl(s0, 48, 64);
    }
    return (R0 + R1) + (R2 + R3);
}

template <int G, int NM, bool SD, bool SA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void em_lc_loglike(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld,
    const LcSets sets, const int n_sets, const unsigned char* __restrict__ bad_rows, const int lds_per_sample,
    const int always_floor, double* __restrict__ out, double* __restrict__ chi_parts, double* __restrict__ gp_parts) {
    static_assert(G == 64 || G == 32 || G == 16, "a wave per sample, or two / four samples per wave");
    constexpr int SPW = 64 / G;                        // samples per wave
    const EmDev& P = *Pp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1), grp = lane / G;     // lane within the sample's group; the group within the wave
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long b_raw = ((long)blockIdx.x * 4 + wave) * SPW + grp;
    const long b = b_raw < B ? b_raw : B - 1;          // (a group beyond the batch recomputes the last row and stores nothing)
    const int NS = P.NS, M = P.M;
    const double* __restrict__ lc = sets.p[0];
    // block LDS: stl[NS] | dist_grid[n_cosmo] | z_grid[n_cosmo] (the cosmology grid only when it fits STAGE_COSMO nodes),
    // then per sample: app[NS] | estacc[nf_max] | praw[8] | scal[8] | curves[M][NS] (stage_all)
    const bool cosmo_lds = P.redshift_mode == NMMA_Z_GRID && P.n_cosmo <= STAGE_COSMO;
    double* stl = reinterpret_cast<double*>(smem);
    double* dgl = stl + NS;
    double* zgl = dgl + (cosmo_lds ? P.n_cosmo : 0);
    // ... | photometry {t, m, sigma, sigma_tot, ln sigma_tot}[n_data] | first work item of each datum | item descriptors (stage_dat):
    // a datum's term then runs on LDS latency -- from global memory every datum was a chain of L2 round trips, and the kernel was
    // bound by that latency at the few waves per CU its LDS slabs allow
    const int grid_bytes = ((NS + (cosmo_lds ? 2 * P.n_cosmo : 0)) * 8 + 15) & ~15;
    const int ND = P.n_data;
    double* pho = reinterpret_cast<double*>(smem + grid_bytes);
    int* ditl = reinterpret_cast<int*>(pho + 5 * ND);
    int* itl = ditl + ((ND + 3) & ~3);
    const int shared_bytes = SD ? ((grid_bytes + 5 * ND * 8 + ((ND + 3) & ~3) * 4 + P.n_items * ITEM_WORDS * 4 + 15) & ~15) : grid_bytes;
    // (SD is a compile-time switch: the staged pointers are LDS pointers to the compiler, not a run-time choice of address space)
    const double* dt_p = SD ? pho : P.dt;
    const double* dm_p = SD ? pho + ND : P.dm;
    const double* dsig_p = SD ? pho + 2 * ND : P.dsig;
    const double* dsigtot_p = SD ? pho + 3 * ND : P.dsigtot;
    const double* dlogsig_p = SD ? pho + 4 * ND : P.dlogsig;
    const int* d_item_p = SD ? ditl : P.d_item;
    const ItemDesc* item_p = SD ? reinterpret_cast<const ItemDesc*>(itl) : P.item_desc;
    double* app = reinterpret_cast<double*>(smem + shared_bytes + (size_t)(wave * SPW + grp) * lds_per_sample);
    double* estacc = app + NS;
    double* praw = estacc + P.lc_nf_max;
    double* scal = praw + 8;
    double* curves = scal + 8;
    (void)estacc; (void)app;
    const double* row = theta + b * ld;
    auto group_count = [&](const bool pred) -> int {   // lanes of THIS sample's group with pred (a ballot: no cross-lane fp64 reduction)
        unsigned long long m = __ballot(pred);
        if constexpr (G < 64) m = (m >> (grp * G)) & ((1ull << G) - 1ull);
        return __popcll(m);
    };
    // (SA -- the sample's curves staged in LDS -- is a compile-time switch like SD: `cur` below is then an LDS pointer to the compiler
    //  and a node costs a ds_read; as a run-time choice between LDS and global memory it was a flat load, several times the latency,
    //  on the serial bracket / finite-node walks of every datum)
    // (two sets: the flux-sum table of lc_stack_node, complete before any curve is stacked.  It is only read while the curves are
    //  staged, so it borrows the LDS of the staged photometry, which is filled afterwards -- a table of its own behind the slabs took
    //  the fourth workgroup per CU away at config 3's shape; behind the last sample's slab when the photometry region is too small)
    // (the cosmology grid is read by one wave only -- the block's scalar chains below -- which stages it itself: no block barrier between)
    // Which wave: the workgroups that share a CU should pick different ones (wave i sits on SIMD i, and the SIMD's issue slots are
    // what the chains cost).  Workgroups go round-robin over the 8 XCDs; k = blockIdx / 8 counts within the XCD, and whether the
    // XCD's 32 CUs are then filled round-robin (k, k + 32, ... share a CU) or one after the other (4c .. 4c + 3), (k + k / 32) & 3
    // differs among the workgroups of a CU.
    const int kx = blockIdx.x >> 3;
    const int pro_wave = (kx + (kx >> 5)) & 3;
    LC_STAMP(0);
    // That wave first copies what its chains read from memory into LDS -- the cosmology grid, the theta rows of the block's samples
    // (into each sample's own slab) -- so that the chains themselves wait for LDS and scalar loads only and run WHILE the curve loads
    // issued before them are in flight: all the workgroups of a launch of 8192 rows are resident at once and move in phase, so the
    // chip was either loading curves (HBM-bound) or running the chains (latency-bound), never both.
    constexpr int SPB = 4 * SPW;
    const bool th_lds = NS + P.lc_nf_max >= P.D;
    unsigned char bad_s = 0;
    if (wave == pro_wave) {
        if (cosmo_lds)
            for (int j = lane; j < P.n_cosmo; j += 64) { dgl[j] = P.dist_grid[j]; zgl[j] = P.z_grid[j]; }
        if (th_lds)
            for (int idx = lane; idx < SPB * P.D; idx += 64) {
                const int sx = idx / P.D, c = idx - sx * P.D;
                const long bx = (long)blockIdx.x * SPB + sx;
                reinterpret_cast<double*>(smem + shared_bytes + (size_t)sx * lds_per_sample)[c] = theta[(bx < B ? bx : B - 1) * ld + c];
            }
        if (lane < SPB && bad_rows != nullptr) {
            const long bx = (long)blockIdx.x * SPB + lane;
            bad_s = bad_rows[bx < B ? bx : B - 1];
        }
    }
    // The per-sample scalar chains (sample_scalars: conversions, z(d_L), distance modulus -- several hundred dependent instructions
    // with ONE useful lane per sample) of ALL the block's samples run side by side on the first lanes of that one wave.
    auto scalar_chains = [&]() {
        if (wave != pro_wave) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (the LDS copies above, by other lanes of this wave)
        __builtin_amdgcn_wave_barrier();
        if (lane >= SPB) return;
        const long bs_raw = (long)blockIdx.x * SPB + lane;
        const long bs = bs_raw < B ? bs_raw : B - 1;
        double* slab_s = reinterpret_cast<double*>(smem + shared_bytes + (size_t)lane * lds_per_sample);
        double* praw_s = slab_s + NS + P.lc_nf_max;
        double* scal_s = praw_s + 8;
        auto chains = [&](const double* row_s) {
#ifdef NMMA_DBG_LC_NOPRO      // (measurement builds: what the per-sample scalar chain costs)
            scal_s[S_ZP1] = 1.0; scal_s[S_IZP1] = 1.0; scal_s[S_TS] = row_s[0] * 1e-30; scal_s[S_DMOD] = 0.0; scal_s[S_RC] = 0.0; scal_s[S_EBV] = 0.0; scal_s[S_BAD] = 0.0;
#else
            double chk;
            if (cosmo_lds) sample_scalars(P, row_s, praw_s, scal_s, chk, dgl, zgl);
            else sample_scalars(P, row_s, praw_s, scal_s, chk);
            for (int o = 0; o < P.O; ++o)         // (sampled time nodes may be non-finite: autocomplete_data masks them)
                if (P.sys_kind[o] != NMMA_SYS_NODES)
                    for (int q = P.sys_off[o]; q < P.sys_off[o + 1]; ++q) chk += apply_slot(P.sys_slots[q], row_s);
            scal_s[S_BAD] = (chk - chk == 0.0) ? 0.0 : 1.0;
            if (bad_s != 0) scal_s[S_BAD] = 1.0;
#endif
        };
        if (th_lds) chains(slab_s);            // (two inlined copies: the row is an LDS pointer to the compiler in this one)
        else chains(theta + bs * ld);
    };
    const bool tab_in_pho = SD && shared_bytes - grid_bytes >= STACK2_LDS_BYTES;
    double* tab2 = tab_in_pho ? pho : reinterpret_cast<double*>(smem + shared_bytes + (size_t)(4 * SPW) * lds_per_sample);
    if constexpr (SA && NM == 2) {
        stack2_stage(tab2, threadIdx.x, 256);
        __syncthreads();
    }
    (void)tab2; (void)tab_in_pho;
    if constexpr (SA) {
        // ... and the other three waves meanwhile stage the curves of ALL the block's samples (stacking two sets on the way), every
        // load of a trip issued before the first is used.  Work item n = sample * NNP + node, NNP = the nodes of a sample rounded up
        // to whole waves; the three waves take items 192 apart.
        if (wave == pro_wave) {
            scalar_chains();
        } else {
            const int NN = M * NS, NNP = (NN + 63) & ~63, n_items = SPB * NNP;
            const float inv_nnp = 1.0f / (float)NNP;
            const int sl = ((((wave - pro_wave) & 3) - 1) << 6) + lane;
            constexpr int KM = NM > 0 ? NM : 8;
            const int n_models = NM > 0 ? NM : n_sets;
            constexpr int NPT = NM == 0 ? 2 : 8;       // items per lane and trip (12 or 16 -- more bytes in flight -- measured slower: 38.4 against 35.4 us)
            const int curves_off = NS + P.lc_nf_max + 16;
            const long blk_base = (long)blockIdx.x * SPB * NN;
            const long left = B - (long)blockIdx.x * SPB;
            const int n_own = left < SPB ? (int)left : SPB;            // samples of this block inside the batch (>= 1)
            for (int n0 = sl; n0 < n_items; n0 += NPT * 192) {
                double v[NPT][KM];
                unsigned slow = 0u;
                (void)slow;
                // (addresses as a uniform base per set -- the block's first sample -- plus a 32-bit offset: ONE register per item; as
                //  64-bit addresses per item and set they took the registers the loads in flight need)
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const int n = n0 + i * 192;
                    const int sx = (int)(((float)n + 0.5f) * inv_nnp), j = n - sx * NNP;      // (exact: n < 2^20)
                    const int sxe = sx < n_own ? sx : n_own - 1;           // (a sample beyond the batch re-reads the last row)
                    const unsigned off = (unsigned)(sxe * NN + j);
                    const bool ok = n < n_items && j < NN;
#pragma unroll
                    for (int k = 0; k < KM; ++k) v[i][k] = (ok && k < n_models) ? (sets.p[k] + blk_base)[off] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const int n = n0 + i * 192;
                    const int sx = (int)(((float)n + 0.5f) * inv_nnp), j = n - sx * NNP;
                    const long bx = (long)blockIdx.x * SPB + sx;
                    const long g = (bx < B ? bx : B - 1) * NN + j;
                    if (n < n_items && j < NN) {
                        double* dst = reinterpret_cast<double*>(smem + shared_bytes + (size_t)sx * lds_per_sample) + curves_off + j;
                        if constexpr (NM == 1) *dst = v[i][0];
                        else if constexpr (NM == 2) {
                            double r;
                            if (stack2_fast(v[i][0], v[i][1], tab2, r)) *dst = r;
                            else slow |= 1u << i;
                        } else *dst = lc_stack_node<KM>(P, sets, n_models, g, v[i]);
                    }
                }
                if constexpr (NM == 2) {               // (the few nodes with a gap to fill: one copy of the general code, values re-read)
#pragma nounroll
                    for (int i = 0; (slow >> i) != 0u; ++i) {
                        if (((slow >> i) & 1u) == 0u) continue;
                        const int n = n0 + i * 192;
                        const int sx = (int)(((float)n + 0.5f) * inv_nnp), j = n - sx * NNP;
                        const long bx = (long)blockIdx.x * SPB + sx;
                        const long g = (bx < B ? bx : B - 1) * NN + j;
                        const double vv[2] = {sets.p[0][g], sets.p[1][g]};
                        reinterpret_cast<double*>(smem + shared_bytes + (size_t)sx * lds_per_sample)[curves_off + j] = lc_stack_node<2>(P, sets, 2, g, vv, tab2);
                    }
                }
            }
        }
    } else {
        scalar_chains();
    }
    LC_STAMP(1);
    for (int j = threadIdx.x; j < NS; j += 256) stl[j] = P.st[j];
    if constexpr (SA && NM == 2) {
        if (tab_in_pho) __syncthreads();               // every wave is done with the table before the photometry overwrites it
    }
    if constexpr (SD) {
        for (int j = threadIdx.x; j < ND; j += 256) {
            pho[j] = P.dt[j]; pho[ND + j] = P.dm[j]; pho[2 * ND + j] = P.dsig[j]; pho[3 * ND + j] = P.dsigtot[j]; pho[4 * ND + j] = P.dlogsig[j];
            ditl[j] = P.d_item[j];
        }
        const int* src = reinterpret_cast<const int*>(P.item_desc);
        for (int j = threadIdx.x; j < P.n_items * ITEM_WORDS; j += 256) itl[j] = src[j];
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    LC_STAMP(2);
    const double zp1 = scal[S_ZP1], tsh = scal[S_TS], dmod = scal[S_DMOD], rc = scal[S_RC], ebv = scal[S_EBV], izp1 = scal[S_IZP1];
    bool bad = always_floor != 0 || scal[S_BAD] != 0.0;
    const bool st_uniform = P.st_uniform != 0;
    const double st0 = P.st0, st_inv_dt = P.st_inv_dt;

    // sanity_check over ALL model filters: fewer than 2 finite magnitudes -> all-inf -> floor
    // (four filters per trip, their reads issued before the first ballot: filter by filter the loop was a chain of LDS round trips,
    //  2 of the 35 us at config 3's shape)
#ifdef NMMA_DBG_LC_NOSANITY
    for (int m0 = 0; m0 < 0; m0 += 4) {
#else
    for (int m0 = 0; m0 < M; m0 += 4) {
#endif
        int nfin[4] = {0, 0, 0, 0};                    // (a ballot per G nodes: no cross-lane fp64 reduction for a count)
        for (int j0 = 0; j0 < NS; j0 += G) {
            const int j = j0 + gl;
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + q;
                const double* cur = SA ? curves + m * NS : lc + ((size_t)b * M + m) * NS;
                v[q] = HUGE_VAL;
                if (m < M && j < NS) v[q] = cur[j];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) nfin[q] += group_count(v[q] - v[q] == 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (m0 + q < M && nfin[q] < 2) bad = true;
    }

    LC_STAMP(3);
    // One datum: interpolate every source curve of the datum's band at its epoch, average, likelihood term.
    // k0 = first work item of the band (its sources are consecutive items); the bracket depends on the epoch only.
    auto datum_term = [&](const int di, const int k0, double& chi, double& gp) {
        const ItemDesc& it0 = item_p[k0];
        const int o = it0.o, nsrc = it0.nsrc, kind = it0.kind;
        const double lim = it0.lim, e_const = it0.e_const;
        const double t = dt_p[di];
        int lo = -1;                                   // t_obs[lo] <= t (<= t_obs[NS - 1]); -1: outside the grid
        if (t == t && NS >= 1 && t >= stl[0] * zp1 + tsh && t <= stl[NS - 1] * zp1 + tsh) {
            if (st_uniform) {
                // equally spaced sample_times: the index guess, then the exact test np.interp's bracket obeys (t_obs[lo] <= t, and
                // t_obs[lo + 1] > t unless lo is the last node) -- the guess is off by at most one node from rounding
                lo = (int)(((t - tsh) * izp1 - st0) * st_inv_dt);      // (a guess: the reciprocal's rounding is corrected below)
                lo = lo < 0 ? 0 : (lo > NS - 1 ? NS - 1 : lo);
                while (lo > 0 && stl[lo] * zp1 + tsh > t) --lo;
                while (lo < NS - 1 && stl[lo + 1] * zp1 + tsh <= t) ++lo;
            } else {
                int hi = NS - 1;
                lo = 0;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                }
                if (stl[hi] * zp1 + tsh <= t) lo = hi;      // t on the last node
            }
        }
        double acc_e = 0.0;
        for (int ks = 0; ks < nsrc; ++ks) {
            const ItemDesc& it = item_p[k0 + ks];
            const double ext = extinction_mag(P.ext_law, it.ebvc, zp1, ebv);
            // detector-frame curve of this source (model.py:390-397); non-finite stays non-finite
            const double* cur = SA ? curves + it.m * NS : lc + ((size_t)b * M + it.m) * NS;
            auto app = [&](const int j) { double v = cur[j]; if (ext != 0.0) v = v + ext; return (v + dmod) + rc; };
            // np.interp over the FINITE nodes only, left = right = +inf (utils.py:634-645)
            double est = dinf();
            if (t != t) {
                est = t;
            } else if (lo >= 0) {
                int jl = lo;                                  // nearest finite node at or left of t
                while (jl >= 0 && !(cur[jl] - cur[jl] == 0.0)) --jl;
                int jr = lo + 1;                              // nearest finite node right of t
                while (jr < NS && !(cur[jr] - cur[jr] == 0.0)) ++jr;
                if (jl >= 0) {
                    const double x0 = stl[jl] * zp1 + tsh;
                    if (x0 == t) est = app(jl);
                    else if (jr < NS) est = lerp_np(t, x0, stl[jr] * zp1 + tsh, app(jl), app(jr));
                }
            }
            acc_e = ks == 0 ? est : acc_e + est;     // averaged band: (a + b [+ c]) / n  (utils.py:566-584)
        }
        const double est = nsrc > 1 ? acc_e / (double)nsrc : acc_e;
        const double sd = dsig_p[di];
        double e = e_const, sig, lsig;
        if (kind == NMMA_SYS_CONST) {
            sig = dsigtot_p[di]; lsig = dlogsig_p[di];
        } else {
            const nmma_slot* sv = P.sys_slots + P.sys_off[o];
            if (kind == NMMA_SYS_PARAM) {
                e = apply_slot(sv[0], row);
            } else {
                const int K = P.sys_nn[o];
                const int ni = P.sys_nidx[di];
                if (ni < 0) e = apply_slot(sv[0], row);
                else if (ni >= K - 1) e = apply_slot(sv[K - 1], row);
                else {
                    const double v0 = apply_slot(sv[ni], row), v1 = apply_slot(sv[ni + 1], row);
                    e = ((v1 - v0) / P.sys_ndx[di]) * P.sys_noff[di] + v0;
                }
                if (!(e - e == 0.0)) e = masked_nodes_at(sv, P.sys_node_t + P.sys_off[o], K, P.dt[di], row);      // (finite mask on the nodes)
            }
            sig = sqrt(sd * sd + e * e);
            lsig = log(sig);
        }
        const double mobs = dm_p[di];
        if (sig - sig == 0.0) chi += detection_term(mobs, est, sig, lsig, lim);
        else gp += upper_limit_term(mobs, est, e);
    };

    double chi_tot = 0.0, gp_tot = 0.0;
    if (chi_parts == nullptr) {
        // all photometry points of all bands in one pass over the lanes (a band with a dozen points would otherwise leave
        // most of the wave idle for a whole pass): d_item[di] = first work item of the datum's band
        constexpr int NA = 64 / G;                     // virtual lanes per lane (group_total_canon)
        double chi_a[NA], gp_a[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) { chi_a[q] = 0.0; gp_a[q] = 0.0; }
        auto add_chi = [&](const int h, const double c) {            // (h: the trip's index mod NA -- uniform over the wave)
            if constexpr (NA == 1) chi_a[0] += c;
            else {
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h == q) chi_a[q] += c;
            }
        };
        auto add_gp = [&](const int h, const double g) {
            if constexpr (NA == 1) gp_a[0] += g;
            else {
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h == q) gp_a[q] += g;
            }
        };
        auto general_term = [&](const int h, const int di) {         // one datum through the general term, into virtual lane h
            double c = 0.0, g = 0.0;
            datum_term(di, d_item_p[di], c, g);
            add_chi(h, c);
            add_gp(h, g);
        };
        // FAST LANE (compile-time staged data only).  The common datum -- a detection inside the model window whose band has one
        // source, a constant systematic, no finite limit, no extinction, and whose two bracket nodes are finite -- needs none of
        // the general term's machinery (source loop, finite-node walks, systematics kinds, truncation mass, upper limits: ~3 000
        // instructions of code of which a lane executes a few hundred).  It is evaluated here with the SAME operations in the same
        // order (lerp_np's quotient, the residual's quotient, scipy's expression of the Gaussian term), so a lane's sum is bit for
        // bit the general term's; lanes that do not qualify -- an upper limit, a non-finite node, an epoch outside the window --
        // take the general term, and the branch is skipped when no lane of the wave needs it.
        constexpr bool FASTLANE = SD && SA;
        const bool fast_cfg = FASTLANE && !P.has_ebv && NS >= 2;
        const double t_first = stl[0] * zp1 + tsh, t_last = stl[NS - 1] * zp1 + tsh;
        int* glist = reinterpret_cast<int*>(app);      // (the sample's first NS + lc_nf_max doubles of LDS: not used otherwise)
        const int gcap = 2 * (NS + P.lc_nf_max);
        int n_gen = 0;
        // ONE loop, two passes, so that the general term (~3 000 instructions of code) is inlined ONCE: pass 0 walks the data (fast lane;
        // what it turns away is queued, or -- queue full, or no fast lane for this configuration -- evaluated in place), pass 1 the
        // queue.  (Three inlined copies -- in place, overflow, queue -- cost 1 us of instruction fetch at config 3's shape.)
        const int cap_eff = fast_cfg ? gcap : 0;
#ifdef NMMA_DBG_LC_NODATA
        const int nd_eff = ND > 100000 ? ND : 0;
#else
        const int nd_eff = ND;
#endif
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                LC_STAMP(4);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the queue: written by other lanes of this wave)
                __builtin_amdgcn_wave_barrier();
#ifdef NMMA_DBG_LC_NOGENERAL
                n_gen = 0;
#endif
            }
            const int n_pass = pass == 0 ? nd_eff : n_gen;
            for (int d0 = 0; d0 < n_pass; d0 += G) {
                const int h = (d0 / G) & (NA - 1);
                int di = d0 + gl;
                bool general = di < n_pass;
                if (pass == 1) {
                    di = general ? glist[di] : 0;      // (the queue is in increasing datum order for every grouping)
                } else if (fast_cfg && general) {
                    const ItemDesc& it0 = item_p[d_item_p[di]];
                    const double t = dt_p[di], sig = dsigtot_p[di];
                    if (it0.nsrc == 1 && it0.kind == NMMA_SYS_CONST && it0.lim == dinf() && t >= t_first && t < t_last && sig > 0.0 && sig < dinf()) {
                        int lo;
                        if (st_uniform) {
                            lo = (int)(((t - tsh) * izp1 - st0) * st_inv_dt);
                            lo = lo < 0 ? 0 : (lo > NS - 2 ? NS - 2 : lo);
                            while (lo > 0 && stl[lo] * zp1 + tsh > t) --lo;
                            while (lo < NS - 2 && stl[lo + 1] * zp1 + tsh <= t) ++lo;
                        } else {
                            int hi = NS - 1;
                            lo = 0;
                            while (hi - lo > 1) {
                                const int mid = (lo + hi) >> 1;
                                if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                            }
                        }
                        const double* cur = curves + it0.m * NS;
                        const double y0 = cur[lo], y1 = cur[lo + 1];
                        if ((y0 - y0 == 0.0) && (y1 - y1 == 0.0)) {
                            const double x0 = stl[lo] * zp1 + tsh, x1 = stl[lo + 1] * zp1 + tsh;
                            const double est = (x0 == t) ? (y0 + dmod) + rc : lerp_np(t, x0, x1, (y0 + dmod) + rc, (y1 + dmod) + rc);
                            const double x = (dm_p[di] - est) / sig;
                            if (est < dinf() && x == x) {
                                add_chi(h, ((-(x * x) / 2.0 - kNormPdfLogC) - 0.0) - dlogsig_p[di]);
                                general = false;
                            }
                        }
                    }
                }
                // The data the fast lane turned away (upper limits, epochs outside the window, non-finite nodes: a few per sample) are
                // queued in the sample's LDS and take the general term densely packed in pass 1: called in place, one such lane made
                // its whole wave walk the general term in every trip.  (The fast lane is bound by the fp64 issue rate of the SIMD --
                // four waves x ~150 instructions a trip -- not by the latency of a trip: two data per lane and trip in a branch-free
                // form, more instructions for shorter chains, took 38.7 instead of 35.1 us at config 3's shape.  A queue of one datum
                // per sample -- the single upper limit of config 3's data set -- costs 5 of the 35 us: the latency of one chain
                // through the general term and log_ndtr; handing the queues of all the block's samples to ONE wave, a lane per sample,
                // did not change the time.)
                bool in_place = general;
                if (pass == 0) {
                    unsigned long long gm = __ballot(general);
                    if constexpr (G < 64) gm = (gm >> (grp * G)) & ((1ull << G) - 1ull);
                    const int slot = n_gen + __popcll(gm & ((1ull << gl) - 1ull));
                    if (general && slot < cap_eff) { glist[slot] = di; in_place = false; }
                    n_gen += __popcll(gm);
                    n_gen = n_gen > cap_eff ? cap_eff : n_gen;
                }
                if (in_place) general_term(h, di);
            }
        }
        LC_STAMP(5);
        chi_tot = group_total_canon<G>(chi_a);
        gp_tot = group_total_canon<G>(gp_a);
    } else {
        // per-filter parts requested: one pass per band
        for (int k = 0; k < P.n_items; ++k) {
            const ItemDesc& it = item_p[k];
            if (it.ks != 0) continue;
            constexpr int NA = 64 / G;
            double chi_a[NA], gp_a[NA];
#pragma unroll
            for (int q = 0; q < NA; ++q) { chi_a[q] = 0.0; gp_a[q] = 0.0; }
            for (int d0 = 0; d0 < it.nf; d0 += G) {
                double c = 0.0, g = 0.0;
                if (d0 + gl < it.nf) datum_term(it.d0 + d0 + gl, k, c, g);
                const int h = (d0 / G) & (NA - 1);
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h == q) { chi_a[q] += c; gp_a[q] += g; }
            }
            const double chi = group_total_canon<G>(chi_a);
            const double gp = group_total_canon<G>(gp_a);
            chi_tot += chi;
            gp_tot += gp;
            if (gl == 0 && b_raw < B) {
                chi_parts[(long)it.o * B + b] = chi;
                gp_parts[(long)it.o * B + b] = gp;
            }
        }
    }
    LC_STAMP(6);
    if (gl == 0 && b_raw < B) {
        double tot = chi_tot + gp_tot;
        if (bad || !(tot - tot == 0.0)) tot = NMMA_LOGL_FLOOR;      // (a NaN term of any band makes the total NaN)
        out[b] = tot;
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

__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
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
