// Micro-benchmark 7: the MLP record loop of em_logl's MFMA role in isolation (R = 1, KP = 1).
// Variants (template V) explore instruction order / count; prints cycles per 16-hidden-unit record for
// 1 and 2 MFMA waves per SIMD (ideal: 5 MFMAs x 32 = 160).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int RECF = 256 + 64 + 16, RECB = RECF * 4, PF = 8;
__device__ __forceinline__ float relu1(float x) { const int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); }

// V: 0 = baseline order (relu | L1 | loads | L2)   1 = no loads   2 = no relu (h = d of the record before: same distance)
//    3 = relu interleaved between the L2 MFMAs of the PREVIOUS record (deeper software pipeline)
//    4 = loads only every other record pair packed (2 x b128 + 1 x b32 -> same bytes, fewer instrs: a1 packed per 4)
template <int V, int AUX = 0, int COMPACT = 0>
__global__ __launch_bounds__(512) void k(const float* __restrict__ wrec, float* out, long long* cyc, const int nrec, const int wrec_bytes) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wrec, 0, wrec_bytes, 0x00020000);
    const int off_a2 = COMPACT ? (((lane & 15) < 10) ? ((lane >> 4) * 10 + (lane & 15)) * 16 : 0x7fffff00) : lane * 16, off_a1 = (256 + lane) * 4, off_b = (256 + 64 + (lane >> 4) * 4) * 4;
    auto ld4 = [&](int voff, int soff) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, AUX)); };
    auto ld1 = [&](int voff, int soff) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, AUX)); };
    const float xB = 0.01f * lane;
    const int nw = blockDim.x >> 6, per = 128 / nw;      // records of one item per wave
    int soff = wave * per * RECB;
    int left = per - PF;                                  // records of the current item not yet requested
    f32x4 ra2[PF], rbias[PF]; float ra1[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) { ra2[u] = ld4(off_a2, soff + u * RECB); ra1[u] = ld1(off_a1, soff + u * RECB); rbias[u] = ld4(off_b, soff + u * RECB); }
    soff += PF * RECB;
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[0], xB, rbias[0], 0, 0, 0);
    f32x4 dprev = d;
    f32x4 dcur = d;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    f32x4 h = {relu1(d[0]), relu1(d[1]), relu1(d[2]), relu1(d[3])};
    const long long t0 = clock64();
#pragma unroll 1
    for (int i0 = 0; i0 < nrec; i0 += PF) {
        if (left <= 0) { soff += (128 - per) * RECB; left = per; }
        left -= PF;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int nu = (u + 1) % PF;
            if (V == 4 || V == 5) {
                // h = relu'd hidden values of record u (h[0] ready; h[1..3] computed below from dcur)
                const f32x4 a2 = ra2[u];
                h[1] = relu1(dcur[1]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0], h[0], acc0, 0, 0, 0);
                h[2] = relu1(dcur[2]);
                const f32x4 dn = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[nu], xB, rbias[nu], 0, 0, 0);
                h[3] = relu1(dcur[3]);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[1], h[1], acc1, 0, 0, 0);
                ra2[u] = ld4(off_a2, soff);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[2], h[2], acc0, 0, 0, 0);
                ra1[u] = ld1(off_a1, soff);
                rbias[u] = ld4(off_b, soff);
                soff += RECB;
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[3], h[3], acc1, 0, 0, 0);
                dcur = dn;
                h[0] = relu1(dn[0]);
                if (V == 4) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // v_max h1
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // L2(0)
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // v_max h2
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // L1 next
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // v_max h3
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // L2(1)
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // load
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // L2(2)
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // loads
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // L2(3)
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // v_max h0 (next)
                }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if (V == 3) {
                // h (relu of record u) was produced during the previous step; compute d(next) then, interleaved with
                // this record's L2 MFMAs, the relu of d(next)
                const f32x4 dn = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[nu], xB, rbias[nu], 0, 0, 0);
                const f32x4 a2 = ra2[u];
                ra2[u] = ld4(off_a2, soff); ra1[u] = ld1(off_a1, soff); rbias[u] = ld4(off_b, soff);
                soff += RECB;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0], h[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[1], h[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[2], h[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[3], h[3], acc1, 0, 0, 0);
                f32x4 hn = {relu1(d[0]), relu1(d[1]), relu1(d[2]), relu1(d[3])};   // relu of the record computed one step ago
                h = hn; d = dn;
                // order: L1 | loads | L2 x2 | relu x2 | L2 x2 | relu x2
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            f32x4 hh;
            if (V == 2) hh = dprev; else { hh[0] = relu1(d[0]); hh[1] = relu1(d[1]); hh[2] = relu1(d[2]); hh[3] = relu1(d[3]); }
            dprev = d;
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[nu], xB, rbias[nu], 0, 0, 0);
            const f32x4 a2 = ra2[u];
            if (V != 1) { ra2[u] = ld4(off_a2, soff); ra1[u] = ld1(off_a1, soff); rbias[u] = ld4(off_b, soff); }
            soff += RECB;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0], hh[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[1], hh[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[2], hh[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[3], hh[3], acc1, 0, 0, 0);
            if (V != 2) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (V != 1) __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    const f32x4 s = acc0 + acc1 + d + h + dprev + dcur;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V, int AUX = 0, int COMPACT = 0> void run(const char* name, const float* w, float* out, long long* cyc, int wbytes, int nwg = 256) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nrec = 512;
    for (int wps = 1; wps <= 2; ++wps) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k<V, AUX, COMPACT>), dim3(nwg), dim3(256 * wps), 0, 0, w, out, cyc, nrec, wbytes);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s waves/SIMD=%d: %7.1f cycles/record/wave, %6.2f ns per record per SIMD (ideal 160 cyc = %.1f ns @2.4)\n", name, wps,
               (double)c / nrec, ms * 1e6 / (nrec * wps), 160 / 2.4);
    }
}
int main() {
    const int nrec_total = 128 * 140 + 16;
    std::vector<float> w((size_t)nrec_total * RECF);
    for (size_t i = 0; i < w.size(); ++i) w[i] = 0.001f * (float)((i * 2654435761u) % 2001) - 1.0f;
    float* d_w; float* out; long long* cyc;
    hipMalloc(&d_w, w.size() * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    const int wb = (int)(w.size() * 4);
    run<0>("baseline (relu|L1|loads|L2)", d_w, out, cyc, wb);
    run<1>("no loads", d_w, out, cyc, wb);
    run<2>("no relu", d_w, out, cyc, wb);
    run<4>("hand-interleaved + sched groups", d_w, out, cyc, wb);
    run<5>("hand-interleaved, compiler order", d_w, out, cyc, wb);
    run<0>("baseline, 1 workgroup", d_w, out, cyc, wb, 1);
    return 0;
}
