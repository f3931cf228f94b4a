// Micro-benchmark 3 (round 2): does f64 VALU work of OTHER waves overlap with a back-to-back MFMA stream, and does the
// answer depend on the MFMA's data type?  f32 / f64 MFMA peaks equal the vector peaks on gfx950 (157.3 / 78.6 TF/s), which
// suggests shared multipliers; bf16 MFMA (2.5 PF/s) has its own.  nm MFMA waves + nv filler waves (independent
// v_fma_f64 chains) per workgroup, one workgroup per CU.  Two runs per variant: the filler outlasts the MFMA stream
// (-> MFMA rate under load) and the MFMA stream outlasts the filler (-> filler rate under load).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

#define FMA64(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c))
#define FMA32(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(bf), "v"(cf))

template <int KIND>   // 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16
__device__ __forceinline__ void mfma(f32x4& acc, float x, float y, const bf16x8& p, const bf16x8& q) {
    if constexpr (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(p), "v"(q));
}

template <int KIND, int FILL>   // FILL 0: fma_f64, 1: fma_f32
__global__ __launch_bounds__(1024) void k_cross(float* out, long long* stamps, int nm, int mfma_iters, int fill_iters, int prio_fill) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long t0 = 0, t1 = 0;
    float res = 0;
    const bool is_mfma = wave < nm;
    if (is_mfma) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
        bf16x8 p, q;
        for (int j = 0; j < 8; ++j) { p[j] = (__bf16)(0.01f * (lane + j)); q[j] = (__bf16)(1.0f + 0.002f * j); }
        t0 = clock64();
        for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { mfma<KIND>(a0, x, y, p, q); mfma<KIND>(a1, x, y, p, q); mfma<KIND>(a2, x, y, p, q); mfma<KIND>(a3, x, y, p, q); }
        }
        t1 = clock64();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        f32x4 s = a0 + a1 + a2 + a3;
        res = s[0] + s[1] + s[2] + s[3];
    } else {
        if (prio_fill) __builtin_amdgcn_s_setprio(3);
        double a0 = lane * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 1e-9;
        float f0 = lane * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, bf = 1.0001f, cf = 1e-6f;
        t0 = clock64();
        for (int i = 0; i < fill_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (FILL == 0) { FMA64(a0); FMA64(a1); FMA64(a2); FMA64(a3); }
                else { FMA32(f0); FMA32(f1); FMA32(f2); FMA32(f3); }
            }
        }
        t1 = clock64();
        res = (float)(a0 + a1 + a2 + a3) + f0 + f1 + f2 + f3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0) {
        long long* st = stamps + ((long long)blockIdx.x * 16 + wave) * 4;
        st[0] = t1 - t0; st[2] = is_mfma ? 1 : 0;
    }
}

static float* g_out; static long long* g_st;

template <int KIND, int FILL>
void cross(int nm, int nv, int prio) {
    const int MI = 1500, threads = 64 * (nm + nv);
    auto run = [&](int mi, int fi, double& mt, double& ft) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_cross<KIND, FILL>), dim3(256), dim3(threads), 0, 0, g_out, g_st, nm, mi, fi, prio);
            hipDeviceSynchronize();
        }
        std::vector<long long> h(256 * 16 * 4);
        hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost);
        mt = 0; ft = 0; int cm = 0, cf = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < nm + nv; ++w) {
                const long long* st = &h[(b * 16 + w) * 4];
                if (st[2]) { mt += st[0]; ++cm; } else { ft += st[0]; ++cf; }
            }
        mt /= (cm ? cm : 1); ft /= (cf ? cf : 1);
    };
    const double n_mfma_per_simd = 16.0 * MI * nm / 4.0;
    double mt0, ft0, mtA, ftA, mtB, ftB;
    run(MI, 0, mt0, ft0);                          // MFMA stream alone
    run(MI, 4000, mtA, ftA);                       // filler outlasts the MFMA stream
    run(MI * 8, 100, mtB, ftB);                    // MFMA stream outlasts the filler
    const double per_mfma0 = mt0 / n_mfma_per_simd, per_mfmaA = mtA / n_mfma_per_simd;
    const double fill_ticks = ftB / (100 * 32.0);  // ticks per filler instruction per wave, under MFMA load
    const double per_mfmaB = mtB / (16.0 * MI * 8 * nm / 4.0);
    const double fill_per_slot = per_mfmaA / fill_ticks * (nv / 4.0);
    printf("%s mfma  %s filler  nm=%d nv=%d prio=%d | alone %6.2f ticks/MFMA/SIMD | under filler load %6.2f | filler %6.2f ticks/inst/wave "
           "-> %5.2f filler insts per MFMA per SIMD; extra MFMA ticks per filler inst %5.2f\n",
           KIND ? "bf16 16x16x32" : "f32  16x16x4 ", FILL ? "f32" : "f64", nm, nv, prio, per_mfma0, per_mfmaA, fill_ticks, fill_per_slot,
           fill_per_slot > 0 ? (per_mfmaA - per_mfma0) / fill_per_slot : 0.0);
    (void)per_mfmaB;
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_st, 256 * 16 * 4 * 8);
    for (int prio = 0; prio <= 1; ++prio) {
        cross<0, 0>(4, 4, prio); cross<1, 0>(4, 4, prio);
        cross<0, 0>(8, 8, prio); cross<1, 0>(8, 8, prio);
        cross<0, 1>(8, 8, prio); cross<1, 1>(8, 8, prio);
        cross<1, 0>(4, 12, prio); cross<1, 0>(8, 4, prio);
    }
    return 0;
}
