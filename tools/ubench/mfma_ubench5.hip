// Micro-benchmark 5: raw issue rate of v_mfma_f32_4x4x1_16B_f32 (vs 16x16x4) with/without broadcast and interleaved VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int V>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 a[6];
    for (int i = 0; i < 6; ++i) a[i] = f32x4{0, 0, 0, 0};
    float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
    int r0 = lane, r1 = lane * 3;
    for (int i = 0; i < iters; ++i) {
        if (V == 0) {          // 4x4x1, 6 independent chains, no broadcast
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[j], 0, 0, 0);
        } else if (V == 1) {   // with cbsz=4 broadcast
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[j], 4, 3, 0);
        } else if (V == 2) {   // 3 chains (dependency distance 3)
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j % 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[j % 3], 4, 3, 0);
        } else if (V == 3) {   // 6 chains + 1 independent VALU per 6 MFMAs
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[j], 4, 3, 0);
            r0 = max(r0 + 1, 0);
            asm volatile("" : "+v"(r0));
        } else if (V == 4) {   // 6 chains + 2 independent VALU per 6 MFMAs
#pragma unroll
            for (int j = 0; j < 3; ++j) a[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[j], 4, 3, 0);
            r0 = max(r0 + 1, 0);
            asm volatile("" : "+v"(r0));
#pragma unroll
            for (int j = 3; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[j], 4, 3, 0);
            r1 = max(r1 + 1, 0);
            asm volatile("" : "+v"(r1));
        } else if (V == 5) {   // 16x16x4 reference, 6 chains
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[j], 0, 0, 0);
        } else if (V == 6) {   // 2 chains
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j % 2] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[j % 2], 4, 3, 0);
        }
    }
    f32x4 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + r0 + r1;
}
template <int V> void run(const char* name, float* out, double flops_per_mfma) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * wps), 0, 0, out, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double n = (double)wps * iters * 6;       // MFMAs per SIMD
        printf("%-40s waves/SIMD=%d: %.2f ns per MFMA per SIMD, %.1f TF/s\n", name, wps, ms * 1e6 / n,
               flops_per_mfma * 1024 / (ms * 1e6 / n) / 1e3);
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    run<0>("4x4x1 6 chains", out, 512);
    run<1>("4x4x1 6 chains cbsz=4", out, 512);
    run<2>("4x4x1 3 chains cbsz=4", out, 512);
    run<6>("4x4x1 2 chains cbsz=4", out, 512);
    run<3>("4x4x1 6 chains + 2 VALU/6", out, 512);
    run<4>("4x4x1 6 chains + 4 VALU/6", out, 512);
    run<5>("16x16x4 6 chains", out, 2048);
    return 0;
}
