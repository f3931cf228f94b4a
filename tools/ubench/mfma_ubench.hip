// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 from one wave per SIMD under the
// dependency patterns of the surrogate MLP loop.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int VARIANT>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
    f32x4 d = {x, y, x, y};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (VARIANT == 0) {          // 4 independent accumulators, no VALU
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        } else if (VARIANT == 1) {   // 2 accumulators alternating (the layer-2 chain)
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        } else if (VARIANT == 2) {   // MLP step: relu(VALU) feeding each MFMA's B operand
            f32x4 h;
            h[0] = __builtin_amdgcn_fmed3f(d[0], 0.f, __builtin_inff());
            h[1] = __builtin_amdgcn_fmed3f(d[1], 0.f, __builtin_inff());
            h[2] = __builtin_amdgcn_fmed3f(d[2], 0.f, __builtin_inff());
            h[3] = __builtin_amdgcn_fmed3f(d[3], 0.f, __builtin_inff());
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);     // "layer 1 of next record"
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h[1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h[2], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h[3], a1, 0, 0, 0);
        } else if (VARIANT == 3) {   // as 2 but relu hoisted: all four h computed before any MFMA (same here) + B from d directly
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, d[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, d[1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, d[2], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, d[3], a1, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    f32x4 s = a0 + a1 + a2 + a3 + d;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[VARIANT] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4 * sizeof(float)); hipMalloc(&cyc, 8 * sizeof(long long));
    const int iters = 4096;
    for (int waves = 1; waves <= 2; ++waves) {
        const int threads = 256 * waves;   // 4 or 8 waves per CU -> 1 or 2 per SIMD
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const int nm[4] = {4, 4, 5, 5};
        for (int v = 0; v < 4; ++v)
            printf("waves/SIMD=%d variant %d: %.1f cycles per MFMA (wave view)\n", waves, v, (double)h[v] / iters / nm[v]);
    }
    return 0;
}
