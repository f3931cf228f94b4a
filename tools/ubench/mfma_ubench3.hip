// Micro-benchmark 2: which feature of the MLP loop costs MFMA issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float relu_i(float x) { int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); }

template <int VARIANT>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters, const float* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, b0 = a0, b1 = a0;
    float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
    f32x4 d = {x, y, x, y};
    f32x4 wa = {x, y, y, x};
    const float* wp = w + lane * 4;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (VARIANT == 0) {          // in-place, 2 chains (reference: ~34)
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        } else if (VARIANT == 1) {   // out-of-place ping-pong, 2 chains
            b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b1, 0, 0, 0);
        } else if (VARIANT == 2) {   // A operands = 4 different registers
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0], y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1], y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2], y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3], y, a1, 0, 0, 0);
        } else if (VARIANT == 3) {   // integer relu of an OLD value before each MFMA (no fresh MFMA dependency)
            float h0 = relu_i(d[0]), h1 = relu_i(d[1]), h2 = relu_i(d[2]), h3 = relu_i(d[3]);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h1, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h2, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, h3, a1, 0, 0, 0);
            d[0] += 1.f; d[1] -= 1.f; d[2] += 2.f; d[3] -= 2.f;
        } else if (VARIANT == 4) {   // + one global load per 4 MFMAs feeding the A operands of the NEXT iteration
            const f32x4 nw = *reinterpret_cast<const f32x4*>(wp + (i & 255) * 256);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0], y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1], y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2], y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3], y, a1, 0, 0, 0);
            wa = nw;
        } else if (VARIANT == 5) {   // random-valued operands (data-dependent power)
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0], d[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1], d[1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2], d[2], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3], d[3], a1, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    f32x4 s = a0 + a1 + b0 + b1 + d + wa;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[VARIANT] = t1 - t0;
}

int main() {
    float* out; long long* cyc; float* w;
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 8 * sizeof(long long));
    hipMalloc(&w, 256 * 256 * sizeof(float) * 2);
    hipMemset(w, 0x3c, 256 * 256 * sizeof(float) * 2);
    const int iters = 16384;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, w);
                else hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, w);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[v], e0, e1);
            }
        }
        long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        // per SIMD: wps waves x iters x 4 MFMAs
        const double n = (double)wps * iters * 4;
        printf("waves/SIMD=%d: pure MFMA %.2f ns per MFMA per SIMD (%.1f TF/s) | with relu+adds %.2f ns (%.1f TF/s); ticks/MFMA/wave %.1f / %.1f\n",
               wps, ms[0] * 1e6 / n, 2048.0 * 1024 / (ms[0] * 1e6 / n) / 1e3, ms[1] * 1e6 / n, 2048.0 * 1024 / (ms[1] * 1e6 / n) / 1e3,
               (double)h[0] / iters / 4, (double)h[3] / iters / 4);
    }
    return 0;
}
