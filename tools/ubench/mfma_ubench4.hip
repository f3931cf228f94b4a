// Micro-benchmark 4: the "lane = sample" MLP scheme built from v_mfma_f32_4x4x1_16B_f32.
//
// One wave carries 64 samples (lane = sample).  A group = 4 hidden units:
//   layer 1:  h[r] (4 VGPRs, r = hidden unit) = sum_k W1[r,k] x_k + b1[r]  -> NP+1 MFMAs (K = 1 each), the A operand
//             of every MFMA is ONE 4-lane block of a weight VGPR broadcast to all 16 blocks (cbsz = 4, abid = block);
//   relu:     4 v_max;
//   layer 2:  acc[c][r] (3 x 4 VGPRs, coefficient 4c+r) += W2[4c+r, j] * h[j]  -> 12 MFMAs.
// Weights: 16 blocks (4 W1 + 12 W2) = 1 dword per lane per group, bias blocks of 16 groups = 1 more dword per 16 groups.
// Prints cycles per group (ideal: 17 MFMAs x 8 cycles = 136) and checks the numerics against the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

#define MFMA(a, b, c, blk) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, (blk), 0)

__device__ __forceinline__ float relu(float x) { const int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); }

// One superblock = 16 groups; record = 17 dwords per lane: [w(16 groups) | bias blocks].
struct Rec { f32x4 w[4]; float b; };

__device__ __forceinline__ Rec load_rec(const __amdgpu_buffer_rsrc_t rsrc, const int voff, const int soff) {
    Rec r;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        r.w[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + q * 1024, 0));
    r.b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (voff >> 2), soff + 4096, 0));
    return r;
}

__global__ __launch_bounds__(1024) void k_mlp(const float* __restrict__ wrec, const float* __restrict__ xin, float* out,
                                               long long* cyc, const int n_super, const int check) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wrec, 0, n_super * 17 * 256, 0x00020000);
    float x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = xin[k * 64 + lane];
    const float ones = 1.0f;
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 hp = {0, 0, 0, 0};      // relu'd hidden values of the previous group (zero for the first: contributes nothing)
    float wprev = 0.f;            // its W2 blocks
    const long long t0 = clock64();
    Rec cur = load_rec(rsrc, lane * 16, 0);
    for (int s = 0; s < n_super; ++s) {
        const int nxt = (s + 1 < n_super) ? (s + 1) : s;
        Rec nx = load_rec(rsrc, lane * 16, nxt * 17 * 256);
#define GROUP(G)                                                                                   \
        {                                                                                          \
            const float w = cur.w[(G) >> 2][(G) & 3];                                              \
            f32x4 h = {0.f, 0.f, 0.f, 0.f};                                                        \
            h = MFMA(cur.b, ones, h, (G));                                                         \
            acc[0] = MFMA(wprev, hp[0], acc[0], 4);                                                \
            acc[1] = MFMA(wprev, hp[0], acc[1], 5);                                                \
            acc[2] = MFMA(wprev, hp[0], acc[2], 6);                                                \
            h = MFMA(w, x[0], h, 0);                                                               \
            acc[0] = MFMA(wprev, hp[1], acc[0], 7);                                                \
            acc[1] = MFMA(wprev, hp[1], acc[1], 8);                                                \
            acc[2] = MFMA(wprev, hp[1], acc[2], 9);                                                \
            h = MFMA(w, x[1], h, 1);                                                               \
            acc[0] = MFMA(wprev, hp[2], acc[0], 10);                                               \
            acc[1] = MFMA(wprev, hp[2], acc[1], 11);                                               \
            acc[2] = MFMA(wprev, hp[2], acc[2], 12);                                               \
            h = MFMA(w, x[2], h, 2);                                                               \
            acc[0] = MFMA(wprev, hp[3], acc[0], 13);                                               \
            acc[1] = MFMA(wprev, hp[3], acc[1], 14);                                               \
            acc[2] = MFMA(wprev, hp[3], acc[2], 15);                                               \
            h = MFMA(w, x[3], h, 3);                                                               \
            hp[0] = relu(h[0]); hp[1] = relu(h[1]); hp[2] = relu(h[2]); hp[3] = relu(h[3]);        \
            wprev = w;                                                                             \
        }
        GROUP(0) GROUP(1) GROUP(2) GROUP(3) GROUP(4) GROUP(5) GROUP(6) GROUP(7)
        GROUP(8) GROUP(9) GROUP(10) GROUP(11) GROUP(12) GROUP(13) GROUP(14) GROUP(15)
        cur = nx;
    }
    // drain: layer 2 of the last group
    acc[0] = MFMA(wprev, hp[0], acc[0], 4);  acc[1] = MFMA(wprev, hp[0], acc[1], 5);  acc[2] = MFMA(wprev, hp[0], acc[2], 6);
    acc[0] = MFMA(wprev, hp[1], acc[0], 7);  acc[1] = MFMA(wprev, hp[1], acc[1], 8);  acc[2] = MFMA(wprev, hp[1], acc[2], 9);
    acc[0] = MFMA(wprev, hp[2], acc[0], 10); acc[1] = MFMA(wprev, hp[2], acc[1], 11); acc[2] = MFMA(wprev, hp[2], acc[2], 12);
    acc[0] = MFMA(wprev, hp[3], acc[0], 13); acc[1] = MFMA(wprev, hp[3], acc[1], 14); acc[2] = MFMA(wprev, hp[3], acc[2], 15);
    const long long t1 = clock64();
    if (check) {
        if (blockIdx.x == 0 && wave == 0)
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 4; ++r) out[(c * 4 + r) * 64 + lane] = acc[c][r];
    } else {
        float s = 0;
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    const int n_super = 32;              // 512 groups = 2048 hidden units
    const int NH = n_super * 64, NP = 4, NC = 12;
    std::vector<float> W1(NH * NP), b1(NH), W2(NC * NH), x(NP * 64);
    srand(3);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    for (auto& v : W1) v = rnd();
    for (auto& v : b1) v = 0.3f * rnd();
    for (auto& v : W2) v = rnd() * 0.05f;
    for (auto& v : x) v = rnd();
    // pack records: per superblock s: 16 groups x (64 lanes: block b = lane/4, row i = lane%4) then bias dword
    std::vector<float> rec((size_t)n_super * 17 * 64, 0.f);
    for (int s = 0; s < n_super; ++s) {
        for (int g = 0; g < 16; ++g) {
            const int h0 = (s * 16 + g) * 4;
            for (int lane = 0; lane < 64; ++lane) {
                const int b = lane / 4, i = lane % 4;
                float v;
                if (b < 4) v = W1[(h0 + i) * NP + b];                     // A = W1[hid i, param b]
                else { const int j = (b - 4) / 3, c = (b - 4) % 3; v = W2[(4 * c + i) * NH + h0 + j]; }   // A = W2[coef 4c+i, hid j]
                // dwordx4 load q = g>>2 gives lane 4 consecutive floats e = g&3: offset (q*256 + lane*4 + e)
                rec[(size_t)s * 17 * 64 + (g >> 2) * 256 + lane * 4 + (g & 3)] = v;
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane / 4, i = lane % 4;                        // bias block g = bias of group g
            rec[(size_t)s * 17 * 64 + 1024 + lane] = b1[(s * 16 + g) * 4 + i];
        }
    }
    float *d_rec, *d_x, *d_out; long long* d_cyc;
    hipMalloc(&d_rec, rec.size() * 4); hipMalloc(&d_x, x.size() * 4); hipMalloc(&d_out, 256 * 1024 * 4); hipMalloc(&d_cyc, 64);
    hipMemcpy(d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    // numerics
    hipLaunchKernelGGL(k_mlp, dim3(1), dim3(64), 0, 0, d_rec, d_x, d_out, d_cyc, n_super, 1);
    std::vector<float> got(12 * 64);
    hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int smp = 0; smp < 64; ++smp)
        for (int c = 0; c < NC; ++c) {
            double a = 0;
            for (int h = 0; h < NH; ++h) {
                double t = b1[h];
                for (int k = 0; k < NP; ++k) t += (double)W1[h * NP + k] * x[k * 64 + smp];
                if (t > 0) a += (double)W2[c * NH + h] * t;
            }
            maxerr = fmax(maxerr, fabs(a - got[c * 64 + smp]));
        }
    printf("numerics: max abs err vs fp64 host = %.3e\n", maxerr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_mlp, dim3(256), dim3(256 * wps), 0, 0, d_rec, d_x, d_out, d_cyc, n_super, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        const double groups = (double)n_super * 16;
        const double pairs = groups * 256 * wps * 4 * 256;             // hidden-sample pairs on the chip
        const double alg_flops = pairs * (2.0 * 4 + 2.0 * 10);
        printf("waves/SIMD=%d: %.1f us, %.1f cycles/group/wave (ideal 136), algorithmic %.1f TF/s\n", wps, ms * 1e3,
               (double)cyc / groups, alg_flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
