// Micro-benchmark 2 (round 2): how can VALU work run in the shadow of a back-to-back f32 MFMA stream?
//  E1/E2  cross-wave: the MFMA wave puts `s_nop K` behind every MFMA (it stops being an issue candidate while its
//         MFMA executes); a filler wave on the same SIMD issues independent v_fma_f64.  nm MFMA + nv filler waves.
//  E4/E5  in-wave: ONE instruction stream = MFMA followed by N independent VALU (f64 fma / i32 max / ds_read), 1 or 2
//         such waves per SIMD.
// Output per variant: shader ticks per MFMA per SIMD and filler instructions issued per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
#define FMA64(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c))
#define MAXI(a) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(lane))

template <int K> __device__ __forceinline__ void snop() {
    if constexpr (K > 0) {
        if constexpr (K >= 16) { asm volatile("s_nop 15"); snop<K - 16>(); }
        else asm volatile("s_nop %0" ::"n"(K - 1));
    }
}

// ---- cross-wave: MFMA wave with s_nop K behind every MFMA; filler = fma_f64 x4 chains (or prio variants)
template <int K, int FILL>   // FILL: 0 = fma_f64, 1 = mixed (fma64, max_i32, ds_read_b64)
__global__ __launch_bounds__(1024) void k_cross(float* out, long long* stamps, int nm, int nv, int mfma_iters, int fill_iters, int prio_fill) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int j = threadIdx.x; j < 2048; j += blockDim.x) lds[j] = j;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    float res = 0;
    const bool is_mfma = wave < nm;
    if (is_mfma) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
        t0 = clock64();
        for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                MFMA(a0); snop<K>(); MFMA(a1); snop<K>(); MFMA(a2); snop<K>(); MFMA(a3); snop<K>();
            }
        }
        t1 = clock64();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        f32x4 s = a0 + a1 + a2 + a3;
        res = s[0] + s[1] + s[2] + s[3];
    } else {
        if (prio_fill == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_fill == 3) __builtin_amdgcn_s_setprio(3);
        double a0 = lane * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 1e-9;
        int i0 = lane, i1 = lane + 1;
        const unsigned laddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds + lane * 8;
        t0 = clock64();
        for (int i = 0; i < fill_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (FILL == 0) { FMA64(a0); FMA64(a1); FMA64(a2); FMA64(a3); }
                else {
                    double r;
                    FMA64(a0); MAXI(i0); FMA64(a1);
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(laddr) : "memory");
                    a2 += r; MAXI(i1);
                }
            }
        }
        t1 = clock64();
        res = (float)(a0 + a1 + a2 + a3 + i0 + i1);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0) {
        long long* st = stamps + ((long long)blockIdx.x * 16 + wave) * 4;
        st[0] = t1 - t0; st[2] = is_mfma ? 1 : 0; st[3] = t0; st[1] = t1;
    }
}

// ---- in-wave: MFMA + N fillers in ONE stream
template <int N, int KIND>   // KIND 0: fma_f64, 1: max_i32, 2: half/half, 3: ds_read_b64 (one waitcnt per 4 MFMAs)
__global__ __launch_bounds__(1024) void k_inwave(float* out, long long* stamps, int iters) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int j = threadIdx.x; j < 2048; j += blockDim.x) lds[j] = j;
    __syncthreads();
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
    double d[8]; int q[8];
    for (int j = 0; j < 8; ++j) { d[j] = lane * 1e-3 + j; q[j] = lane + j; }
    double b = 1.0000001, c = 1e-9;
    const unsigned laddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds + lane * 8;
    auto fill = [&]() {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (KIND == 0) FMA64(d[j % 8]);
            else if (KIND == 1) MAXI(q[j % 8]);
            else if (KIND == 2) { if (j & 1) MAXI(q[j % 8]); else FMA64(d[j % 8]); }
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[j % 8]) : "v"(laddr), "n"(512 * (j % 8)) : "memory");
        }
    };
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            MFMA(a0); fill(); MFMA(a1); fill(); MFMA(a2); fill(); MFMA(a3); fill();
            if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const long long t1 = clock64();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    f32x4 s = a0 + a1 + a2 + a3;
    float r = s[0] + s[1] + s[2] + s[3];
    for (int j = 0; j < 8; ++j) r += (float)d[j] + q[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) { long long* st = stamps + ((long long)blockIdx.x * 16 + wave) * 4; st[0] = t1 - t0; st[2] = 1; }
}

static float* g_out; static long long* g_st;
static std::vector<long long> fetch() { std::vector<long long> h(256 * 16 * 4); hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost); return h; }

template <int K, int FILL>
void cross(int nm, int nv, int prio) {
    const int MI = 1500, threads = 64 * (nm + nv);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cross<K, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const int per_group = FILL == 0 ? 32 : 40;
    auto run = [&](int FI, double& mt, double& ft) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_cross<K, FILL>), dim3(256), dim3(threads), 96 * 1024, 0, g_out, g_st, nm, nv, MI, FI, prio);
            hipDeviceSynchronize();
        }
        auto h = fetch();
        mt = 0; ft = 0; int cm = 0, cf = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < nm + nv; ++w) {
                const long long* st = &h[(b * 16 + w) * 4];
                if (st[2]) { mt += st[0]; ++cm; } else { ft += st[0]; ++cf; }
            }
        mt /= cm; ft /= (cf ? cf : 1);
    };
    // A: the filler outlasts the MFMA stream (MFMA rate under load); B: the filler finishes first (its rate under load)
    double mtA, ftA, mtB, ftB;
    const int FI_A = (int)(3.0 * MI * 16 * (nm / 4.0) * (32 + K) / (per_group * 5.0)) + 1;
    const int FI_B = (int)(0.4 * MI * 16 * (nm / 4.0) / per_group) + 1;
    run(FI_A, mtA, ftA);
    run(FI_B, mtB, ftB);
    const double per_mfma = mtA / (MI * 16.0) / (nm / 4.0);
    const double fill_ticks = ftB / ((double)FI_B * per_group);            // per instruction per filler wave, loaded
    const double per_mfma_B = mtB / (MI * 16.0) / (nm / 4.0);
    printf("cross  s_nop=%2d fill=%d nm=%d nv=%d prio=%d | %6.2f ticks/MFMA/SIMD | filler %6.2f ticks/inst/wave -> %5.2f filler insts per MFMA slot per SIMD %s\n",
           K, FILL, nm, nv, prio, per_mfma, fill_ticks, per_mfma / fill_ticks * (nv / 4.0), ftB < mtB ? "" : "(filler B outlasted the MFMAs!)");
    (void)per_mfma_B;
    fflush(stdout);
}

template <int N, int KIND>
void inwave(int waves) {
    const int MI = 1500;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inwave<N, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_inwave<N, KIND>), dim3(256), dim3(64 * waves), 96 * 1024, 0, g_out, g_st, MI);
        hipDeviceSynchronize();
    }
    auto h = fetch();
    double mt = 0; int cm = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { mt += h[(b * 16 + w) * 4]; ++cm; }
    mt /= cm;
    printf("inwave N=%d kind=%d waves/SIMD=%d | %6.2f ticks/MFMA/SIMD  (%5.2f filler/MFMA)\n", N, KIND, waves / 4, mt / (MI * 16.0) / (waves / 4.0), (double)N);
    fflush(stdout);
}

template <int KIND> void inwave_sweep() {
    for (int w = 4; w <= 8; w += 4) {
        inwave<0, KIND>(w); inwave<1, KIND>(w); inwave<2, KIND>(w); inwave<3, KIND>(w); inwave<4, KIND>(w);
        inwave<5, KIND>(w); inwave<6, KIND>(w); inwave<8, KIND>(w);
    }
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_st, 256 * 16 * 4 * 8);
    hipMemset(g_st, 0, 256 * 16 * 4 * 8);
    const int sets[3][2] = {{4, 4}, {8, 8}, {4, 12}};
    for (int si = 0; si < 3; ++si) {
        const int nm = sets[si][0], nv = sets[si][1];
        cross<0, 0>(nm, nv, 0); cross<2, 0>(nm, nv, 0); cross<4, 0>(nm, nv, 0); cross<6, 0>(nm, nv, 0); cross<8, 0>(nm, nv, 0);
        cross<12, 0>(nm, nv, 0); cross<16, 0>(nm, nv, 0); cross<20, 0>(nm, nv, 0); cross<24, 0>(nm, nv, 0); cross<28, 0>(nm, nv, 0);
        cross<8, 0>(nm, nv, 1); cross<16, 0>(nm, nv, 1); cross<24, 0>(nm, nv, 1);
        cross<0, 1>(nm, nv, 0); cross<16, 1>(nm, nv, 0); cross<24, 1>(nm, nv, 0);
    }
    inwave_sweep<0>(); inwave_sweep<1>(); inwave_sweep<2>(); inwave_sweep<3>();
    return 0;
}
