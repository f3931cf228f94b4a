// Micro-benchmark 6: what does one instruction of another wave cost the MFMA pipe of the same SIMD?
// Workgroup = 4 MFMA-only waves (one per SIMD, back-to-back v_mfma_f32_16x16x4_f32) + NF "filler" waves that loop
// over ONE instruction type.  Reported: MFMA rate with fillers, filler instruction rate, and the MFMA cycles lost
// per filler instruction (all per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int T>
__device__ __forceinline__ void filler_body(double& a, double& b, double& c, int& i0, int& i1, const double* lds, f32x4& q) {
    if (T == 0) { asm volatile("v_max_i32 %0, %0, %1" : "+v"(i0) : "v"(i1)); }
    else if (T == 1) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c)); }
    else if (T == 2) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b)); }
    else if (T == 3) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); }
    else if (T == 4) { asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(i1)); }
    else if (T == 5) { asm volatile("v_rcp_f64 %0, %0" : "+v"(a)); }
    else if (T == 6) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i0) : "v"(i1)); }
    else if (T == 7) { asm volatile("s_nop 0"); }
    else if (T == 8) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(q[0]) : "v"(q[1])); }
    else if (T == 9) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(q[1])); }
    else if (T == 10) { asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc"); }
    else if (T == 11) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i1) : "vcc"); }
    else if (T == 12) { asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(i1)); }
    else if (T == 13) { asm volatile("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %3, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(c), "v"(b)); }
}

template <int T, int PRIO>
__global__ __launch_bounds__(1024) void k(float* out, long long* cnt, const int mfma_iters, const int nf) {
    __shared__ double lds[2048];
    __shared__ int done;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) __builtin_amdgcn_s_setprio(PRIO);
    if (threadIdx.x == 0) done = 0;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 1.0 + i;
    __syncthreads();
    if (wave < 4) {
        f32x4 a[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
        const long long t0 = clock64();
        for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[j], 0, 0, 0);
        }
        const long long t1 = clock64();
        f32x4 s = a[0] + a[1] + a[2] + a[3];
        out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
        if (lane == 0) { __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                         if (blockIdx.x == 0 && wave == 0) { cnt[0] = t1 - t0; cnt[4] = t0; cnt[5] = t1; } }
    } else {
        double a = 1.0 + lane, b = 1.0000001, c = 1e-9;
        int i0 = lane, i1 = (lane * 16) & 2047;
        f32x4 q = {1.f, 1.0001f, 0.f, 0.f};
        long long n = 0;
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
            for (int u = 0; u < 64; ++u) filler_body<T>(a, b, c, i0, i1, lds, q);
            n += 64;
        }
        if (lane == 0 && blockIdx.x == 0 && wave == 4) { cnt[1] = n; cnt[2] = __hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); cnt[3] = clock64(); }
        if (a == 12345.678 && i0 == -77 && q[0] == 3.f) out[0] = (float)b;
    }
}

template <int T, int PRIO = 3> void run(const char* name, float* out, long long* cnt) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int nf = 0; nf <= 8; nf += 4) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k<T, PRIO>), dim3(256), dim3(256 + 64 * nf), 0, 0, out, cnt, iters, nf);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        long long h[6] = {0}; hipMemcpy(h, cnt, 48, hipMemcpyDeviceToHost);
        const double nm = (double)iters * 4; 
        const double ns_per_mfma = ms * 1e6 / nm;
        static double base = 0;
        if (nf == 0) { base = ns_per_mfma; printf("%-22s fillers/SIMD=0: %.2f ns/MFMA\n", name, ns_per_mfma); continue; }
        const double fill_per_simd = (double)h[1] * (nf / 4);          // instructions issued on one SIMD by its fillers
        printf("%-22s fillers/SIMD=%d: %.2f ns/MFMA (%.0f%% of MFMA rate) | %.1f filler instr per MFMA | cost %.2f ns = %.1f cycles @2.4GHz per filler instr\n",
               name, nf / 4, ns_per_mfma, 100 * base / ns_per_mfma, fill_per_simd / nm, (ns_per_mfma - base) * nm / fill_per_simd,
               (ns_per_mfma - base) * nm / fill_per_simd * 2.4);
    }
}
int main() {
    float* out; long long* cnt; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cnt, 64);
    run<0, 0>("v_max_i32 (prio 0)", out, cnt);
    run<0>("v_max_i32", out, cnt);
    run<8>("v_fma_f32", out, cnt);
    run<1>("v_fma_f64", out, cnt);
    run<13>("v_fma_f64 x2 indep", out, cnt);
    run<2>("v_mul_f64", out, cnt);
    run<3>("v_add_f64", out, cnt);
    run<5>("v_rcp_f64", out, cnt);
    run<9>("v_cvt_f64_f32", out, cnt);
    run<10>("v_cmp_lt_f64", out, cnt);
    run<11>("v_cndmask_b32", out, cnt);
    run<6>("v_mov_b32_dpp", out, cnt);
    run<4>("ds_read_b128+wait", out, cnt);
    run<12>("ds_read_b64+wait", out, cnt);
    run<7>("s_nop", out, cnt);
    return 0;
}
