// Micro-benchmark (round 2): what does ONE instruction of a co-resident wave cost the f32 MFMA stream of its SIMD?
// A workgroup = NM "MFMA waves" (v_mfma_f32_16x16x4_f32, 4 independent accumulators) + NV "filler waves" that issue
// one instruction kind in an unrolled loop.  Every wave stamps s_memtime at start and end; the host prints, per kind,
//   ticks per MFMA (MFMA waves, filler running the whole time)  and  ticks per filler instruction,
// against the two solo baselines.  One workgroup per CU (LDS padding), 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

enum Kind { K_NONE = 0, K_FMA64, K_MUL64, K_ADD64, K_FMA32, K_MAXI32, K_MOVDPP, K_DSREAD64, K_SALU, K_FMA64_HALF, K_CMP64, K_CNDMASK,
            K_FMA64_DEP, K_RELU_DEP, K_LAST };
static const char* kname[] = {"none", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_max_i32", "v_mov_dpp", "ds_read_b64",
                              "s_add_u32", "v_fma_f64(16 lanes)", "v_cmp_f64", "v_cndmask", "v_fma_f64 dep chain", "mfma+4relu(in-wave)"};

template <int KIND>
__device__ __forceinline__ void filler(int iters, double& sink, int lane, double* lds) {
    double a0 = lane * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 1e-9;
    float f0 = lane, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fb = 1.0000001f, fc = 1e-9f;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    unsigned s0 = 0;
    const unsigned laddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds + lane * 8;
    if (KIND == K_FMA64_HALF) { asm volatile("s_mov_b64 exec, 0xffff" ::: "memory"); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == K_FMA64 || KIND == K_FMA64_HALF) {
                asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            } else if (KIND == K_FMA64_DEP) {
                asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2"
                             : "+v"(a0) : "v"(b), "v"(c));
            } else if (KIND == K_MUL64) {
                asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
            } else if (KIND == K_ADD64) {
                asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));
            } else if (KIND == K_FMA32) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));
            } else if (KIND == K_MAXI32) {
                asm volatile("v_max_i32 %0, %0, %4\n\tv_max_i32 %1, %1, %4\n\tv_max_i32 %2, %2, %4\n\tv_max_i32 %3, %3, %4"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(lane));
            } else if (KIND == K_MOVDPP) {
                asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
            } else if (KIND == K_DSREAD64) {
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(laddr) : "memory");
            } else if (KIND == K_SALU) {
                asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7" : "+s"(s0));
            } else if (KIND == K_CMP64) {
                asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cmp_lt_f64 vcc, %1, %2\n\tv_cmp_lt_f64 vcc, %2, %3\n\tv_cmp_lt_f64 vcc, %3, %0"
                             :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
            } else if (KIND == K_CNDMASK) {
                asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(lane) : "vcc");
            }
        }
    }
    if (KIND == K_FMA64_HALF) { asm volatile("s_mov_b64 exec, -1" ::: "memory"); }
    sink = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + i0 + i1 + i2 + i3 + s0;
}

// MODE 0: MFMA waves = pure MFMA.  MODE 1: MFMA waves also do 4 integer relu per 5 MFMAs on the previous result (the MLP loop shape)
template <int KIND, int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* stamps, int nm, int nv, int mfma_iters, int fill_iters, int prio_fill) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int j = threadIdx.x; j < 2048; j += blockDim.x) lds[j] = j;
    __syncthreads();
    long long t0 = 0, t1 = 0, w0 = 0, w1 = 0;
    float res = 0;
    // waves are placed on SIMDs round-robin: interleave the roles so that every SIMD gets nm/4 MFMA + nv/4 filler waves
    const bool is_mfma = wave < nm;   // consecutive waves go to SIMDs round-robin: every SIMD gets nm/4 MFMA + nv/4 filler waves
    const int per_simd = (nm + nv) / 4;   // waves per SIMD
    (void)per_simd;
    if (is_mfma && nm > 0) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, d = {1.f, -1.f, 2.f, -2.f};
        float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
        t0 = clock64(); w0 = wall_clock64();
        for (int i = 0; i < mfma_iters; ++i) {
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {     // 5 MFMAs + 4 relu, like one weight record (accounted as 4 "MFMA-equivalents" x 1.25)
                    int h0 = __builtin_bit_cast(int, d[0]), h1 = __builtin_bit_cast(int, d[1]), h2 = __builtin_bit_cast(int, d[2]), h3 = __builtin_bit_cast(int, d[3]);
                    h0 = h0 > 0 ? h0 : 0; h1 = h1 > 0 ? h1 : 0; h2 = h2 > 0 ? h2 : 0; h3 = h3 > 0 ? h3 : 0;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, f32x4{y, x, y, x}, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, __builtin_bit_cast(float, h0), a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, __builtin_bit_cast(float, h1), a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, __builtin_bit_cast(float, h2), a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, __builtin_bit_cast(float, h3), a1, 0, 0, 0);
                }
            }
        }
        t1 = clock64(); w1 = wall_clock64();
        f32x4 s = a0 + a1 + a2 + a3 + d;
        res = s[0] + s[1] + s[2] + s[3];
    } else if (!is_mfma && nv > 0) {
        if (prio_fill == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_fill == 3) __builtin_amdgcn_s_setprio(3);
        double sink = 0;
        t0 = clock64(); w0 = wall_clock64();
        filler<KIND>(fill_iters, sink, lane, lds);
        t1 = clock64(); w1 = wall_clock64();
        res = (float)sink;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0) {
        long long* st = stamps + ((long long)blockIdx.x * 16 + wave) * 4;
        st[0] = t1 - t0; st[1] = w1 - w0; st[2] = is_mfma ? 1 : 0; st[3] = t0;
    }
}

struct Res { double mfma_ticks, fill_ticks, mfma_wall, fill_wall; };

template <int KIND, int MODE>
Res run(float* out, long long* stamps, int nm, int nv, int mfma_iters, int fill_iters, int prio) {
    const int threads = 64 * (nm + nv);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, MODE>), dim3(256), dim3(threads), 96 * 1024, 0, out, stamps, nm, nv, mfma_iters, fill_iters, prio);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(256 * 16 * 4);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    Res r{0, 0, 0, 0};
    int cm = 0, cf = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < nm + nv; ++w) {
            const long long* st = &h[(b * 16 + w) * 4];
            if (st[2]) { r.mfma_ticks += st[0]; r.mfma_wall += st[1]; ++cm; } else { r.fill_ticks += st[0]; r.fill_wall += st[1]; ++cf; }
        }
    if (cm) { r.mfma_ticks /= cm; r.mfma_wall /= cm; }
    if (cf) { r.fill_ticks /= cf; r.fill_wall /= cf; }
    return r;
}

template <int KIND, int MODE>
void scenario(float* out, long long* stamps, int nm, int nv, int prio) {
    const int MI = 2000;                   // x16 MFMAs (MODE 0) or x20 (MODE 1) per MFMA wave
    const int n_mfma = MI * (MODE == 0 ? 16 : 20);
    // solo baselines
    Res m0 = run<K_NONE, MODE>(out, stamps, nm, 0, MI, 0, 0);
    // calibrate the filler so that it runs ~1.5x as long as the MFMA waves (covers them fully)
    int FI = 500;
    Res f0 = run<KIND, MODE>(out, stamps, 0, nv, 0, FI, prio);
    const double per_fill_solo = f0.fill_ticks / (FI * 32.0);
    // co-run A: filler longer than the MFMA stream -> cost per MFMA under load
    int fi_long = (int)(3.0 * m0.mfma_ticks / (per_fill_solo * 32.0)) + 1;
    Res a = run<KIND, MODE>(out, stamps, nm, nv, MI, fi_long, prio);
    // co-run B: MFMA stream longer than the filler -> cost per filler instruction under load
    int fi_short = (int)(0.3 * m0.mfma_ticks / (per_fill_solo * 32.0)) + 1;
    Res b = run<KIND, MODE>(out, stamps, nm, nv, MI, fi_short, prio);
    const double mf_solo = m0.mfma_ticks / n_mfma / (nm / 4), mf_load = a.mfma_ticks / n_mfma / (nm / 4);
    const double fl_load = b.fill_ticks / (fi_short * 32.0);
    // exchange rate: filler instructions issued per SIMD while the MFMA waves ran (co-run A) vs the MFMA slowdown
    //   filler rate under load (per SIMD) = (nv/4) / (a.fill_ticks / (fi_long*32))   [insts per tick]
    const double fill_rate = (nv / 4.0) / fl_load;
    const double lost_per_fill = (mf_load - mf_solo) / (mf_load * fill_rate);    // SIMD ticks of MFMA time lost per filler instruction
    printf("%-22s mode %d nm=%d nv=%d prio=%d | MFMA/SIMD solo %6.2f  loaded %6.2f ticks | filler solo %6.2f loaded %6.2f ticks/inst/wave | "
           "MFMA ticks lost per filler inst %6.2f | clk %.3f GHz\n",
           kname[KIND], MODE, nm, nv, prio, mf_solo, mf_load, per_fill_solo, fl_load, lost_per_fill,
           a.mfma_ticks / (a.mfma_wall * 10.0));
    fflush(stdout);
}

int main(int argc, char** argv) {
    float* out; long long* stamps;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&stamps, 256 * 16 * 4 * 8);
    hipMemset(stamps, 0, 256 * 16 * 4 * 8);
    const int sets[3][2] = {{4, 4}, {8, 8}, {4, 12}};
    for (int si = 0; si < 3; ++si) {
        const int nm = sets[si][0], nv = sets[si][1];
        scenario<K_FMA64, 0>(out, stamps, nm, nv, 0);
        scenario<K_FMA64, 0>(out, stamps, nm, nv, 3);
        scenario<K_FMA64_DEP, 0>(out, stamps, nm, nv, 0);
        scenario<K_FMA64_HALF, 0>(out, stamps, nm, nv, 0);
        scenario<K_MUL64, 0>(out, stamps, nm, nv, 0);
        scenario<K_ADD64, 0>(out, stamps, nm, nv, 0);
        scenario<K_FMA32, 0>(out, stamps, nm, nv, 0);
        scenario<K_MAXI32, 0>(out, stamps, nm, nv, 0);
        scenario<K_MOVDPP, 0>(out, stamps, nm, nv, 0);
        scenario<K_CMP64, 0>(out, stamps, nm, nv, 0);
        scenario<K_CNDMASK, 0>(out, stamps, nm, nv, 0);
        scenario<K_DSREAD64, 0>(out, stamps, nm, nv, 0);
        scenario<K_SALU, 0>(out, stamps, nm, nv, 0);
        scenario<K_FMA64, 1>(out, stamps, nm, nv, 0);
        scenario<K_MAXI32, 1>(out, stamps, nm, nv, 0);
        scenario<K_SALU, 1>(out, stamps, nm, nv, 0);
    }
    return 0;
}
