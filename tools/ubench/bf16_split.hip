// Micro-benchmark (round 2): layer 2 of the surrogate MLP on bf16 MFMAs with an exact three-way bf16 split of both
// operands (fp32 = 3 x 8 significant bits), against the fp32 MFMA chain the kernel uses today.
//   accuracy : C[16 coef x 16 samples] = sum_k W2[k][coef] * H[k][sample], K = 2048, vs an fp64 host reference
//   speed    : ticks per 16-hidden record for both formulations (registers only; 1, 2 or 4 waves per SIMD), including the
//              split's VALU work (v_cvt_pk_bf16_f32 + v_dot2_f32_bf16) and the relu
// 16x16x32 bf16 operand layout: lane l holds row/col (l % 16) and the 8 consecutive K slots 8*(l/16) .. +7.
// K-slot assignment per lane group g = l/16 (hidden rows 4g..4g+3 of the record, i = 0..3):
//   MFMA1: slots (i, 4+i) = A (w1_i, w2_i) x B (h1_i, h2_i)   -> h1 w1 + h2 w2
//   MFMA2:                  A (w1_i, w2_i) x B (h2_i, h1_i)   -> h2 w1 + h1 w2
//   MFMA3:                  A (w3_i, w1_i) x B (h1_i, h3_i)   -> h1 w3 + h3 w1        (dropped: h2 w3, h3 w2, h3 w3 <= 2^-25)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// builtins, not inline asm: the compiler's hazard recognizer must see these instructions (with inline asm the split returned
// garbage: missing wait states / dst == src2 in v_dot2)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {        // v_cvt_pk_bf16_f32 (RNE): lo = a, hi = b
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// x - float(bf16 half of pk): HI = 0 low half, 1 high half.  v_dot2(c)_f32_bf16: d = a.lo*b.lo + a.hi*b.hi + c with b = (-1, 0) or
// (0, -1) (gfx950 has no v_fma_mix_f32_bf16); the result is exactly representable, so rounding inside the dot product keeps it exact
// The selector must not be a compile-time constant: hipcc 7.2 folds {-1, 0} into the inline constant -1.0, which the hardware reads as
// the fp32 bit pattern 0xBF800000 = (lo 0, hi -1) (build_dbg/dbg_dot2b.hip) -- so it is materialised through an opaque s_mov.
template <int HI> __device__ __forceinline__ float sub_bf16(float x, unsigned pk) {
    unsigned sel;
    if (HI) asm("s_mov_b32 %0, 0xbf800000" : "=s"(sel)); else asm("s_mov_b32 %0, 0xbf80" : "=s"(sel));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, sel), x, false);
}
__device__ __forceinline__ void split3(const f32x4 h, unsigned (&p1)[2], unsigned (&p2)[2], unsigned (&p3)[2]) {
    p1[0] = cvt_pk_bf16(h[0], h[1]); p1[1] = cvt_pk_bf16(h[2], h[3]);
    const float r0 = sub_bf16<0>(h[0], p1[0]), r1 = sub_bf16<1>(h[1], p1[0]), r2 = sub_bf16<0>(h[2], p1[1]), r3 = sub_bf16<1>(h[3], p1[1]);
    p2[0] = cvt_pk_bf16(r0, r1); p2[1] = cvt_pk_bf16(r2, r3);
    const float s0 = sub_bf16<0>(r0, p2[0]), s1 = sub_bf16<1>(r1, p2[0]), s2 = sub_bf16<0>(r2, p2[1]), s3 = sub_bf16<1>(r3, p2[1]);
    p3[0] = cvt_pk_bf16(s0, s1); p3[1] = cvt_pk_bf16(s2, s3);
}
// (builtins, not inline asm: the compiler's hazard recognizer must see the MFMAs to space them from the VALU writes of their operands)
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- accuracy: one wave, K = 16 * nrec.  H[k][16], W[k][16] fp32 row-major.
__global__ void k_accuracy(const float* H, const float* W, int nrec, float* c_f32, float* c_bf16) {
    const int l = threadIdx.x, n = l & 15, g = l >> 4;
    f32x4 acc32 = {0, 0, 0, 0}, acc16 = {0, 0, 0, 0};
    for (int rec = 0; rec < nrec; ++rec) {
        const int k0 = rec * 16;
        f32x4 h, w;
        for (int i = 0; i < 4; ++i) { h[i] = H[(k0 + 4 * g + i) * 16 + n]; w[i] = W[(k0 + 4 * g + i) * 16 + n]; }
        // fp32 chain: MFMA j contracts hidden rows {4g' + j} over the four lane groups g'
        for (int j = 0; j < 4; ++j) acc32 = mfma_f32(w[j], h[j], acc32);
        // bf16 three-way split of both operands (the weights' split is an offline step in the real kernel)
        unsigned h1[2], h2[2], h3[2], w1[2], w2[2], w3[2];
        split3(h, h1, h2, h3);
        split3(w, w1, w2, w3);
        acc16 = mfma_bf16(u32x4{w1[0], w1[1], w2[0], w2[1]}, u32x4{h1[0], h1[1], h2[0], h2[1]}, acc16);
        acc16 = mfma_bf16(u32x4{w1[0], w1[1], w2[0], w2[1]}, u32x4{h2[0], h2[1], h1[0], h1[1]}, acc16);
        acc16 = mfma_bf16(u32x4{w3[0], w3[1], w1[0], w1[1]}, u32x4{h1[0], h1[1], h3[0], h3[1]}, acc16);
    }
    for (int r = 0; r < 4; ++r) { c_f32[(4 * g + r) * 16 + n] = acc32[r]; c_bf16[(4 * g + r) * 16 + n] = acc16[r]; }
}

// ---- speed: nw waves per workgroup, all streaming records (layer 1 fp32 MFMA -> relu -> layer 2), registers only
template <int MODE>   // 0: fp32 layer 2 (4 MFMAs), 1: bf16 split layer 2 (3 MFMAs + split), 2: as 1 without the split VALU (bound)
__global__ __launch_bounds__(1024) void k_stream(float* out, long long* stamps, int nrec) {
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 c0 = {0, 0, 0, 0};
    float x = l * 0.001f, w1f = 0.5f + l * 0.002f;
    f32x4 b1 = {0.1f, -0.2f, 0.3f, -0.1f};
    float w2f[4] = {0.01f * l, 0.02f, -0.01f * l, 0.03f};
    u32x4 wa = {0x3c003c01u + l, 0x3c023c03u, 0x38003801u, 0x38023803u}, wb = {0x34003401u, 0x34023403u, 0x3c003c01u, 0x3c023c03u + l};
    const long long t0 = clock64();
    for (int rec = 0; rec < nrec; ++rec) {
        f32x4 h = mfma_f32(w1f, x, b1);
        for (int i = 0; i < 4; ++i) h[i] = fmaxf(h[i], 0.0f);
        if (MODE == 0) {
            for (int j = 0; j < 4; ++j) c0 = mfma_f32(w2f[j], h[j], c0);
        } else if (MODE == 1) {
            unsigned h1[2], h2[2], h3[2];
            split3(h, h1, h2, h3);
            c0 = mfma_bf16(wa, u32x4{h1[0], h1[1], h2[0], h2[1]}, c0);
            c0 = mfma_bf16(wa, u32x4{h2[0], h2[1], h1[0], h1[1]}, c0);
            c0 = mfma_bf16(wb, u32x4{h1[0], h1[1], h3[0], h3[1]}, c0);
        } else {
            u32x4 hb; memcpy(&hb, &h, 16);
            c0 = mfma_bf16(wa, hb, c0); c0 = mfma_bf16(wa, hb, c0); c0 = mfma_bf16(wb, hb, c0);
        }
        x += 1e-6f;
    }
    const long long t1 = clock64();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3];
    if (l == 0) stamps[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE> double stream(int nw, float* out, long long* st) {
    const int nrec = 4000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k_stream<MODE>), dim3(256), dim3(64 * nw), 0, 0, out, st, nrec); (void)hipDeviceSynchronize(); }
    std::vector<long long> h(256 * 16);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) s += h[b * 16 + w];
    return s / (256.0 * nw) / nrec;     // ticks per record per wave
}

int main() {
    const int NREC = 128, K = 16 * NREC;
    std::vector<float> H(K * 16), W(K * 16);
    srand(12345);
    auto nrm = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v); };
    for (int k = 0; k < K; ++k) for (int n = 0; n < 16; ++n) { double h = 3.0 * nrm() + 0.5; H[k * 16 + n] = h > 0 ? (float)h : 0.0f; W[k * 16 + n] = (float)(0.05 * nrm()); }
    float *dH, *dW, *d32, *d16; long long* st; float* out;
    (void)hipMalloc(&dH, H.size() * 4); (void)hipMalloc(&dW, W.size() * 4); (void)hipMalloc(&d32, 1024); (void)hipMalloc(&d16, 1024);
    (void)hipMalloc(&st, 256 * 16 * 8); (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_accuracy, dim3(1), dim3(64), 0, 0, dH, dW, NREC, d32, d16);
    std::vector<float> c32(256), c16(256);
    (void)hipMemcpy(c32.data(), d32, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(c16.data(), d16, 1024, hipMemcpyDeviceToHost);
    double e32 = 0, e16 = 0, ediff = 0, scale = 0, eseq = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double ref = 0, mag = 0; float seq = 0;
        for (int k = 0; k < K; ++k) { ref += (double)W[k * 16 + m] * H[k * 16 + n]; mag += std::fabs((double)W[k * 16 + m] * H[k * 16 + n]); seq = std::fmaf(W[k * 16 + m], H[k * 16 + n], seq); }
        e32 = std::fmax(e32, std::fabs(c32[m * 16 + n] - ref)); e16 = std::fmax(e16, std::fabs(c16[m * 16 + n] - ref));
        eseq = std::fmax(eseq, std::fabs(seq - ref));
        ediff = std::fmax(ediff, std::fabs((double)c32[m * 16 + n] - c16[m * 16 + n])); scale = std::fmax(scale, mag);
    }
    printf("K = %d: max |err| vs fp64: fp32 MFMA chain %.3e, bf16x3 split MFMA %.3e, scalar fmaf chain %.3e; max |fp32 - bf16x3| %.3e; max sum|terms| %.3e\n",
           K, e32, e16, eseq, ediff, scale);
    for (int nw : {4, 8, 16}) {
        const double a = stream<0>(nw, out, st), b = stream<1>(nw, out, st), c = stream<2>(nw, out, st);
        printf("%2d waves/CU: ticks per record per wave: fp32 layer 2 %.1f | bf16x3 layer 2 with split %.1f | bf16x3 MFMAs only %.1f  -> per SIMD per record: %.1f | %.1f | %.1f\n",
               nw, a, b, c, a * 4 / nw, b * 4 / nw, c * 4 / nw);
    }
    return 0;
}
