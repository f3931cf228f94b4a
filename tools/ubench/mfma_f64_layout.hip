// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, determined empirically (result: A lane -> (i = lane % 16,
// k = lane / 16), B lane -> (k = lane / 16, j = lane % 16), D register r of a lane -> (i = 4 r + lane / 16, j = lane % 16);
// read the kk = 3 block of the output, kk = 0 factorises ambiguously):
// A[i][k] = 1000 (i + 1) + k, B[k][j] = (k == kk ? (j + 1) : 0) for each kk -> D[i][j] = (1000 (i + 1) + kk) (j + 1).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_layout.hip -o mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, int kk) {
    const int lane = threadIdx.x;
    // assumption under test: A lane -> (i = lane % 16, k = lane / 16); B lane -> (k = lane / 16, j = lane % 16)
    const double a = 1000.0 * ((lane & 15) + 1) + (lane >> 4);
    const double b = ((lane >> 4) == kk) ? (double)((lane & 15) + 1) : 0.0;
    f64x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}
int main() {
    double* d; (void)hipMalloc(&d, 256 * 8);
    double h[256];
    for (int kk = 0; kk < 4; kk += 3) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, kk);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("kk = %d: lane r -> (i, j, k)\n", kk);
        for (int lane = 0; lane < 64; lane += 5)
            for (int r = 0; r < 4; ++r) {
                const double v = h[lane * 4 + r];
                // v = (1000 (i + 1) + k) (j + 1): try every j
                for (int j = 0; j < 16; ++j) {
                    const double q = v / (j + 1);
                    const long iq = (long)(q + 0.5);
                    if (q == (double)iq && iq % 1000 == kk && iq / 1000 >= 1 && iq / 1000 <= 16) {
                        printf("  lane %2d r %d -> i %2ld j %2d   (expected by the kernel: i %2d j %2d)\n", lane, r, iq / 1000 - 1, j,
                               4 * r + (lane >> 4), lane & 15);
                        break;
                    }
                }
            }
    }
    return 0;
}
