// v_rcp_f64 followed by one / two Newton steps against the correctly rounded quotient 1 / x: max error in ulp over 2^22 inputs
// (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/rcp_f64_steps.hip -o /tmp/rcp_steps && /tmp/rcp_steps)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = x[i];
    double rc = __builtin_amdgcn_rcp(s);
    r0[i] = rc;
    rc = fma(fma(-s, rc, 1.0), rc, rc);
    r1[i] = rc;
    rc = fma(fma(-s, rc, 1.0), rc, rc);
    r2[i] = rc;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;
        x[i] = std::exp((u - 0.5) * 40.0);          // 2e-9 .. 5e8
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double want = 1.0L / (long double)x[i];
        const double ulp = std::nextafter((double)want, INFINITY) - (double)want;
        e0 = std::fmax(e0, std::fabs((double)((long double)a[i] - want)) / ulp);
        e1 = std::fmax(e1, std::fabs((double)((long double)b[i] - want)) / ulp);
        e2 = std::fmax(e2, std::fabs((double)((long double)c[i] - want)) / ulp);
    }
    printf("v_rcp_f64: max error %.3g ulp; + one Newton step %.3g ulp; + two %.3g ulp\n", e0, e1, e2);
    return 0;
}
