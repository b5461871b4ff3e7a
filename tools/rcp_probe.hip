// How accurate is v_rcp_f64 on gfx950, bare and after 1 / 2 Newton steps?  (decides how many fma the fast reciprocal needs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    double e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r);
    r1[i] = r;
    e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = ldexp(1.0 + (double)rand() / RAND_MAX, rand() % 80 - 40);
    double *x, *r0, *r1, *r2;
    hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, r0, r1, r2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / (long double)h[i];
        e0 = fmax(e0, (double)fabsl((a[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((b[i] - ex) / ex)); e2 = fmax(e2, (double)fabsl((c[i] - ex) / ex));
    }
    printf("max rel err: bare v_rcp_f64 %.3e   +1 Newton %.3e   +2 Newton %.3e   (eps = %.3e)\n", e0, e1, e2, ldexp(1.0, -53));
    return 0;
}
