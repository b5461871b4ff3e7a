// What does a plain copy reach on THIS box?  torch's copy kernel is bench.py's `box.hbm_copy_GBps`; this probe tries the variants a hand-written
// kernel has: bytes per lane, loads in flight per lane, grid size, non-temporal hints -- and hipMemcpyAsync.
//   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o /tmp/copy_probe && /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UN - 1) * stride < n; i += UN * stride) {
        v4 x[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) x[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) { if (NT) __builtin_nontemporal_store(x[u], b + i + u * stride); else b[i + u * stride] = x[u]; }
    }
    for (; i < n; i += stride) b[i] = a[i];
}
// contiguous chunk per workgroup (the marching kernels' pattern, one array)
template <int UN>
__global__ __launch_bounds__(256) void k_copy_chunk(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) if (i + u * 256 < hi) b[i + u * 256] = a[i + u * 256];
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) f();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    v4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    auto rep = [&](const char *name, float ms) { printf("%-44s %7.1f us  %.2f TB/s\n", name, ms * 1e3, 2.0 * bytes / ms / 1e9); fflush(stdout); };
    for (int wg : {4, 8, 16, 32}) {
        char nm[96];
        snprintf(nm, 96, "grid-stride 16B/lane UN=1 grid=%dxCUs", wg); rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(cus * wg), dim3(256), 0, 0, a, b, n); }, 20));
        snprintf(nm, 96, "grid-stride 16B/lane UN=4 grid=%dxCUs", wg); rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(cus * wg), dim3(256), 0, 0, a, b, n); }, 20));
        snprintf(nm, 96, "grid-stride 16B/lane UN=4 NT grid=%dxCUs", wg); rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(cus * wg), dim3(256), 0, 0, a, b, n); }, 20));
        snprintf(nm, 96, "chunk per WG 16B/lane UN=4 grid=%dxCUs", wg); rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy_chunk<4>), dim3(cus * wg), dim3(256), 0, 0, a, b, n); }, 20));
    }
    rep("one element per thread (n/256 workgroups)", timeit([&] { hipLaunchKernelGGL((k_copy<1, false>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n); }, 20));
    rep("hipMemcpyAsync device-to-device", timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 20));
    return 0;
}
