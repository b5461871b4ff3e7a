// Microbenchmark (tuning aid, not product): HBM streaming rate of the row-marching access pattern -- a workgroup walks up a
// strip of columns reading two fields and writing two, one row per iteration -- with 8-byte and 16-byte accesses per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o gpurun_out/stream_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W>   // W doubles per lane
__global__ __launch_bounds__(256) void k_probe(const double *A, const double *h, double *Fx, double *Fy, int Nx, int Ny, long sy, int LY,
                                               int nstrips, int PF) {
    const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
    const int x = (strip * 256 + threadIdx.x) * W;
    if (x >= Nx) return;
    const int J0 = seg * LY, J1 = min(J0 + LY, Ny);
    for (int j = J0; j < J1; ++j) {
        const long o = (long)j * sy + x;
        if constexpr (W == 1) {
            double a = A[o], b = h[o];
            Fx[o] = a + b; Fy[o] = a - b;
        } else {
            double2 a = *reinterpret_cast<const double2 *>(A + o), b = *reinterpret_cast<const double2 *>(h + o);
            *reinterpret_cast<double2 *>(Fx + o) = make_double2(a.x + b.x, a.y + b.y);
            *reinterpret_cast<double2 *>(Fy + o) = make_double2(a.x - b.x, a.y - b.y);
        }
    }
}

int main() {
    const int N = 4096; const long sy = N + 8; const size_t bytes = (size_t)sy * (N + 6) * 8;
    double *A, *h, *Fx, *Fy;
    hipMalloc(&A, bytes); hipMalloc(&h, bytes); hipMalloc(&Fx, bytes); hipMalloc(&Fy, bytes);
    hipMemset(A, 0, bytes); hipMemset(h, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int W : {1, 2}) for (int LY : {16, 46, 92, 4096 / 8}) {
        const int nstrips = (N / W + 255) / 256, nseg = (N + LY - 1) / LY;
        auto run = [&]() { if (W == 1) hipLaunchKernelGGL(k_probe<1>, dim3(nstrips * nseg), dim3(256), 0, 0, A, h, Fx, Fy, N, N, sy, LY, nstrips, 0);
                           else hipLaunchKernelGGL(k_probe<2>, dim3(nstrips * nseg), dim3(256), 0, 0, A, h, Fx, Fy, N, N, sy, LY, nstrips, 0); };
        for (int i = 0; i < 5; ++i) run();
        hipEventRecord(e0);
        for (int i = 0; i < 30; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
        printf("W=%d (%2d B/lane) LY=%4d blocks=%5d : %7.1f us  %6.0f GB/s\n", W, 8 * W, LY, nstrips * nseg, ms * 1e3, 32.0 * N * N / (ms * 1e-3) / 1e9);
    }
    return 0;
}
