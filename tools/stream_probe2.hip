// Microbenchmark (tuning aid, not product): HBM streaming rate of the FUSED STAGE's access pattern -- a workgroup of 256 lanes
// walks up LY rows of a strip reading NR fields and writing NW fields per row (8-byte accesses, next row's loads issued before
// this row's stores, like k_tendency_*_march) -- for the three RK3 stage patterns (4R 8W, 8R 8W, 8R 4W) and the copy-like 2R 2W.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe2.hip -o tools/stream_probe2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
struct P { const double *r[12]; double *w[8]; };
template <int NR, int NW>
__global__ __launch_bounds__(256, 3) void k(P p, int Nx, int Ny, long sy, int LY, int nstrips) {
    const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
    const int x = strip * 250 + threadIdx.x - 3;
    const int xc = x < 0 ? 0 : (x >= Nx ? Nx - 1 : x);
    const bool ok = threadIdx.x >= 3 && threadIdx.x < 253 && x < Nx;
    const int J0 = seg * LY, J1 = min(J0 + LY, Ny);
    double v[NR];
#pragma unroll
    for (int f = 0; f < NR; ++f) v[f] = p.r[f][(long)J0 * sy + xc];
    for (int j = J0; j < J1; ++j) {
        double n[NR];
        const long on = (long)min(j + 1, Ny - 1) * sy + xc;
#pragma unroll
        for (int f = 0; f < NR; ++f) n[f] = p.r[f][on];
        double s = 0;
#pragma unroll
        for (int f = 0; f < NR; ++f) s += v[f];
        if (ok) {
            const long o = (long)j * sy + x;
#pragma unroll
            for (int f = 0; f < NW; ++f) p.w[f][o] = s + f;
        }
#pragma unroll
        for (int f = 0; f < NR; ++f) v[f] = n[f];
    }
}
// same pattern with TWO rows of loads in flight
template <int NR, int NW>
__global__ __launch_bounds__(256, 3) void k2(P p, int Nx, int Ny, long sy, int LY, int nstrips) {
    const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
    const int x = strip * 250 + threadIdx.x - 3;
    const int xc = x < 0 ? 0 : (x >= Nx ? Nx - 1 : x);
    const bool ok = threadIdx.x >= 3 && threadIdx.x < 253 && x < Nx;
    const int J0 = seg * LY, J1 = min(J0 + LY, Ny);
    double v[NR], n1[NR];
#pragma unroll
    for (int f = 0; f < NR; ++f) { v[f] = p.r[f][(long)J0 * sy + xc]; n1[f] = p.r[f][(long)min(J0 + 1, Ny - 1) * sy + xc]; }
    for (int j = J0; j < J1; j += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            double n2[NR];
            const long on = (long)min(j + u + 2, Ny - 1) * sy + xc;
#pragma unroll
            for (int f = 0; f < NR; ++f) n2[f] = p.r[f][on];
            double s = 0;
#pragma unroll
            for (int f = 0; f < NR; ++f) s += v[f];
            if (ok && j + u < J1) {
                const long o = (long)(j + u) * sy + x;
#pragma unroll
                for (int f = 0; f < NW; ++f) p.w[f][o] = s + f;
            }
#pragma unroll
            for (int f = 0; f < NR; ++f) { v[f] = n1[f]; n1[f] = n2[f]; }
        }
    }
}
template <int NR, int NW> void run(P p, int N, long sy) {
    const int LY = 92, nstrips = (N + 249) / 250, nseg = (N + LY - 1) / LY;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<NR, NW>), dim3(nstrips * nseg), dim3(256), 0, 0, p, N, N, sy, LY, nstrips);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<NR, NW>), dim3(nstrips * nseg), dim3(256), 0, 0, p, N, N, sy, LY, nstrips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
    printf("%dR %dW (%3d B/cell): %7.1f us  %6.0f GB/s", NR, NW, 8 * (NR + NW), ms * 1e3, 8.0 * (NR + NW) * N * N / (ms * 1e-3) / 1e9);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k2<NR, NW>), dim3(nstrips * nseg), dim3(256), 0, 0, p, N, N, sy, LY, nstrips);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k2<NR, NW>), dim3(nstrips * nseg), dim3(256), 0, 0, p, N, N, sy, LY, nstrips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); ms /= 50;
    printf("   | two rows in flight: %7.1f us  %6.0f GB/s\n", ms * 1e3, 8.0 * (NR + NW) * N * N / (ms * 1e-3) / 1e9);
}
int main() {
    const int N = 4096; const long sy = N + 6; const size_t bytes = (size_t)sy * (N + 6) * 8;
    P p; double *b[20];
    for (int i = 0; i < 20; ++i) { hipMalloc(&b[i], bytes); hipMemset(b[i], 0, bytes); }
    for (int i = 0; i < 12; ++i) p.r[i] = b[i];
    for (int i = 0; i < 8; ++i) p.w[i] = b[12 + i];
    run<2, 2>(p, N, sy); run<4, 8>(p, N, sy); run<8, 8>(p, N, sy); run<8, 4>(p, N, sy); run<4, 4>(p, N, sy); run<12, 4>(p, N, sy);
    return 0;
}
