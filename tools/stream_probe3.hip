// Microbenchmark (tuning aid, not product): does a different TRAVERSAL raise the streaming ceiling of the fused RK3 stages?
// Pure load/store kernels with the stage-2 pattern (8 fields read, 8 written per cell: 128 B/cell) and the stage-3 pattern (8R 4W),
// 4096^2 fp64 fields with a 3-cell halo, varying: lanes per workgroup (row length per stream: 2 KB .. 8 KB), bytes per lane (8 / 16),
// rows per segment, block -> (strip, segment) order (with / without the XCD-contiguous remap), and a plain grid-stride 16-B copy of
// the same bytes as the upper bound.  Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe3.hip -o tools/stream_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
struct P { const double *r[8]; double *w[8]; };
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    unsigned q = nblk / 8, r = nblk % 8, x = bid % 8, k = bid / 8;
    unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + k;
}
// W = doubles per lane (1 or 2); ORDER: 0 strip-fastest, 1 strip-fastest + XCD remap, 2 segment-fastest
template <int NR, int NW, int NT, int W, int ORDER>
__global__ __launch_bounds__(NT) void k(P p, int Nx, int Ny, long sy, int LY, int nstrips, int nseg) {
    unsigned bid = blockIdx.x;
    if (ORDER == 1) bid = xcd_remap(bid, nstrips * nseg);
    const int strip = ORDER == 2 ? bid / nseg : bid % nstrips, seg = ORDER == 2 ? bid % nseg : bid / nstrips;
    constexpr int COLS = NT * W;
    const int x = strip * (COLS - 6) + threadIdx.x * W - 3 + 3;      // (halo handling folded away: aligned columns)
    const bool ok = x + W <= Nx;
    const int xc = ok ? x : 0;
    const int J0 = seg * LY, J1 = min(J0 + LY, Ny);
    typedef double vec __attribute__((ext_vector_type(W == 1 ? 1 : 2)));
    vec v[NR];
#pragma unroll
    for (int f = 0; f < NR; ++f) v[f] = *(const vec *)(p.r[f] + (long)J0 * sy + xc);
    for (int j = J0; j < J1; ++j) {
        vec n[NR];
        const long on = (long)min(j + 1, Ny - 1) * sy + xc;
#pragma unroll
        for (int f = 0; f < NR; ++f) n[f] = *(const vec *)(p.r[f] + on);
        vec s = v[0];
#pragma unroll
        for (int f = 1; f < NR; ++f) s += v[f];
        if (ok) {
            const long o = (long)j * sy + x;
#pragma unroll
            for (int f = 0; f < NW; ++f) *(vec *)(p.w[f] + o) = s + (double)f;
        }
#pragma unroll
        for (int f = 0; f < NR; ++f) v[f] = n[f];
    }
}
// plain copy of the same bytes: NR arrays read, NW written, 16 B per lane, grid-stride
template <int NR, int NW>
__global__ __launch_bounds__(256) void kcopy(P p, long n2) {
    typedef double vec __attribute__((ext_vector_type(2)));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
        vec s = ((const vec *)p.r[0])[i];
#pragma unroll
        for (int f = 1; f < NR; ++f) s += ((const vec *)p.r[f])[i];
#pragma unroll
        for (int f = 0; f < NW; ++f) ((vec *)p.w[f])[i] = s + (double)f;
    }
}
static float timeit(void (*launch)(void *), void *ctx) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 60; ++i) launch(ctx);
    hipEventRecord(e0);
    for (int i = 0; i < 40; ++i) launch(ctx);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 40;
}
struct Ctx { P p; int N; long sy; int LY; };
template <int NR, int NW, int NT, int W, int ORDER> void run(Ctx c, const char *tag) {
    const int cols = NT * W - 6, nstrips = (c.N + cols - 1) / cols, nseg = (c.N + c.LY - 1) / c.LY;
    struct L { static void go(void *q) { Ctx *c = (Ctx *)q; const int cols = NT * W - 6, ns = (c->N + cols - 1) / cols, ng = (c->N + c->LY - 1) / c->LY;
        hipLaunchKernelGGL((k<NR, NW, NT, W, ORDER>), dim3(ns * ng), dim3(NT), 0, 0, c->p, c->N, c->N, c->sy, c->LY, ns, ng); } };
    const float ms = timeit(L::go, &c);
    printf("%dR%dW NT=%4d W=%d LY=%3d order=%d grid=%5d %-18s: %7.1f us  %6.0f GB/s\n", NR, NW, NT, W, c.LY, ORDER, nstrips * nseg, tag, ms * 1e3,
           8.0 * (NR + NW) * c.N * c.N / (ms * 1e-3) / 1e9);
}
template <int NR, int NW> void run_copy(Ctx c) {
    struct L { static void go(void *q) { Ctx *c = (Ctx *)q; hipLaunchKernelGGL((kcopy<NR, NW>), dim3(256 * 16), dim3(256), 0, 0, c->p, (long)c->N * c->N / 2); } };
    const float ms = timeit(L::go, &c);
    printf("%dR%dW plain 16-B grid-stride copy                         : %7.1f us  %6.0f GB/s\n", NR, NW, ms * 1e3, 8.0 * (NR + NW) * c.N * c.N / (ms * 1e-3) / 1e9);
}
template <int NR, int NW> void sweep(Ctx c) {
    run_copy<NR, NW>(c);
    c.LY = 92;  run<NR, NW, 256, 1, 1>(c, "current");
    c.LY = 92;  run<NR, NW, 256, 1, 0>(c, "no xcd remap");
    c.LY = 92;  run<NR, NW, 256, 1, 2>(c, "segment-fastest");
    c.LY = 46;  run<NR, NW, 256, 1, 1>(c, "LY 46");
    c.LY = 184; run<NR, NW, 256, 1, 1>(c, "LY 184");
    c.LY = 512; run<NR, NW, 256, 1, 1>(c, "LY 512");
    c.LY = 92;  run<NR, NW, 512, 1, 1>(c, "512 lanes");
    c.LY = 184; run<NR, NW, 512, 1, 1>(c, "512 lanes LY 184");
    c.LY = 92;  run<NR, NW, 1024, 1, 1>(c, "1024 lanes");
    c.LY = 92;  run<NR, NW, 256, 2, 1>(c, "16 B per lane");
    c.LY = 184; run<NR, NW, 256, 2, 1>(c, "16 B/lane LY 184");
    c.LY = 92;  run<NR, NW, 128, 2, 1>(c, "128 lanes x 16 B");
    c.LY = 92;  run<NR, NW, 128, 1, 1>(c, "128 lanes");
    c.LY = 92;  run<NR, NW, 64, 1, 1>(c, "64 lanes");
}
int main() {
    const int N = 4096; const long sy = N + 8; const size_t bytes = (size_t)sy * (N + 8) * 8;
    Ctx c; c.N = N; c.sy = sy; c.LY = 92;
    double *b[16];
    for (int i = 0; i < 16; ++i) { hipMalloc(&b[i], bytes); hipMemset(b[i], 0, bytes); }
    for (int i = 0; i < 8; ++i) { c.p.r[i] = b[i]; c.p.w[i] = b[8 + i]; }
    sweep<8, 8>(c);
    sweep<8, 4>(c);
    sweep<4, 8>(c);
    return 0;
}
