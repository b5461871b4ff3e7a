// Energy and extrema diagnostics in one pass over the prognostic fields.
//
// The reference evaluates these every iteration through Oceananigans AbstractOperations + NetCDFOutputWriter:
//   kinetic/magnetic/potential/total_energy_func   jacobian_formulation/SWMHD_example.jl:67-77, :87-92
//                                                  divergence_formulation/divergence_sw_mhd.jl:63-74, :85-91
//   progress callback max|u|, max|A|, min h        SWMHD_example.jl:47-65
// Definitions: the reference's expressions evaluated the way Oceananigans' AbstractOperations place them -- a binary operation of
// two fields at different locations is located where its FIRST operand is, the second is interpolated there (ℑ); a division by a
// field interpolates the divisor (library-internal and unpinned, like A9):
//   Jacobian driver    KE = Σ ½ h · ℑxᶜ[u² + ℑxyᶠᶜ(v²)] Δx Δy                       mean((1/2)*h*(u^2 + v^2))*Lx*Ly          SWMHD_example.jl:74
//   divergence driver  KE = Σ ½ (1/h) · ℑxᶜ[uh² + ℑxyᶠᶜ(vh²)] Δx Δy                 mean((1/2)*(1/h)*(uh^2 + vh^2))*Lx*Ly   divergence_sw_mhd.jl:71
//   both               ME = Σ ½ h · ℑyᶜ[Bx² + ℑxyᶜᶠ(By²)] Δx Δy,  Bx = −∂yA/ℑyᶠh @cfc, By = ∂xA/ℑxᶠh @fcc   (:69-70,:72 / :67-68,:75)
//                      PE = Σ ½ g (h − hᵢ)² Δx Δy                                                                             (:73 / :76)
//   max|u|, max|v| (divergence driver: u = uh/ℑxᶠh, v = vh/ℑyᶠh), max|A|, min h over the interior                                                            SWMHD_example.jl:47-65
// Two deterministic stages: per-workgroup partials (fixed grid), then one workgroup folds them in index order.
#include "common.hpp"

namespace swmhd {
namespace {

constexpr int NB = 1024, NT = 256, NQ = 7;

template <typename T>
struct DiagArgs {
    const T *q1, *q2, *h, *A;
    int Nx, Ny, j0, j1;
    long sy;
    T dx, dy, grav, href;
    int form;
    double *part;   // [NB][NQ]
    double *out;    // [NQ]
};

__device__ __forceinline__ void fold(double *acc, const double *v) {
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2];
    acc[3] = fmax(acc[3], v[3]); acc[4] = fmax(acc[4], v[4]); acc[5] = fmax(acc[5], v[5]); acc[6] = fmin(acc[6], v[6]);
}

__device__ void block_reduce(double *acc, double (*sm)[NQ]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q) sm[t][q] = acc[q];
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (t < s) {
            double a[NQ], b[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { a[q] = sm[t][q]; b[q] = sm[t + s][q]; }
            fold(a, b);
#pragma unroll
            for (int q = 0; q < NQ; ++q) sm[t][q] = a[q];
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void k_diag_partial(DiagArgs<T> a) {
    __shared__ double sm[NT][NQ];
    double acc[NQ] = {0, 0, 0, 0, 0, 0, 1e300};
    const double rdx = 1.0 / (double)a.dx, rdy = 1.0 / (double)a.dy;
    const long ncell = (long)a.Nx * (a.j1 - a.j0);
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < ncell; e += (long)NB * NT) {
        const int y = a.j0 + (int)(e / a.Nx), x = (int)(e % a.Nx);
        const long o = (long)y * a.sy + x;
        auto Q1 = [&](int di, int dj) -> double { return (double)a.q1[o + dj * a.sy + di]; };
        auto Q2 = [&](int di, int dj) -> double { return (double)a.q2[o + dj * a.sy + di]; };
        auto H = [&](int di, int dj) -> double { return (double)a.h[o + dj * a.sy + di]; };
        auto AA = [&](int di, int dj) -> double { return (double)a.A[o + dj * a.sy + di]; };
        const double hc = H(0, 0);
        // kinetic energy: W(i) = q1² + ℑxyᶠᶜ(q2²) at the faces i and i+1, then ℑxᶜ
        auto W = [&](int di) -> double {
            const double p = Q1(di, 0);
            const double g00 = Q2(di - 1, 0), g10 = Q2(di, 0), g01 = Q2(di - 1, 1), g11 = Q2(di, 1);
            return p * p + 0.5 * (0.5 * (g00 * g00 + g10 * g10) + 0.5 * (g01 * g01 + g11 * g11));
        };
        const double wbar = 0.5 * (W(0) + W(1));
        const double ke = a.form == 0 ? 0.5 * (1.0 / hc) * wbar : 0.5 * hc * wbar;
        // magnetic energy: Z(j) = Bx² + ℑxyᶜᶠ(By²) at the faces j and j+1, then ℑyᶜ
        auto BX = [&](int di, int dj) -> double { return -((AA(di, dj) - AA(di, dj - 1)) * rdy) / (0.5 * (H(di, dj - 1) + H(di, dj))); };
        auto BY = [&](int di, int dj) -> double { return ((AA(di, dj) - AA(di - 1, dj)) * rdx) / (0.5 * (H(di - 1, dj) + H(di, dj))); };
        auto Z = [&](int dj) -> double {
            const double b = BX(0, dj);
            const double c00 = BY(0, dj - 1), c10 = BY(1, dj - 1), c01 = BY(0, dj), c11 = BY(1, dj);
            return b * b + 0.5 * (0.5 * (c00 * c00 + c10 * c10) + 0.5 * (c01 * c01 + c11 * c11));
        };
        const double me = 0.5 * hc * (0.5 * (Z(0) + Z(1)));
        // progress callback: maximum(abs, u) with u = uh / h located at uh's faces (divergence_sw_mhd.jl:45-47,53); u itself for the Jacobian driver
        double uw = Q1(0, 0), vs = Q2(0, 0);
        if (a.form == 0) { uw /= 0.5 * (H(-1, 0) + hc); vs /= 0.5 * (H(0, -1) + hc); }
        const double dh = hc - (double)a.href;
        const double pe = 0.5 * (double)a.grav * dh * dh;
        const double v[NQ] = {ke, me, pe, fabs(uw), fabs(vs), fabs(AA(0, 0)), hc};
        fold(acc, v);
    }
    block_reduce(acc, sm);
    if (threadIdx.x == 0)
        for (int q = 0; q < NQ; ++q) a.part[(long)blockIdx.x * NQ + q] = sm[0][q];
}

template <typename T>
__global__ __launch_bounds__(NT) void k_diag_final(DiagArgs<T> a) {
    __shared__ double sm[NT][NQ];
    double acc[NQ] = {0, 0, 0, 0, 0, 0, 1e300};
    for (int b = threadIdx.x; b < NB; b += NT) fold(acc, a.part + (long)b * NQ);
    block_reduce(acc, sm);
    if (threadIdx.x == 0) {
        const double cell = (double)a.dx * (double)a.dy;
        a.out[0] = sm[0][0] * cell; a.out[1] = sm[0][1] * cell; a.out[2] = sm[0][2] * cell;
        for (int q = 3; q < NQ; ++q) a.out[q] = sm[0][q];
    }
}

}  // namespace

template <typename T>
hipError_t launch_diagnostics(const T *q1, const T *q2, const T *h, const T *A, int Nx, int Ny, int j0, int j1, long sy, T dx, T dy,
                              T grav, T href, int form, double *workspace, double *out, hipStream_t s) {
    DiagArgs<T> a{q1, q2, h, A, Nx, Ny, j0, j1, sy, dx, dy, grav, href, form, workspace, out};
    hipLaunchKernelGGL((k_diag_partial<T>), dim3(NB), dim3(NT), 0, s, a);
    hipLaunchKernelGGL((k_diag_final<T>), dim3(1), dim3(NT), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_diagnostics<double>(const double *, const double *, const double *, const double *, int, int, int, int, long,
                                               double, double, double, double, int, double *, double *, hipStream_t);
template hipError_t launch_diagnostics<float>(const float *, const float *, const float *, const float *, int, int, int, int, long, float,
                                              float, float, float, int, double *, double *, hipStream_t);

}  // namespace swmhd

// ---- measurement hook: what does the fp64 vector ALU of this box sustain right now? ------------------------------------------------
// Three waves per SIMD (the stage kernels' occupancy), four independent fma chains each: one wave-instruction leaves a 16-lane SIMD
// every 4 cycles, so ns per wave-instruction = 4 / (shader clock under fp64 load).  bench.py prints it beside the kernel times: the
// boxes of a pool differ in the clock they hold under load, and the VALU floor of the stage kernels scales with this number.
namespace swmhd {
namespace {
__global__ __launch_bounds__(256) void k_fp64_issue_probe(double *out, double a, double b, int iters) {
    double x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
    const double s = (x[0] + x[1]) + (x[2] + x[3]);
    if (s == 12345.678) out[0] = s;     // (never true: keeps the chains alive)
}
}  // namespace
}  // namespace swmhd

extern "C" int swmhd_probe_fp64_issue(double *scratch, float *ns_per_wave_instruction, void *stream) {
    if (!scratch || !ns_per_wave_instruction) return 1;   // SWMHD_EINVAL
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return 1;
    hipStream_t s = (hipStream_t)stream;
    const int W = 3, iters = 4000, blocks = cus * W;   // 4 waves per workgroup = one per SIMD, W workgroups per CU
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return 1;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return 1; }
    hipLaunchKernelGGL(swmhd::k_fp64_issue_probe, dim3(blocks), dim3(256), 0, s, scratch, 1.0000001, 1e-9, iters);   // warm
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(swmhd::k_fp64_issue_probe, dim3(blocks), dim3(256), 0, s, scratch, 1.0000001, 1e-9, iters);
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) return -(int)e;
    *ns_per_wave_instruction = (float)(ms * 1e6 / ((double)iters * 8 * 4 * W));
    return 0;
}

// ---- measurement hook: the plain-copy rate of this box, best case ---------------------------------------------------------------
// One 16-byte element per thread, one 4-KB chunk per workgroup, workgroups dispatched in address order: the access pattern that reaches
// the guide's 6.29 TB/s copy figure (tools/copy_probe.hip: 6.2 TB/s where torch's persistent copy kernel gets 5.0 and hipMemcpy 4.8).
// Kernels that carry state along a direction (the row-marching ones here) are of the persistent kind; bench.py quotes both rates.
namespace swmhd {
namespace {
typedef float probe_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_probe(const probe_v4 *__restrict__ a, probe_v4 *__restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
}  // namespace
}  // namespace swmhd

extern "C" int swmhd_probe_copy(void *dst, const void *src, size_t bytes, int reps, float *gbytes_per_s, void *stream) {
    if (!dst || !src || !gbytes_per_s || bytes < 4096 || (bytes & 15) || reps < 1) return 1;   // SWMHD_EINVAL
    hipStream_t s = (hipStream_t)stream;
    const size_t n = bytes / 16;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return 1;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return 1; }
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(swmhd::k_copy_probe, dim3(blocks), dim3(256), 0, s, (const swmhd::probe_v4 *)src, (swmhd::probe_v4 *)dst, n);
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(swmhd::k_copy_probe, dim3(blocks), dim3(256), 0, s, (const swmhd::probe_v4 *)src, (swmhd::probe_v4 *)dst, n);
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess || ms <= 0.f) return e != hipSuccess ? -(int)e : 1;
    *gbytes_per_s = (float)(2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9);
    return 0;
}
