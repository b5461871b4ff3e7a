// Periodic halo fill: the stand-alone engine's counterpart of Oceananigans' fill_halo_regions! for
// topology = (Periodic, Periodic, Flat) (reference: jacobian_formulation/SWMHD_example.jl:16).
#include "common.hpp"
// (gradient boundary values are formed as c - g*d in two roundings, like the oracle: the Makefile builds this file -ffp-contract=off)

namespace swmhd {
namespace {

// x halos: for every interior row, west halo <- east interior edge, east halo <- west interior edge.
template <typename T>
__global__ void k_halo_x(T *f, int Nx, int Ny, int Hx, long sy) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int per_row = 2 * Hx;
    if (t >= per_row * Ny) return;
    int y = t / per_row, c = t - y * per_row;
    T *row = f + (long)y * sy;
    if (c < Hx) row[-Hx + c] = row[Nx - Hx + c];
    else row[Nx + (c - Hx)] = row[c - Hx];
}

// y halos over the full padded width (corners come along because x halos are filled first).
template <typename T>
__global__ void k_halo_y(T *f, int Nx, int Ny, int Hx, int Hy, long sy) {
    int W = Nx + 2 * Hx;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)W * 2 * Hy) return;
    int r = (int)(t / W), c = (int)(t - (long)r * W) - Hx;
    if (r < Hy) f[(long)(-Hy + r) * sy + c] = f[(long)(Ny - Hy + r) * sy + c];
    else f[(long)(Ny + (r - Hy)) * sy + c] = f[(long)(r - Hy) * sy + c];
}

// All halo cells of up to 4 fields in ONE launch: halo cell (x, y) <- interior (x mod Nx, y mod Ny).  Reads only
// interior cells, writes only halo cells => no ordering between x and y halos, corners included.
template <typename T>
struct HaloMulti {
    T *f[4];
    int nf, Nx, Ny, Hx, Hy, which;
    long sy;
};
template <typename T>
__global__ void k_halo_multi(HaloMulti<T> a) {
    // cells are enumerated as: [south+north strips: 2*Hy rows x (Nx+2Hx)] then [west+east strips: Ny rows x 2*Hx]
    const int W = a.Nx + 2 * a.Hx;
    const long nsn = (a.which & 2) ? (long)2 * a.Hy * W : 0;
    const long nwe = (a.which & 1) ? (long)a.Ny * 2 * a.Hx : 0;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsn + nwe) return;
    int x, y;
    if (t < nsn) {
        int r = (int)(t / W);
        x = (int)(t - (long)r * W) - a.Hx;
        y = r < a.Hy ? r - a.Hy : a.Ny + (r - a.Hy);
        if (!(a.which & 1) && (x < 0 || x >= a.Nx)) return;   // y-only fill: leave corners to the caller's x fill
    } else {
        t -= nsn;
        int r = (int)(t / (2 * a.Hx)), c = (int)(t - (long)r * 2 * a.Hx);
        y = r;
        x = c < a.Hx ? c - a.Hx : a.Nx + (c - a.Hx);
    }
    int sx = x < 0 ? x + a.Nx : (x >= a.Nx ? x - a.Nx : x);
    int sy_ = y < 0 ? y + a.Ny : (y >= a.Ny ? y - a.Ny : y);
    if (!(a.which & 2)) sy_ = y;
    if (!(a.which & 1)) sx = x;
    const long dst = (long)y * a.sy + x, src = (long)sy_ * a.sy + sx;
    for (int k = 0; k < a.nf; ++k) a.f[k][dst] = a.f[k][src];
}

// fill_halo_regions! with boundary conditions, one pass per direction (x pass over interior rows, then y pass over the padded
// width, like the oracle and like Oceananigans' west/east-then-south/north order).  One thread per (field, line): the line's halo
// cells are few (<= 2 Hx) and the passes are launch-latency bound anyway.
template <typename T, int DIR>
__global__ void k_halo_bc(HaloBc<T> a) {
    const int nlines = DIR == 0 ? a.Ny : a.Nx + 2 * a.Hx;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines * a.nf) return;
    const int fld = t / nlines, line = t - fld * nlines;
    T *f = a.f[fld];
    const int N = DIR == 0 ? a.Nx : a.Ny, H = DIR == 0 ? a.Hx : a.Hy;
    const int topo = DIR == 0 ? a.topo_x : a.topo_y;
    const bool face = ((DIR == 0 ? a.face_x : a.face_y) >> fld) & 1;
    const T d = DIR == 0 ? a.dx : a.dy;
    const T glo = a.grad[fld][DIR == 0 ? 0 : 2], ghi = a.grad[fld][DIR == 0 ? 1 : 3];
    // element k (0-based along the direction, halo cells have k < 0 or k >= N) of this line
    auto at = [&](int k) -> T & { return DIR == 0 ? f[(long)line * a.sy + k] : f[(long)k * a.sy + (line - a.Hx)]; };
    if (topo == 0) {
        for (int m = 1; m <= H; ++m) { at(-m) = at(N - m); at(N + m - 1) = at(m - 1); }
    } else if (face) {                                      // impenetrable walls at Julia indices 1 and N+1; zeros beyond them
        at(0) = T(0);
        for (int m = 1; m <= H; ++m) { at(-m) = T(0); at(N + m - 1) = T(0); }
    } else {                                                // gradient side: first halo point extrapolated, the others zero
        for (int m = 1; m <= H; ++m) at(-m) = (glo == glo) ? (m == 1 ? at(0) - glo * d : T(0)) : at(m - 1);
        for (int m = 1; m <= H; ++m) at(N + m - 1) = (ghi == ghi) ? (m == 1 ? at(N - 1) + ghi * d : T(0)) : at(N - m);
    }
}

}  // namespace

template <typename T>
hipError_t launch_fill_halo_bc(const HaloBc<T> &a, hipStream_t s) {
    const int nx = a.Ny * a.nf, ny = (a.Nx + 2 * a.Hx) * a.nf;
    hipLaunchKernelGGL((k_halo_bc<T, 0>), dim3((nx + 127) / 128), dim3(128), 0, s, a);
    hipLaunchKernelGGL((k_halo_bc<T, 1>), dim3((ny + 127) / 128), dim3(128), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_fill_halo_bc<double>(const HaloBc<double> &, hipStream_t);
template hipError_t launch_fill_halo_bc<float>(const HaloBc<float> &, hipStream_t);

template <typename T>
hipError_t launch_fill_halo_periodic_multi(T *const *f, int nf, int Nx, int Ny, int Hx, int Hy, long sy, int which,
                                           hipStream_t s) {
    HaloMulti<T> a;
    for (int k = 0; k < 4; ++k) a.f[k] = k < nf ? f[k] : nullptr;
    a.nf = nf; a.Nx = Nx; a.Ny = Ny; a.Hx = Hx; a.Hy = Hy; a.which = which; a.sy = sy;
    long n = ((which & 2) ? (long)2 * Hy * (Nx + 2 * Hx) : 0) + ((which & 1) ? (long)Ny * 2 * Hx : 0);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((k_halo_multi<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_fill_halo_periodic_multi<double>(double *const *, int, int, int, int, int, long, int, hipStream_t);
template hipError_t launch_fill_halo_periodic_multi<float>(float *const *, int, int, int, int, int, long, int, hipStream_t);

template <typename T>
hipError_t launch_fill_halo_periodic(T *f, int Nx, int Ny, int Hx, int Hy, long sy, int which, hipStream_t s) {
    if ((which & 1) && Hx > 0) {
        int n = 2 * Hx * Ny;
        hipLaunchKernelGGL((k_halo_x<T>), dim3((n + 255) / 256), dim3(256), 0, s, f, Nx, Ny, Hx, sy);
    }
    if ((which & 2) && Hy > 0) {
        long n = (long)(Nx + 2 * Hx) * 2 * Hy;
        hipLaunchKernelGGL((k_halo_y<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f, Nx, Ny, Hx, Hy, sy);
    }
    return hipGetLastError();
}
template hipError_t launch_fill_halo_periodic<double>(double *, int, int, int, int, long, int, hipStream_t);
template hipError_t launch_fill_halo_periodic<float>(float *, int, int, int, int, long, int, hipStream_t);

}  // namespace swmhd
