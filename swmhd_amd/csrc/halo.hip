// Periodic halo fill: the stand-alone engine's counterpart of Oceananigans' fill_halo_regions! for
// topology = (Periodic, Periodic, Flat) (reference: jacobian_formulation/SWMHD_example.jl:16).
#include "common.hpp"

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

}  // namespace

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
