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

}  // namespace

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
