// Shared host/device declarations for libswmhd (gfx950).  Internal: the public surface is include/swmhd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace swmhd {

// Arguments of the Lorentz operator kernels.  Pointers address interior cell (i,j) = (1,1) of the
// halo-padded parent (Julia indexing), so 0-based cell (x,y) is ptr[y*sy + x] and halo cells have
// negative x / y.
template <typename T>
struct OpArgs {
    const T *A;
    const T *h;
    T *Fx;
    T *Fy;
    int Nx, Ny, Hx, Hy;
    long sy;
    T dx, dy, rdx, rdy;
    int j0, j1;  // rows [j0, j1) are computed (0-based)
    int topo_x, topo_y;
    int kernel_variant;  // 0 = by size, 1 = LDS-tiled kernel, 2 = row-marching kernel
    int edge_cols;       // LDS-tiled divergence kernel: only tile column 0 and the last one (or two): the x-wall frame of a Bounded grid
};

// launchers, one pair per translation unit (fast: reciprocal multiplies + FMA; strict: reference op order,
// compiled with -ffp-contract=off)
template <typename T> hipError_t launch_lorentz_jacobian_fast(const OpArgs<T> &a, hipStream_t s);
template <typename T> hipError_t launch_lorentz_jacobian_strict(const OpArgs<T> &a, hipStream_t s);
template <typename T> hipError_t launch_lorentz_divergence_fast(const OpArgs<T> &a, hipStream_t s);
template <typename T> hipError_t launch_lorentz_divergence_strict(const OpArgs<T> &a, hipStream_t s);

template <typename T>
hipError_t launch_fill_halo_periodic(T *interior, int Nx, int Ny, int Hx, int Hy, long sy, int which, hipStream_t s);
// up to 4 fields in one launch; corners resolved by wrapping both coordinates, so x and y halos need no ordering
template <typename T>
hipError_t launch_fill_halo_periodic_multi(T *const *interiors, int nfields, int Nx, int Ny, int Hx, int Hy, long sy,
                                           int which, hipStream_t s);
// fill_halo_regions! for any (Periodic | Bounded) topology pair with Oceananigans' default boundary conditions or gradient BCs
// (oracle_fill_halo): west/east pass, then south/north pass over the padded width.  face_x/face_y: bit f set = field f is
// located at Face in that direction; grad[f][4] = west, east, south, north GradientBoundaryCondition values (NaN = default).
template <typename T>
struct HaloBc {
    T *f[4];
    T grad[4][4];
    int nf, Nx, Ny, Hx, Hy, topo_x, topo_y, face_x, face_y;
    long sy;
    T dx, dy;
};
template <typename T> hipError_t launch_fill_halo_bc(const HaloBc<T> &a, hipStream_t s);

// Arguments of the fused tendency kernels; pointers address interior cell (1,1) like OpArgs.
template <typename T>
struct TendArgs {
    const T *q1, *q2, *h, *A;   // (u|uh, v|vh, h, A)
    T *G1, *G2, *Gh, *GA;
    int Nx, Ny, Hx, Hy;
    long sy;
    T dx, dy, rdx, rdy, grav, fcor;
    int j0, j1;
    int j0b, j1b;         // optional SECOND row range served by the same launch (LDS-tiled kernel; empty when j1b <= j0b): the two
                          // boundary strips of a y-slab are one launch on the exchange's critical path instead of two
    // optional fused RK3 substep (fuse != 0):  Unew[f] = U[f] + dt (gamma G[f] + zeta Gm[f])  written to a SECOND set of
    // fields (neighbouring workgroups still read the old U through their halos); store_G = 0 skips writing G (last stage)
    int fuse, first, store_G;
    int gm_prev;          // Gm[] holds the previous STATE U- (U = U- + dt gamma- G-), zeta holds zeta/gamma-: Unew = U + dt gamma G + zeta (U - U-)
    int drop_G;           // marching kernels: issue the G stores with an out-of-range offset (the hardware drops them): lets the last RK3
                          // stage run the stage-2 kernel variant where that one has the better register allocation
    int wrap;             // periodic index wrapping of the READS: bit0 = x, bit1 = y -- the kernel takes (x mod Nx, y mod Ny) instead of the
                          // halo cells, so the caller need not have filled those halos (no halo-fill launch between RK3 stages)
    int kernel_variant;   // 0 = by size, 1 = LDS-tiled kernel, 2 = row-marching kernel
    int topo_x, topo_y;   // 0 Periodic, 1 Bounded (wall orders of the reconstructions; the LDS-tiled kernel implements them)
    int leave_room;       // marching kernels: leave ~5 % of the workgroup slots free for another stream's kernels
    int edge_cols;        // LDS-tiled kernel: only the first and the last (narrow last: last two) 64-column tile columns -- the x-wall frame of a Bounded grid
    T *Unew[4];
    const T *Gm[4];
    T dt, gamma, zeta;
    T dtg;                // dt * gamma, formed on the host (a uniform fp64 product would sit in a VGPR pair for the whole kernel)
    T cu, cg;             // the substep in coefficient form, Unew = (U + cu U) + dt gamma G + cg Gm: (0, dt zeta) for Gm = G-, (zeta', -zeta') for
                          // Gm = previous state (gm_prev).  The stage variant that reads Gm AND stores G (the second RK3 stage) uses this form,
                          // so that one compiled kernel serves both operands without a branch (a branch there cost the other variants their
                          // register allocation: 80 B of scratch).
};
template <typename T>
struct Rk3Args {
    T *U[4];
    const T *Gn[4];
    const T *Gm[4];
    int Nx;
    long sy;
    int j0, j1;
    T dt, gamma, zeta;
    int first;
};
// Memory operations of the marching kernels are STRAIGHT-LINE code: every iteration issues the same loads and the same stores.
// Lanes / iterations that own no output (strip-halo lanes, columns beyond Nx, warm-up rows) still issue their stores, with an
// out-of-range buffer offset that the hardware drops, and loads are made unconditional by clamping.  Reason: the compiler's
// s_waitcnt insertion assumes, at every control-flow join, the incoming path with the FEWEST operations in flight; with an
// `if (col_ok)` around the stores (an execz branch) or a `continue` in the warm-up rows the wait for a prefetched row
// degenerated to vmcnt(0) at the top of every iteration -- each wave also waited for the stores it had just issued and for
// the row it had prefetched one iteration ago (PF rows of prefetch were in effect none).  With a fixed pattern the exact wait
// is stated once, at the end of the iteration: only operations OLDER than the row about to enter the windows must be back.
typedef int sw_v2i __attribute__((ext_vector_type(2)));
// packed fp32: two adjacent columns of a row in one 64-bit register pair (v_pk_* arithmetic); _u = as it lies in global memory,
// where a pair is only dword-aligned (the interior starts Hx = 3 floats into a row)
typedef float sw_f2 __attribute__((ext_vector_type(2)));
typedef sw_f2 sw_f2_u __attribute__((aligned(4)));
constexpr unsigned SW_OOB = 0xFFFFFFC0u;   // >= any parent's size in bytes (the launcher keeps parents below 4 GiB - 64 B)
template <typename T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(T *parent, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(parent, 0, (int)bytes, 0x00020000);
}
template <typename T> __device__ __forceinline__ void buffer_store(T v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    if constexpr (sizeof(T) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sw_v2i, v), r, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 0);
}
// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
template <int N> __device__ __forceinline__ void wait_vmem_all_but() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// Wave priority that falls with progress ("laggards first") for the row-marching kernels.  Their grid is ONE round of resident
// workgroups, so the workgroups sharing a CU start together -- but the SIMD arbiter issues oldest-wave-first, and with equal
// priorities they finish one after the other: measured for the vector-invariant tendency kernel at 55 %, 75 % and 100 % of the
// launch (profiles/r01/workgroup_timeline.json), i.e. the last quarter of the launch ran one wave per SIMD, which cannot hide
// the ~8-cycle dependent-issue latency of fp64 (tools/valu_probe.hip).  Dropping a wave's priority at 60 / 85 / 95 % of its rows
// lets the others catch up: all workgroups finish within 10 % of each other, launch -6 %.
struct ProgressPriority {
    int t1, t2, t3;
    __device__ __forceinline__ explicit ProgressPriority(int niter)
        : t1((niter * 3) / 5), t2((niter * 17) / 20), t3((niter * 19) / 20) {
        __builtin_amdgcn_s_setprio(3);
    }
    __device__ __forceinline__ void at(int it) const {
        if (it == t1) __builtin_amdgcn_s_setprio(2);
        if (it == t2) __builtin_amdgcn_s_setprio(1);
        if (it == t3) __builtin_amdgcn_s_setprio(0);
    }
};
// (Slab driver, two streams on one chip: starting the interior launch one priority level lower and holding the boundary-zone launch
//  beside it at the top level was measured and dropped -- the ring-of-one step got 2-8 % SLOWER at every slab height: the falling
//  priorities are what keeps the interior's own workgroups in step.)

// ---- launch geometry of the row-marching kernels --------------------------------------------------------------------------
// One workgroup = a strip of nt - 2*xh output columns x LY rows; the grid is a whole number of rounds of resident workgroups.
struct MarchGeometry {
    int nt;        // threads per workgroup (strip width incl. 2*xh halo lanes)
    int nstrips, nseg, LY;
    int wg_per_cu; // resident workgroups per CU the kernel is built for
};
// Compute units of the current device (hipDeviceAttributeMultiprocessorCount; 256 on MI355X), cached per process.
inline int device_cu_count() {
    static int cus = 0;
    if (cus <= 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    return cus;
}
// Tuning knobs are read from the environment ONCE per process (not on every launch).
inline int env_knob(const char *name, int &cache) {   // cache: 0 = not read yet, -1 = unset, > 0 = value
    if (cache == 0) {
        const char *e = getenv(name);
        const int v = e ? atoi(e) : 0;
        cache = v > 0 ? v : -1;
    }
    return cache > 0 ? cache : 0;
}
// Strip width: the FIRST candidate workgroup size unless a later one covers Nx with at least 5 % fewer lanes (1024 columns: 5 strips x
// 256 lanes = 1280, but 9 x 128 = 1152).  Rows per segment: the smallest whole number of rounds of resident workgroups
// (wg_per_cu x CUs slots; ~5 % fewer with leave_room, so that another stream's kernels find room) whose segments are at most
// 128 rows, but never shorter than ly_min rows.
// Share of the workgroup slots an interior launch of the slab driver leaves free for the comm stream's kernels, in 64ths: 3 (4.7 %),
// and 1 (1.6 %) for slabs of 3072 rows and more -- the boundary work per step is fixed, a long interior launch gives it time enough in
// few slots, and every slot costs the interior launch its share of the chip (4096 x 4096 ring-of-one step: +1.4-1.9 % over the plain
// step with 1, +3.7-4.0 % with 3; 4096 x 2048: +4.8-5.2 % vs +4.6-4.7 %; three alternating runs each).  SWMHD_RING_ROOM overrides (tuning).
inline int leave_room_64ths(int rows) {
    static int cache = 0;
    const int v = env_knob("SWMHD_RING_ROOM", cache);
    return v > 0 ? (v < 32 ? v : 32) : (rows >= 3072 ? 1 : 3);
}
inline MarchGeometry march_geometry(int Nx, int rows, int xh, const int *nts, const int *wgs, int ncand, int ly_min, bool leave_room,
                                    int force_nt, int force_ly) {
    MarchGeometry g{};
    long best = -1;
    for (int k = 0; k < ncand; ++k) {
        const int txo = nts[k] - 2 * xh, ns = (Nx + txo - 1) / txo;
        const long lanes = (long)ns * nts[k];
        const bool take = force_nt ? nts[k] == force_nt : (best < 0 || lanes * 20 <= best * 19);
        if (take) { best = lanes; g.nt = nts[k]; g.nstrips = ns; g.wg_per_cu = wgs[k]; }
    }
    if (best < 0) { g.nt = nts[0]; g.nstrips = (Nx + nts[0] - 2 * xh - 1) / (nts[0] - 2 * xh); g.wg_per_cu = wgs[0]; }
    int slots = device_cu_count() * g.wg_per_cu;
    if (leave_room) slots -= (slots * leave_room_64ths(rows)) / 64;
    int LY = 32;
    for (int k = 1; k <= 64; ++k) {
        const int ns = (slots * k) / g.nstrips;
        if (ns < 1) continue;
        const int ly = (rows + ns - 1) / ns;
        if (ly <= 128) { LY = ly < ly_min ? ly_min : ly; break; }
    }
    if (force_ly > 0) LY = force_ly;
    g.LY = LY;
    g.nseg = (rows + LY - 1) / LY;
    return g;
}

template <typename T> hipError_t launch_tendency_fast(const TendArgs<T> &a, int formulation, int lorentz, hipStream_t s);
// geometry the fast tendency launcher would use (kind: 1 = LDS-tiled kernel, 2 = row-marching kernel); for bench.py's VALU floor
int tendency_launch_geometry(int Nx, int rows, int formulation, int elem_size, int kernel_variant, int leave_room, int wrap, int out[8]);
template <typename T> hipError_t launch_tendency_strict(const TendArgs<T> &a, int formulation, int lorentz, hipStream_t s);
template <typename T> hipError_t launch_rk3_substep_fast(const Rk3Args<T> &a, hipStream_t s);
// internal twin of swmhd_tendencies_rk3_* for the slab driver (ring.hip): rows [j0, j1) and [j0b, j1b) of one RK3 stage in ONE launch
// where the kernel chosen supports it (same argument checks and return codes as the exported call)
template <typename T>
int tendencies_rk3_two_ranges(const T *const *q, T *const *qnew, T *const *Gn, const T *const *Gm, int Nx, int Ny, int Hx, int Hy, long sy,
                              T dx, T dy, T grav, T fcor, int formulation, int lorentz, T dt, T gamma, T zeta, int store_G, int j0, int j1,
                              int j0b, int j1b, int flags, void *stream);
template <typename T> hipError_t launch_rk3_substep_strict(const Rk3Args<T> &a, hipStream_t s);

// energies + extrema; workspace >= SWMHD_DIAG_WORKSPACE doubles, out = 7 doubles (both device memory)
template <typename T>
hipError_t launch_diagnostics(const T *q1, const T *q2, const T *h, const T *A, int Nx, int Ny, int j0, int j1, long sy, T dx, T dy,
                              T grav, T href, int form, double *workspace, double *out, hipStream_t s);

// Periodic "gather on read": with TendArgs::wrap the tendency kernels map a halo index to its periodic image in the interior when
// they LOAD (one integer select per row / per lane, outside the arithmetic), so the state needs no halo-fill launch between RK3
// stages: 3 launches per step instead of 6 on one GPU, and no x-halo kernel in front of the ring exchange on a slab.  (Round 1 had
// the tile kernel scatter the images on write instead; that could not serve the marching kernels without costing them registers.)
__device__ __forceinline__ int wrap_index(int v, int n, int lo, int hi, bool wrap) {
    if (wrap) v = v < 0 ? v + n : (v >= n ? v - n : v);
    return v < lo ? lo : (v > hi ? hi : v);   // (lanes / rows beyond one period only feed outputs that are never stored)
}

// XCD-aware block remap (cdna_hip_programming.md T1): hardware deals consecutive block ids round-robin
// over the 8 XCDs; remapping gives each XCD (and its private 4 MiB L2) a contiguous run of tiles, so the
// halo rows/columns that y-/x-adjacent tiles share are L2 hits instead of second HBM/MALL fetches.
// Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    constexpr unsigned NXCD = 8;
    unsigned q = nblk / NXCD, r = nblk % NXCD;  // XCDs [0,r) own q+1 blocks, the rest q
    unsigned x = bid % NXCD, k = bid / NXCD;
    unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + k;
}

}  // namespace swmhd
