// C-ABI of libswmhd.so (include/swmhd.h): argument validation + kernel launches.  No torch types, no
// global state; every call only enqueues work on the caller's stream.
#include "../../include/swmhd.h"
#include "common.hpp"
#include <math.h>

using namespace swmhd;

namespace {

inline int hiprc(hipError_t e) { return e == hipSuccess ? SWMHD_OK : -(int)e; }

template <typename T>
int lorentz_common(bool divergence, const T *A, const T *h, T *Fx, T *Fy, int Nx, int Ny, int Hx, int Hy,
                   int64_t sy, T dx, T dy, int topo_x, int topo_y, int j_begin, int j_end, int flags, void *stream) {
    if (!A || !h || !Fx || !Fy) return SWMHD_EINVAL;
    if (Nx <= 0 || Ny <= 0 || Hx < 0 || Hy < 0 || sy < (int64_t)Nx + 2 * Hx) return SWMHD_EINVAL;
    if (!(dx > T(0)) || !(dy > T(0))) return SWMHD_EINVAL;
    if (j_begin < 0 || j_end > Ny || j_begin > j_end) return SWMHD_EINVAL;
    const int need = divergence ? 3 : 2;
    if (Hx < need || Hy < need) return SWMHD_EHALO;
    if ((topo_x != SWMHD_BOUNDED && topo_x != SWMHD_PERIODIC) || (topo_y != SWMHD_BOUNDED && topo_y != SWMHD_PERIODIC))
        return SWMHD_EINVAL;
    if (!divergence && (topo_x != SWMHD_PERIODIC || topo_y != SWMHD_PERIODIC)) return SWMHD_EINVAL;
    if (flags & ~(SWMHD_STRICT | SWMHD_TILE_KERNEL | SWMHD_MARCH_KERNEL)) return SWMHD_EINVAL;
    if (j_begin == j_end) return SWMHD_OK;
    OpArgs<T> a;
    a.kernel_variant = (flags & SWMHD_TILE_KERNEL) ? 1 : ((flags & SWMHD_MARCH_KERNEL) ? 2 : 0);
    const long off = (long)Hy * sy + Hx;
    a.A = A + off; a.h = h + off; a.Fx = Fx + off; a.Fy = Fy + off;
    a.Nx = Nx; a.Ny = Ny; a.Hx = Hx; a.Hy = Hy; a.sy = (long)sy;
    a.dx = dx; a.dy = dy; a.rdx = T(1) / dx; a.rdy = T(1) / dy;
    a.j0 = j_begin; a.j1 = j_end; a.topo_x = topo_x; a.topo_y = topo_y;
    a.edge_cols = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool strict = (flags & SWMHD_STRICT) != 0;
    hipError_t e;
    if (divergence) e = strict ? launch_lorentz_divergence_strict<T>(a, s) : launch_lorentz_divergence_fast<T>(a, s);
    else e = strict ? launch_lorentz_jacobian_strict<T>(a, s) : launch_lorentz_jacobian_fast<T>(a, s);
    return hiprc(e);
}

template <typename T>
int halo_common(T *f, int Nx, int Ny, int Hx, int Hy, int64_t sy, int which, void *stream) {
    if (!f || Nx <= 0 || Ny <= 0 || Hx < 0 || Hy < 0 || sy < (int64_t)Nx + 2 * Hx) return SWMHD_EINVAL;
    if (Hx > Nx || Hy > Ny) return SWMHD_EHALO;
    if (which & ~(SWMHD_HALO_X | SWMHD_HALO_Y)) return SWMHD_EINVAL;
    return hiprc(launch_fill_halo_periodic<T>(f + (long)Hy * sy + Hx, Nx, Ny, Hx, Hy, (long)sy, which, (hipStream_t)stream));
}

template <typename T>
int halo_multi_common(T *const *f, int nf, int Nx, int Ny, int Hx, int Hy, int64_t sy, int which, void *stream) {
    if (!f || nf < 1 || nf > 4 || Nx <= 0 || Ny <= 0 || Hx < 0 || Hy < 0 || sy < (int64_t)Nx + 2 * Hx) return SWMHD_EINVAL;
    if (Hx > Nx || Hy > Ny) return SWMHD_EHALO;
    if (which & ~(SWMHD_HALO_X | SWMHD_HALO_Y)) return SWMHD_EINVAL;
    T *p[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < nf; ++k) {
        if (!f[k]) return SWMHD_EINVAL;
        p[k] = f[k] + (long)Hy * sy + Hx;
    }
    return hiprc(launch_fill_halo_periodic_multi<T>(p, nf, Nx, Ny, Hx, Hy, (long)sy, which, (hipStream_t)stream));
}

template <typename T>
int halo_bc_common(T *const *f, int nf, int Nx, int Ny, int Hx, int Hy, int64_t sy, int topo_x, int topo_y, int face_x, int face_y,
                   const T *gradient, T dx, T dy, void *stream) {
    if (!f || nf < 1 || nf > 4 || Nx <= 0 || Ny <= 0 || Hx < 0 || Hy < 0 || sy < (int64_t)Nx + 2 * Hx) return SWMHD_EINVAL;
    if ((topo_x != SWMHD_BOUNDED && topo_x != SWMHD_PERIODIC) || (topo_y != SWMHD_BOUNDED && topo_y != SWMHD_PERIODIC)) return SWMHD_EINVAL;
    if (Hx > Nx || Hy > Ny) return SWMHD_EHALO;
    if ((topo_x == SWMHD_BOUNDED && Hx < 1) || (topo_y == SWMHD_BOUNDED && Hy < 1)) return SWMHD_EHALO;   // the far wall lives in the first halo line
    if (!(dx > T(0)) || !(dy > T(0))) return SWMHD_EINVAL;
    HaloBc<T> a;
    for (int k = 0; k < 4; ++k) {
        a.f[k] = nullptr;
        for (int e = 0; e < 4; ++e) a.grad[k][e] = (gradient && k < nf) ? gradient[4 * k + e] : T(NAN);
    }
    for (int k = 0; k < nf; ++k) {
        if (!f[k]) return SWMHD_EINVAL;
        a.f[k] = f[k] + (long)Hy * sy + Hx;
    }
    a.nf = nf; a.Nx = Nx; a.Ny = Ny; a.Hx = Hx; a.Hy = Hy; a.topo_x = topo_x; a.topo_y = topo_y; a.face_x = face_x; a.face_y = face_y;
    a.sy = (long)sy; a.dx = dx; a.dy = dy;
    return hiprc(launch_fill_halo_bc<T>(a, (hipStream_t)stream));
}

template <typename T>
struct FuseRk3 {
    T *Unew[4];
    const T *const *Gm;
    T dt, gamma, zeta;
    int store_G;
};

template <typename T>
int tend_common(const T *q1, const T *q2, const T *h, const T *A, T *G1, T *G2, T *Gh, T *GA, int Nx, int Ny, int Hx, int Hy,
                int64_t sy, T dx, T dy, T grav, T fcor, int formulation, int lorentz, int j0, int j1, int flags, void *stream,
                const FuseRk3<T> *rk = nullptr, int j0b = 0, int j1b = 0) {
    if (!q1 || !q2 || !h || !A || !G1 || !G2 || !Gh || !GA) return SWMHD_EINVAL;
    if (Nx <= 0 || Ny <= 0 || Hx < 0 || Hy < 0 || sy < (int64_t)Nx + 2 * Hx) return SWMHD_EINVAL;
    if (!(dx > T(0)) || !(dy > T(0))) return SWMHD_EINVAL;
    // Rows may reach up to Hy - 3 rows into the y halo (the stencil stays inside the padded array) unless y is wrapped or Bounded:
    // a slab with a deep halo evaluates its neighbours' edge rows itself instead of exchanging them every stage (ring.hip).
    const int ext = ((flags & (SWMHD_WRAP_Y | SWMHD_BOUNDED_Y)) || Hy < 3) ? 0 : Hy - 3;
    if (j0 < -ext || j1 > Ny + ext || j0 > j1) return SWMHD_EINVAL;
    if (j1b > j0b && (j0b < j1 || j1b > Ny + ext)) return SWMHD_EINVAL;   // (internal: second row range of the slab driver, above the first)
    if (flags & ~(SWMHD_STRICT | SWMHD_TILE_KERNEL | SWMHD_MARCH_KERNEL | SWMHD_WRAP_X | SWMHD_WRAP_Y | SWMHD_LEAVE_ROOM | SWMHD_BOUNDED_X | SWMHD_BOUNDED_Y | SWMHD_GM_IS_PREV_STATE)) return SWMHD_EINVAL;
    if (flags & SWMHD_GM_IS_PREV_STATE) {   // fast, periodic, fused stage with a G- operand only (a Bounded grid's frame launch would read cells the first launch has overwritten)
        if (!rk || !rk->Gm) return SWMHD_EINVAL;
        if (flags & (SWMHD_STRICT | SWMHD_BOUNDED_X | SWMHD_BOUNDED_Y)) return SWMHD_ENOTSUP;
    }
    if (((flags & SWMHD_BOUNDED_X) && (flags & SWMHD_WRAP_X)) || ((flags & SWMHD_BOUNDED_Y) && (flags & SWMHD_WRAP_Y))) return SWMHD_EINVAL;
    if ((flags & (SWMHD_BOUNDED_X | SWMHD_BOUNDED_Y)) && (flags & SWMHD_MARCH_KERNEL)) return SWMHD_ENOTSUP;   // walls: LDS-tiled kernel only
    if (((flags & SWMHD_WRAP_X) && Hx > Nx) || ((flags & SWMHD_WRAP_Y) && Hy > Ny)) return SWMHD_EHALO;   // one period must cover the halo
    if (formulation != SWMHD_CONSERVATIVE && formulation != SWMHD_VECTOR_INVARIANT) return SWMHD_EINVAL;
    // the Jacobian forcing acts on (u, v), the divergence forcing on (uh, vh)  (SWMHD_example.jl:30-31, divergence_sw_mhd.jl:28-29)
    const bool ok = lorentz == SWMHD_LORENTZ_NONE || (formulation == SWMHD_VECTOR_INVARIANT && lorentz == SWMHD_LORENTZ_JACOBIAN) ||
                    (formulation == SWMHD_CONSERVATIVE && lorentz == SWMHD_LORENTZ_DIVERGENCE);
    if (!ok) return SWMHD_EINVAL;
    if (Hx < 3 || Hy < 3) return SWMHD_EHALO;
    if (j0 == j1 && j1b <= j0b) return SWMHD_OK;
    TendArgs<T> a;
    const long off = (long)Hy * sy + Hx;
    a.q1 = q1 + off; a.q2 = q2 + off; a.h = h + off; a.A = A + off;
    a.G1 = G1 + off; a.G2 = G2 + off; a.Gh = Gh + off; a.GA = GA + off;
    a.Nx = Nx; a.Ny = Ny; a.Hx = Hx; a.Hy = Hy; a.sy = (long)sy;
    a.dx = dx; a.dy = dy; a.rdx = T(1) / dx; a.rdy = T(1) / dy; a.grav = grav; a.fcor = fcor; a.j0 = j0; a.j1 = j1;
    a.j0b = j1b > j0b ? j0b : 0; a.j1b = j1b > j0b ? j1b : 0;
    a.fuse = 0; a.first = 0; a.store_G = 1; a.drop_G = 0; a.dt = a.gamma = a.zeta = T(0);
    a.gm_prev = (flags & SWMHD_GM_IS_PREV_STATE) ? 1 : 0; a.cu = a.cg = a.dtg = T(0);
    a.wrap = ((flags & SWMHD_WRAP_X) ? 1 : 0) | ((flags & SWMHD_WRAP_Y) ? 2 : 0);
    a.leave_room = (flags & SWMHD_LEAVE_ROOM) ? 1 : 0;
    a.edge_cols = 0;
    a.topo_x = (flags & SWMHD_BOUNDED_X) ? SWMHD_BOUNDED : SWMHD_PERIODIC; a.topo_y = (flags & SWMHD_BOUNDED_Y) ? SWMHD_BOUNDED : SWMHD_PERIODIC;
    a.kernel_variant = (flags & SWMHD_TILE_KERNEL) ? 1 : ((flags & SWMHD_MARCH_KERNEL) ? 2 : 0);
    for (int f = 0; f < 4; ++f) { a.Unew[f] = nullptr; a.Gm[f] = nullptr; }
    if (rk) {
        a.fuse = 1; a.first = rk->Gm ? 0 : 1; a.store_G = rk->store_G; a.dt = rk->dt; a.gamma = rk->gamma; a.zeta = rk->zeta;
        a.cu = a.gm_prev ? rk->zeta : T(0); a.cg = a.gm_prev ? -rk->zeta : rk->dt * rk->zeta; a.dtg = rk->dt * rk->gamma;
        for (int f = 0; f < 4; ++f) { a.Unew[f] = rk->Unew[f] + off; a.Gm[f] = rk->Gm ? rk->Gm[f] + off : nullptr; }
    }
    hipStream_t s = (hipStream_t)stream;
    return hiprc((flags & SWMHD_STRICT) ? launch_tendency_strict<T>(a, formulation, lorentz, s)
                                        : launch_tendency_fast<T>(a, formulation, lorentz, s));
}

template <typename T>
int tend_rk3_common(const T *const *q, T *const *qnew, T *const *Gn, const T *const *Gm, int Nx, int Ny, int Hx, int Hy, int64_t sy,
                    T dx, T dy, T grav, T fcor, int formulation, int lorentz, T dt, T gamma, T zeta, int store_G, int j0, int j1,
                    int flags, void *stream, int j0b = 0, int j1b = 0) {
    if (!q || !qnew || !Gn) return SWMHD_EINVAL;
    for (int f = 0; f < 4; ++f) {
        if (!q[f] || !qnew[f] || !Gn[f] || (Gm && !Gm[f])) return SWMHD_EINVAL;
        for (int k = 0; k < 4; ++k)
            if (qnew[f] == q[k]) return SWMHD_EINVAL;   // the new state must not alias the state being read through halos
    }
    FuseRk3<T> rk{{qnew[0], qnew[1], qnew[2], qnew[3]}, Gm, dt, gamma, zeta, store_G ? 1 : 0};
    return tend_common<T>(q[0], q[1], q[2], q[3], Gn[0], Gn[1], Gn[2], Gn[3], Nx, Ny, Hx, Hy, sy, dx, dy, grav, fcor, formulation,
                          lorentz, j0, j1, flags, stream, &rk, j0b, j1b);
}

template <typename T>
int rk3_common(T *const *U, const T *const *Gn, const T *const *Gm, int Nx, int Ny, int Hx, int Hy, int64_t sy, T dt, T gamma,
               T zeta, int j0, int j1, int flags, void *stream) {
    if (!U || !Gn || Nx <= 0 || Ny <= 0 || Hx < 0 || Hy < 0 || sy < (int64_t)Nx + 2 * Hx) return SWMHD_EINVAL;
    if (j0 < 0 || j1 > Ny || j0 > j1 || (flags & ~SWMHD_STRICT)) return SWMHD_EINVAL;
    Rk3Args<T> a;
    const long off = (long)Hy * sy + Hx;
    for (int f = 0; f < 4; ++f) {
        if (!U[f] || !Gn[f] || (Gm && !Gm[f])) return SWMHD_EINVAL;
        a.U[f] = U[f] + off; a.Gn[f] = Gn[f] + off; a.Gm[f] = Gm ? Gm[f] + off : Gn[f] + off;
    }
    a.Nx = Nx; a.sy = (long)sy; a.j0 = j0; a.j1 = j1; a.dt = dt; a.gamma = gamma; a.zeta = zeta; a.first = Gm ? 0 : 1;
    hipStream_t s = (hipStream_t)stream;
    return hiprc((flags & SWMHD_STRICT) ? launch_rk3_substep_strict<T>(a, s) : launch_rk3_substep_fast<T>(a, s));
}

template <typename T>
int step_common(T *const *q, T *const *q_alt, T *const *Ga, T *const *Gb, int Nx, int Ny, int Hx, int Hy, int64_t sy, T dx, T dy, T grav,
                T fcor, int formulation, int lorentz, T dt, int nsteps, int flags, int *state_in_alt, void *stream) {
    if (!q || !q_alt || !Ga || !Gb || nsteps < 0) return SWMHD_EINVAL;
    if (flags & (SWMHD_BOUNDED_X | SWMHD_BOUNDED_Y)) return SWMHD_ENOTSUP;   // the driver's halo fill is the periodic one
    // RungeKutta3 coefficients (Oceananigans TimeSteppers: gamma = 8/15, 5/12, 3/4; zeta = -, -17/60, -5/12)
    const T gam[3] = {T(8.0 / 15.0), T(5.0 / 12.0), T(3.0 / 4.0)};
    const T zet[3] = {T(0), T(-17.0 / 60.0), T(-5.0 / 12.0)};
    T *cur[4], *alt[4], *gn[4], *gm[4];
    for (int f = 0; f < 4; ++f) {
        if (!q[f] || !q_alt[f] || !Ga[f] || !Gb[f]) return SWMHD_EINVAL;
        cur[f] = q[f]; alt[f] = q_alt[f]; gn[f] = Ga[f]; gm[f] = Gb[f];
    }
    int swaps = 0;
    // Fast builds: the second stage takes G- = (U1 - U0) / (dt gamma1) from the two states (SWMHD_GM_IS_PREV_STATE) -- U0 lives in the
    // very buffer the stage writes U2 to -- so the first stage stores no tendencies: 288 instead of 320 B/cell-step.
    const bool from_state = !(flags & SWMHD_STRICT);
    for (int n = 0; n < nsteps; ++n)
        for (int st = 0; st < 3; ++st) {
            const T *cq[4] = {cur[0], cur[1], cur[2], cur[3]};
            const T *cgm[4] = {gm[0], gm[1], gm[2], gm[3]};
            const bool fs = from_state && st == 1;
            if (fs) for (int f = 0; f < 4; ++f) cgm[f] = alt[f];
            const int store = st == 1 ? 1 : (st == 0 && !from_state ? 1 : 0);
            int rc = tend_rk3_common<T>(cq, alt, gn, st == 0 ? nullptr : cgm, Nx, Ny, Hx, Hy, sy, dx, dy, grav, fcor, formulation,
                                        lorentz, dt, gam[st], fs ? zet[1] / gam[0] : zet[st], store, 0, Ny,
                                        flags | (fs ? SWMHD_GM_IS_PREV_STATE : 0), stream);
            if (rc) return rc;
            for (int f = 0; f < 4; ++f) { T *t = cur[f]; cur[f] = alt[f]; alt[f] = t; t = gn[f]; gn[f] = gm[f]; gm[f] = t; }
            ++swaps;
            const int need = (SWMHD_HALO_X | SWMHD_HALO_Y) & ~(((flags & SWMHD_WRAP_X) ? SWMHD_HALO_X : 0) | ((flags & SWMHD_WRAP_Y) ? SWMHD_HALO_Y : 0));
            if (need) {   // whatever the kernel did not wrap itself
                rc = halo_multi_common<T>(cur, 4, Nx, Ny, Hx, Hy, sy, need, stream);
                if (rc) return rc;
            }
        }
    if (state_in_alt) *state_in_alt = swaps & 1;
    return SWMHD_OK;
}

template <typename T>
int diag_common(const T *q1, const T *q2, const T *h, const T *A, int Nx, int Ny, int Hx, int Hy, int64_t sy, T dx, T dy, T grav,
                T href, int form, int j0, int j1, double *ws, double *out, void *stream) {
    if (!q1 || !q2 || !h || !A || !ws || !out) return SWMHD_EINVAL;
    if (Nx <= 0 || Ny <= 0 || Hx < 0 || Hy < 0 || sy < (int64_t)Nx + 2 * Hx || !(dx > T(0)) || !(dy > T(0))) return SWMHD_EINVAL;
    if (j0 < 0 || j1 > Ny || j0 > j1) return SWMHD_EINVAL;
    if (form != SWMHD_CONSERVATIVE && form != SWMHD_VECTOR_INVARIANT) return SWMHD_EINVAL;
    if (Hx < 1 || Hy < 1) return SWMHD_EHALO;
    const long off = (long)Hy * sy + Hx;
    return hiprc(launch_diagnostics<T>(q1 + off, q2 + off, h + off, A + off, Nx, Ny, j0, j1, (long)sy, dx, dy, grav, href, form, ws,
                                       out, (hipStream_t)stream));
}

}  // namespace

namespace swmhd {
template <typename T>
int tendencies_rk3_two_ranges(const T *const *q, T *const *qnew, T *const *Gn, const T *const *Gm, int Nx, int Ny, int Hx, int Hy, long sy,
                              T dx, T dy, T grav, T fcor, int formulation, int lorentz, T dt, T gamma, T zeta, int store_G, int j0, int j1,
                              int j0b, int j1b, int flags, void *stream) {
    return tend_rk3_common<T>(q, qnew, Gn, Gm, Nx, Ny, Hx, Hy, (int64_t)sy, dx, dy, grav, fcor, formulation, lorentz, dt, gamma, zeta,
                              store_G, j0, j1, flags, stream, j0b, j1b);
}
#define SW_INST(T)                                                                                                                   \
    template int tendencies_rk3_two_ranges<T>(const T *const *, T *const *, T *const *, const T *const *, int, int, int, int, long, T, \
                                              T, T, T, int, int, T, T, T, int, int, int, int, int, int, void *);
SW_INST(double)
SW_INST(float)
#undef SW_INST
}  // namespace swmhd

extern "C" {

int swmhd_version(void) { return SWMHD_VERSION; }

const char *swmhd_strerror(int rc) {
    switch (rc) {
        case SWMHD_OK: return "success";
        case SWMHD_EINVAL: return "invalid argument (null pointer, bad extents, stride, spacing, row range or flags)";
        case SWMHD_EHALO: return "halo too small for the operator's stencil (Jacobian form needs 2, divergence form 3)";
        case SWMHD_ENOTSUP: return "not supported by this build (or: the RCCL library could not be loaded)";
        case SWMHD_ECOMM: return "RCCL reported an error (swmhd_ring_last_error has its text)";
        default: return rc < 0 ? hipGetErrorString((hipError_t)(-rc)) : "unknown swmhd error";
    }
}

int swmhd_event_create(void **event) {
    if (!event) return SWMHD_EINVAL;
    hipEvent_t e;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    if (rc != hipSuccess) return hiprc(rc);
    *event = (void *)e;
    return SWMHD_OK;
}
int swmhd_event_record(void *event, void *stream) {
    return event ? hiprc(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)) : SWMHD_EINVAL;
}
int swmhd_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!start || !stop || !ms) return SWMHD_EINVAL;
    const hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
    if (rc != hipSuccess) return hiprc(rc);
    return hiprc(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
}
int swmhd_event_destroy(void *event) { return event ? hiprc(hipEventDestroy((hipEvent_t)event)) : SWMHD_OK; }

int swmhd_tendency_launch_geometry(int Nx, int rows, int formulation, int elem_size, int flags, int out[8]) {
    if (!out || Nx <= 0 || rows <= 0 || (elem_size != 4 && elem_size != 8)) return SWMHD_EINVAL;
    if (formulation != SWMHD_CONSERVATIVE && formulation != SWMHD_VECTOR_INVARIANT) return SWMHD_EINVAL;
    const int variant = (flags & SWMHD_TILE_KERNEL) ? 1 : ((flags & SWMHD_MARCH_KERNEL) ? 2 : 0);
    return tendency_launch_geometry(Nx, rows, formulation, elem_size, (flags & SWMHD_STRICT) ? 1 : variant, (flags & SWMHD_LEAVE_ROOM) ? 1 : 0,
                                    ((flags & SWMHD_WRAP_X) ? 1 : 0) | ((flags & SWMHD_WRAP_Y) ? 2 : 0), out);
}

#define SWMHD_DEF_LORENTZ(sfx, T)                                                                                      \
    int swmhd_lorentz_jacobian_##sfx(const T *A, const T *h, T *Fx, T *Fy, int Nx, int Ny, int Hx, int Hy,            \
                                     int64_t sy, T dx, T dy, int flags, void *stream) {                                \
        return lorentz_common<T>(false, A, h, Fx, Fy, Nx, Ny, Hx, Hy, sy, dx, dy, 0, 0, 0, Ny, flags, stream);         \
    }                                                                                                                  \
    int swmhd_lorentz_jacobian_rows_##sfx(const T *A, const T *h, T *Fx, T *Fy, int Nx, int Ny, int Hx, int Hy,       \
                                          int64_t sy, T dx, T dy, int j0, int j1, int flags, void *stream) {           \
        return lorentz_common<T>(false, A, h, Fx, Fy, Nx, Ny, Hx, Hy, sy, dx, dy, 0, 0, j0, j1, flags, stream);        \
    }                                                                                                                  \
    int swmhd_lorentz_divergence_##sfx(const T *A, const T *h, T *Fx, T *Fy, int Nx, int Ny, int Hx, int Hy,          \
                                       int64_t sy, T dx, T dy, int flags, void *stream) {                              \
        return lorentz_common<T>(true, A, h, Fx, Fy, Nx, Ny, Hx, Hy, sy, dx, dy, 0, 0, 0, Ny, flags, stream);          \
    }                                                                                                                  \
    int swmhd_lorentz_divergence_rows_##sfx(const T *A, const T *h, T *Fx, T *Fy, int Nx, int Ny, int Hx, int Hy,     \
                                            int64_t sy, T dx, T dy, int tx, int ty, int j0, int j1, int flags,         \
                                            void *stream) {                                                            \
        return lorentz_common<T>(true, A, h, Fx, Fy, Nx, Ny, Hx, Hy, sy, dx, dy, tx, ty, j0, j1, flags, stream);       \
    }                                                                                                                  \
    int swmhd_fill_halo_periodic_##sfx(T *f, int Nx, int Ny, int Hx, int Hy, int64_t sy, int which, void *stream) {    \
        return halo_common<T>(f, Nx, Ny, Hx, Hy, sy, which, stream);                                                   \
    }                                                                                                                  \
    int swmhd_fill_halo_##sfx(T *const *f, int nf, int Nx, int Ny, int Hx, int Hy, int64_t sy, int topo_x, int topo_y,  \
                              int face_x, int face_y, const T *gradient, T dx, T dy, void *stream) {                       \
        return halo_bc_common<T>(f, nf, Nx, Ny, Hx, Hy, sy, topo_x, topo_y, face_x, face_y, gradient, dx, dy, stream);     \
    }                                                                                                                  \
    int swmhd_fill_halo_periodic_multi_##sfx(T *const *f, int nf, int Nx, int Ny, int Hx, int Hy, int64_t sy,          \
                                             int which, void *stream) {                                                \
        return halo_multi_common<T>(f, nf, Nx, Ny, Hx, Hy, sy, which, stream);                                         \
    }                                                                                                                  \
    int swmhd_tendencies_##sfx(const T *q1, const T *q2, const T *h, const T *A, T *G1, T *G2, T *Gh, T *GA, int Nx,   \
                               int Ny, int Hx, int Hy, int64_t sy, T dx, T dy, T g, T f, int formulation, int lorentz, \
                               int j0, int j1, int flags, void *stream) {                                              \
        return tend_common<T>(q1, q2, h, A, G1, G2, Gh, GA, Nx, Ny, Hx, Hy, sy, dx, dy, g, f, formulation, lorentz,    \
                              j0, j1, flags, stream);                                                                  \
    }                                                                                                                  \
    int swmhd_tendencies_rk3_##sfx(const T *const *q, T *const *qnew, T *const *Gn, const T *const *Gm, int Nx, int Ny, \
                                   int Hx, int Hy, int64_t sy, T dx, T dy, T g, T f, int formulation, int lorentz, T dt, \
                                   T gamma, T zeta, int store_G, int j0, int j1, int flags, void *stream) {              \
        return tend_rk3_common<T>(q, qnew, Gn, Gm, Nx, Ny, Hx, Hy, sy, dx, dy, g, f, formulation, lorentz, dt, gamma,   \
                                  zeta, store_G, j0, j1, flags, stream);                                                \
    }                                                                                                                  \
    int swmhd_diagnostics_##sfx(const T *q1, const T *q2, const T *h, const T *A, int Nx, int Ny, int Hx, int Hy,      \
                                int64_t sy, T dx, T dy, T g, T href, int form, int j0, int j1, double *ws, double *out, \
                                void *stream) {                                                                        \
        return diag_common<T>(q1, q2, h, A, Nx, Ny, Hx, Hy, sy, dx, dy, g, href, form, j0, j1, ws, out, stream);       \
    }                                                                                                                  \
    int swmhd_step_rk3_##sfx(T *const *q, T *const *q_alt, T *const *Ga, T *const *Gb, int Nx, int Ny, int Hx, int Hy,   \
                             int64_t sy, T dx, T dy, T g, T f, int formulation, int lorentz, T dt, int nsteps, int flags,\
                             int *state_in_alt, void *stream) {                                                        \
        return step_common<T>(q, q_alt, Ga, Gb, Nx, Ny, Hx, Hy, sy, dx, dy, g, f, formulation, lorentz, dt, nsteps, flags, \
                              state_in_alt, stream);                                                                   \
    }                                                                                                                  \
    int swmhd_rk3_substep_##sfx(T *const *U, const T *const *Gn, const T *const *Gm, int Nx, int Ny, int Hx, int Hy,   \
                                int64_t sy, T dt, T gamma, T zeta, int j0, int j1, int flags, void *stream) {          \
        return rk3_common<T>(U, Gn, Gm, Nx, Ny, Hx, Hy, sy, dt, gamma, zeta, j0, j1, flags, stream);                   \
    }

SWMHD_DEF_LORENTZ(f64, double)
SWMHD_DEF_LORENTZ(f32, float)

}  // extern "C"
