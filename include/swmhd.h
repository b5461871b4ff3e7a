/*
 * swmhd.h -- C-ABI of libswmhd.so, the MI355X (gfx950) shallow-water-MHD tendency engine.
 *
 * This is the drop-in boundary for the hot path of writingindy/SWMHD.  The
 * reference injects its MHD physics into Oceananigans' ShallowWaterModel as per-cell callbacks
 *
 *     Forcing(lorentz_force_func_x, discrete_form = true)     jacobian_formulation/SWMHD_example.jl:30-31
 *     Forcing(div_lorentz_x,        discrete_form = true)     divergence_formulation/divergence_sw_mhd.jl:28-29
 *
 * with signature  func(i, j, k, grid, clock, fields)::FT .  A per-cell callback cannot launch a kernel, so
 * every entry point here is the *whole-field* form of one such callback (or of the Oceananigans tendency
 * kernel that calls it): it evaluates the callback for all i = 1:Nx, j = j_begin+1:j_end (k = 1) and writes
 * the result at the same (i,j) of an output field.  INTEGRATION.md shows the Julia `ccall` glue.
 *
 * Conventions (identical to what Oceananigans hands the reference):
 *   - Every field pointer is the address of the FIRST element of the halo-padded PARENT array, i.e. Julia's
 *     `pointer(parent(field))`: shape (Nx+2Hx, Ny+2Hy, 1) column-major, x fastest.  Interior element
 *     (i,j), 1-based, lives at  ptr[(j-1+Hy)*stride_y + (i-1+Hx)].  stride_y >= Nx+2Hx (elements).
 *   - Locations follow the staggered C-grid: Fx / u / uh are at (Face,Center), Fy / v / vh at
 *     (Center,Face), h / A at (Center,Center); face i lies between centres i-1 and i.
 *   - Halos of the INPUT fields must be filled by the caller before the call (Oceananigans guarantees
 *     this; swmhd_fill_halo_* does it for the stand-alone engine).  Halos of outputs are never written.
 *   - All pointers are DEVICE pointers, borrowed for the duration of the enqueued work, never freed or
 *     retained.  Calls only enqueue on `stream` (a hipStream_t passed as void*, NULL = default stream)
 *     and return immediately.  No global state: thread-safe per stream.
 *   - Return value: 0 on success, a positive SWMHD_E* code for argument errors, or a negative hipError_t
 *     (negated) if the launch failed.  Nothing throws or aborts.
 *
 * `flags`:
 *   SWMHD_STRICT  evaluate in the reference's exact operation order with IEEE divides and no FMA
 *                 contraction: bit-identical to the CPU oracle (oracle/) and hence to a non-fusing
 *                 Julia evaluation of the same scheme.  Default (0) is the fast path: reciprocal
 *                 multiplies + FMA, within the tolerances below.
 *
 * Tolerances of the fast path (asserted in tests/, achieved values in profiles/r03/fullsize_parity.json):
 *   Lorentz operators (swmhd_lorentz_*):  max|dF| <= 1e-13 * max|F| (fp64),  <= 2e-5 * max|F| (fp32), against strict.
 *   Tendency entry points (swmhd_tendencies*, swmhd_step_rk3*, swmhd_ring_step_rk3*), per tendency G of one evaluation:
 *       max|dG| <= tol * max(max|G|, S),   tol = 1e-13 (fp64; 1e-12 on rough random fields), 1e-4 (fp32),
 *     where S is the magnitude of the LARGEST TERM summed into G -- rounding errors scale with the terms, not with their sum:
 *       vector-invariant u, v :  max( (|u|+|v|) max|zeta|,  (u^2+v^2)/2 (1/dx+1/dy),  g max|h| (1/dx+1/dy),  f (|u|+|v|),  max|F_Lorentz| )
 *       conservative uh, vh   :  max( (|uh|+|vh|)(|u|+|v|)(1/dx+1/dy),  g max|h|^2/2 (1/dx+1/dy),  f (|uh|+|vh|),  max|F_Lorentz| )
 *       h                     :  (|u|/dx + |v|/dy) max|h|        (conservative: |uh|/dx + |vh|/dy)
 *       A                     :  (|u|/dx + |v|/dy) max|A|
 *     (maxima over the field; tests/test_fullsize_gpu.py::term_scales).  Achieved on MI355X at every BASELINE size: <= 5e-15 of
 *     max(max|G|, S) in fp64, <= 5e-7 in fp32.  Measured against max|G| ALONE the same errors are as large as 2e-9 (fp64: config 2's
 *     vh, 7.9e-10 for config 3's h) because at those resolutions the tendency is orders of magnitude smaller than the terms it is the
 *     difference of.  fp32 CAVEAT: where the terms cancel strongly the fp32 tendency carries no significant digits relative to its own
 *     size -- G_h of the 16384^2 Bickley-jet state (mass fluxes u h/dx ~ 650, G_h ~ 3e-4) is off by 0.15 * max|G_h| in fp32, fast or
 *     strict alike (one ulp of a flux).  The Jacobian-form force takes THIRD differences of A: on a field with a large smooth part
 *     (A = -0.05 y + perturbation) their fp32 rounding noise grows like eps |A| / dx^3 and stirs the flow once dx <~ 0.08 (the
 *     reference's 128^2 low_B_low_U run: energy error 3.9 instead of 0.53 by t = 15 in Float32; the divergence form, first differences
 *     only, is unaffected).  Use fp32 where 1e-7 of the fluxes is enough; the fp32-vs-fp64 state after 100 steps of that
 *     configuration agrees to 9e-5 of the velocity scale (tests/test_fullsize_gpu.py).
 *   Which scheme: the oracle's base right-hand side restates the Oceananigans version the reference ran and is pinned to the reference's
 *     twelve committed energy plots at plot-reading accuracy (tests/test_reference_plots.py); last bits of a Julia
 *     run are not pinned.
 */
#ifndef SWMHD_H
#define SWMHD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWMHD_VERSION 300

/* flags (bit mask) */
#define SWMHD_FAST 0
#define SWMHD_STRICT 1        /* oracle operation order, IEEE divides, no FMA contraction (bit parity); see above                      */
#define SWMHD_TILE_KERNEL 2   /* force the LDS-tiled kernel.  Default (neither bit): chosen by size -- tiles for small grids and thin  */
#define SWMHD_MARCH_KERNEL 4  /* force the row-marching kernel.   strips, row-marching from ~0.3-2 Mcell up (per entry point)          */
                              /* (8 is unassigned: it selected an experimental kernel that was removed)                               */
#define SWMHD_WRAP_X 16       /* tendency entry points: READ the inputs with periodic index wrapping in x / in y -- cell (x mod Nx,    */
#define SWMHD_WRAP_Y 32       /*   y mod Ny) instead of the halo cell -- so those halos need not be filled (no halo launch per stage)  */
#define SWMHD_BOUNDED_X 256    /* tendency entry points: the grid's topology in x / in y is Bounded (default: Periodic).  Reconstructions  */
#define SWMHD_BOUNDED_Y 512    /*   near the walls use Oceananigans' boundary schemes, the divergence forcing the reference's wall branches */
#define SWMHD_GM_IS_PREV_STATE 1024 /* swmhd_tendencies_rk3 only (fast builds, periodic): Gm[f] holds the PREVIOUS STATE U- the current state was
                                    stepped from with G- alone (U = U- + dt gamma- G-), not G- itself; pass zeta / gamma- as `zeta`; the kernel
                                    forms Unew = U + dt gamma G + zeta (U - U-).  Gm may then alias qnew (each cell reads its own U- before it
                                    writes Unew): the first RK3 stage need not store its tendencies at all (32 B/cell less HBM traffic per
                                    step).  Results differ from the G- form by less than one ulp of U.  The step drivers use it.            */
#define SWMHD_LEAVE_ROOM 64   /* tendency entry points: size the row-marching grid ~5 % short of filling the chip, so that kernels of
                                 another stream (the ring's halo exchange and boundary strips) can start while it runs                */

/* topology codes (Oceananigans.Grids.topology) */
#define SWMHD_PERIODIC 0
#define SWMHD_BOUNDED 1

/* error codes */
#define SWMHD_OK 0
#define SWMHD_EINVAL 1     /* bad extents / null pointer / bad row range */
#define SWMHD_EHALO 2      /* halo too small for the operator's stencil  */
#define SWMHD_ENOTSUP 3    /* valid request this build does not implement (or: the RCCL library could not be loaded) */
#define SWMHD_ECOMM 4      /* RCCL reported an error: swmhd_ring_last_error() has its text */

int swmhd_version(void);
/* Human-readable text for a return code of any entry point (static storage). */
const char *swmhd_strerror(int rc);

/* Timing events for benchmarks (no counterpart in the reference: its timings are wall-clock, SWMHD_example.jl:87-92).  HIP events
 * created with hipEventDisableSystemFence: a default event performs a system-scope fence when it is recorded, which costs 1-1.5 %
 * of a 4096^2 RK3 step with one pair per stage launch and 16 % when another stream's kernels run beside the timed launch.
 * swmhd_event_elapsed_ms waits for `stop`. */
int swmhd_event_create(void **event);
int swmhd_event_record(void *event, void *stream);
int swmhd_event_elapsed_ms(void *start, void *stop, float *ms);
int swmhd_event_destroy(void *event);
/* Measurement hook: nanoseconds one fp64 wave-instruction occupies a SIMD right now (three waves per SIMD, independent fma chains:
 * 4 cycles of the shader clock the box holds under fp64 load).  Synchronises `stream`.  scratch: >= 8 bytes of device memory. */
int swmhd_probe_fp64_issue(double *scratch, float *ns_per_wave_instruction, void *stream);
/* Measurement hook: best-case plain-copy rate of the box, GB/s of bytes read + written: one 16-byte element per thread, workgroups in
 * address order (the pattern that reaches the guide's 6.29 TB/s; persistent copy kernels -- torch's, hipMemcpy -- reach ~20 % less).
 * bytes: multiple of 16, >= 4096; `reps` timed launches after 3 warm-up ones.  Synchronises `stream`. */
int swmhd_probe_copy(void *dst, const void *src, size_t bytes, int reps, float *gbytes_per_s, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Jacobian-form Lorentz force.   Replaces lorentz_force_func_x / lorentz_force_func_y
 * (jacobian_formulation/sw_mhd_jacobian_functions.jl:20-26, built on Bx/By :1-7 and jacobian_x/y :10-18):
 *     Fx(i,j) = (1/ℑxᶠᵃᵃ h) * jacobian_x(i,j,A,h)    forcing on u (fcc)
 *     Fy(i,j) = (1/ℑyᵃᶠᵃ h) * jacobian_y(i,j,A,h)    forcing on v (cfc)
 * Stencil: A +-2 (cross-shaped), h +-1  ->  requires Hx,Hy >= 2.
 * Rows j = j_begin+1 .. j_end (1-based) are computed; the plain form computes all rows.
 * ---------------------------------------------------------------------------------------------- */
int swmhd_lorentz_jacobian_f64(const double *A, const double *h, double *Fx, double *Fy,
                               int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                               double dx, double dy, int flags, void *stream);
int swmhd_lorentz_jacobian_f32(const float *A, const float *h, float *Fx, float *Fy,
                               int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                               float dx, float dy, int flags, void *stream);
int swmhd_lorentz_jacobian_rows_f64(const double *A, const double *h, double *Fx, double *Fy,
                                    int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                    double dx, double dy, int j_begin, int j_end, int flags, void *stream);
int swmhd_lorentz_jacobian_rows_f32(const float *A, const float *h, float *Fx, float *Fy,
                                    int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                    float dx, float dy, int j_begin, int j_end, int flags, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Divergence-form (Maxwell-stress) Lorentz force.   Replaces div_lorentz_x / div_lorentz_y
 * (divergence_formulation/sw_mhd_divergence_functions.jl:162-170, built on upwind_biased_product :3,
 *  the 3rd-order biased interpolants :25-35, the four advective_lorentz_flux_* :38-132 and the
 *  face-located Bx/By/hBx/hBy :134-148):
 *     Fx(i,j) = (1/Az) * (δxᶠᵃᵃ lorentz_flux_hBx_bx + δyᵃᶜᵃ lorentz_flux_hBy_bx)   forcing on uh (fcc)
 *     Fy(i,j) = (1/Az) * (δxᶜᵃᵃ lorentz_flux_hBx_by + δyᵃᶠᵃ lorentz_flux_hBy_by)   forcing on vh (cfc)
 * Stencil: A +-3, h +-3  ->  requires Hx,Hy >= 3.
 * topo_x/topo_y select the reference's `topology(grid, d) == Bounded` wall branches (:42-53 etc.).
 * ---------------------------------------------------------------------------------------------- */
int swmhd_lorentz_divergence_f64(const double *A, const double *h, double *Fx, double *Fy,
                                 int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                 double dx, double dy, int flags, void *stream);
int swmhd_lorentz_divergence_f32(const float *A, const float *h, float *Fx, float *Fy,
                                 int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                 float dx, float dy, int flags, void *stream);
int swmhd_lorentz_divergence_rows_f64(const double *A, const double *h, double *Fx, double *Fy,
                                      int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                      double dx, double dy, int topo_x, int topo_y,
                                      int j_begin, int j_end, int flags, void *stream);
int swmhd_lorentz_divergence_rows_f32(const float *A, const float *h, float *Fx, float *Fy,
                                      int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                      float dx, float dy, int topo_x, int topo_y,
                                      int j_begin, int j_end, int flags, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Periodic halo fill (what Oceananigans' fill_halo_regions! does for topology = (Periodic, Periodic, Flat),
 * SWMHD_example.jl:16): copies the Hx / Hy interior edge columns / rows into the opposite halos, corners
 * included (x first, then y over the full padded width).  which: bit0 = x halos, bit1 = y halos.
 * ---------------------------------------------------------------------------------------------- */
#define SWMHD_HALO_X 1
#define SWMHD_HALO_Y 2
int swmhd_fill_halo_periodic_f64(double *field, int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                 int which, void *stream);
int swmhd_fill_halo_periodic_f32(float *field, int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                                 int which, void *stream);

int swmhd_fill_halo_periodic_multi_f64(double *const *fields, int nfields, int Nx, int Ny, int Hx, int Hy,
                                       int64_t stride_y, int which, void *stream);
int swmhd_fill_halo_periodic_multi_f32(float *const *fields, int nfields, int Nx, int Ny, int Hx, int Hy,
                                       int64_t stride_y, int which, void *stream);

/* ------------------------------------------------------------------------------------------------
 * fill_halo_regions! for any pair of (Periodic | Bounded) topologies: what the models need between RK3 stages when a direction is
 * Bounded -- the reference's scripts are written for it (Grids.topology tests throughout sw_mhd_divergence_functions.jl:42-125;
 * the commented boundary conditions SWMHD_example.jl:18-19 / divergence_sw_mhd.jl:17 put GradientBoundaryCondition(-0.05) on the
 * north and south sides of A).  Periodic directions: as swmhd_fill_halo_periodic.  Bounded directions (Oceananigans' defaults):
 *   field at Center in that direction : no-flux -- halo point m mirrors interior point m;  with a gradient value g (not NaN) instead:
 *                                       the first halo point c[0] = c[1] - g*d, c[N+1] = c[N] + g*d (all the library fills)
 *   field at Face in that direction   : impenetrable wall -- c[1] = 0 and c[N+1] = 0 (the first halo line holds the far wall)
 * Halo points the library never writes (points 2..H of a gradient side, everything beyond a wall of a face field) stay at their
 * allocation zeros for a whole Oceananigans run; this call sets them to zero, so nothing depends on stale memory.
 * West/east first, then south/north over the padded width.  fields: HOST array of nf (1..4) parents; face_x / face_y: bit f set =
 * field f is located at Face in x / y; gradient: HOST array of 4*nf values (west, east, south, north per field; NaN = default),
 * or NULL.  Two launches.
 * ---------------------------------------------------------------------------------------------- */
int swmhd_fill_halo_f64(double *const *fields, int nf, int Nx, int Ny, int Hx, int Hy, int64_t stride_y, int topo_x, int topo_y,
                        int face_x, int face_y, const double *gradient, double dx, double dy, void *stream);
int swmhd_fill_halo_f32(float *const *fields, int nf, int Nx, int Ny, int Hx, int Hy, int64_t stride_y, int topo_x, int topo_y,
                        int face_x, int face_y, const float *gradient, float dx, float dy, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused tendency evaluation: the whole-field form of Oceananigans' ShallowWaterModel tendency kernels
 * (calculate_tendencies!) WITH the reference's forcing callback fused in -- one pass over the four
 * prognostic fields produces all four tendencies; the Lorentz force never round-trips through HBM.
 *   formulation SWMHD_VECTOR_INVARIANT: (q1,q2) = (u,v), lorentz SWMHD_LORENTZ_JACOBIAN (or NONE)
 *       model configuration of jacobian_formulation/SWMHD_example.jl:21-33
 *       (WENO5(vector_invariant = VelocityStencil()), mass/tracer WENO5, g, FPlane f, tracer A)
 *   formulation SWMHD_CONSERVATIVE:     (q1,q2) = (uh,vh), lorentz SWMHD_LORENTZ_DIVERGENCE (or NONE)
 *       model configuration of divergence_formulation/divergence_sw_mhd.jl:19-31
 * The base right-hand side is Oceananigans' (third-party, un-vendored by the reference): it is restated from
 * the library's published scheme, parity UNPINNED; the forcing is the reference's own.
 * All fields need halo >= 3, filled (swmhd_fill_halo for Bounded directions).  Rows j_begin+1..j_end are computed; with a deeper
 * y halo (Hy > 3, y neither wrapped nor Bounded) the range may reach Hy - 3 rows beyond the interior on either side
 * (-(Hy-3) <= j_begin, j_end <= Ny + Hy - 3): those halo rows then receive the tendency / new state their periodic image or
 * neighbouring slab gets -- the slab driver's deep-halo schedule (swmhd_ring_step_rk3) exchanges once per step that way.
 * flags SWMHD_BOUNDED_X / SWMHD_BOUNDED_Y: that direction is Bounded -- within the boundary buffers WENO5 drops to third- and
 * first-order upwind and the centred fourth-order advecting velocity to second order (Oceananigans' topologically conditional
 * interpolation, restated; parity UNPINNED), the divergence forcing takes the reference's wall branches, and the tendency of the
 * wall-normal velocity ON the wall (index 1) is whatever the stencil gives: the caller's halo fill resets that line to zero, as
 * Oceananigans' does.  Bounded grids run on the LDS-tiled kernel; from ~0.3 Mcell on, the row-marching kernel computes every row with
 * the periodic formulas first and the LDS-tiled kernel then overwrites the frame of cells near the walls (same results to rounding).
 * A direction cannot be both Bounded and SWMHD_WRAP-ped.
 * ---------------------------------------------------------------------------------------------- */
#define SWMHD_CONSERVATIVE 0
#define SWMHD_VECTOR_INVARIANT 1
#define SWMHD_LORENTZ_NONE 0
#define SWMHD_LORENTZ_JACOBIAN 1
#define SWMHD_LORENTZ_DIVERGENCE 2
int swmhd_tendencies_f64(const double *q1, const double *q2, const double *h, const double *A,
                         double *G1, double *G2, double *Gh, double *GA,
                         int Nx, int Ny, int Hx, int Hy, int64_t stride_y, double dx, double dy,
                         double g, double f, int formulation, int lorentz,
                         int j_begin, int j_end, int flags, void *stream);
int swmhd_tendencies_f32(const float *q1, const float *q2, const float *h, const float *A,
                         float *G1, float *G2, float *Gh, float *GA,
                         int Nx, int Ny, int Hx, int Hy, int64_t stride_y, float dx, float dy,
                         float g, float f, int formulation, int lorentz,
                         int j_begin, int j_end, int flags, void *stream);

/* ------------------------------------------------------------------------------------------------
 * RK3 substep (Oceananigans TimeSteppers rk3_substep!, `timestepper = :RungeKutta3`, SWMHD_example.jl:23):
 *     U[f] += dt * (gamma * Gn[f] + zeta * Gm[f])    for the four prognostic fields f, interior rows only
 * Gm == NULL selects the first-stage form  U += dt * gamma * Gn.  U, Gn, Gm are HOST arrays of 4 device
 * pointers (parents).  (gamma, zeta) = (8/15, -), (5/12, -17/60), (3/4, -5/12).
 * ---------------------------------------------------------------------------------------------- */
int swmhd_rk3_substep_f64(double *const *U, const double *const *Gn, const double *const *Gm,
                          int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                          double dt, double gamma, double zeta, int j_begin, int j_end, int flags, void *stream);
int swmhd_rk3_substep_f32(float *const *U, const float *const *Gn, const float *const *Gm,
                          int Nx, int Ny, int Hx, int Hy, int64_t stride_y,
                          float dt, float gamma, float zeta, int j_begin, int j_end, int flags, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused RK3 stage = calculate_tendencies! + rk3_substep! in ONE pass (the old state of a cell is already on chip
 * when its tendency is known):
 *     Gn[f]   = tendency of field f evaluated from q[0..3]                     (written iff store_G != 0)
 *     qnew[f] = q[f] + dt * (gamma * Gn[f] + zeta * Gm[f])                     (Gm == NULL: first-stage form)
 * q, qnew, Gn, Gm are HOST arrays of 4 device pointers (parents) in the order (u|uh, v|vh, h, A).  qnew must not
 * alias q: neighbouring workgroups still read the old state through their halos (ping-pong the two sets).
 * Only the interior of qnew is written.  Periodic grids have two ways to give the next stage its halo values: fill qnew's halos
 * (swmhd_fill_halo_periodic_multi), or pass SWMHD_WRAP_X / SWMHD_WRAP_Y to the NEXT call, which then reads (x mod Nx, y mod Ny)
 * instead of the halo cells (needs Nx >= Hx, Ny >= Hy; every tendency kernel implements it, at no measurable cost).  The flags
 * also apply to swmhd_tendencies.
 * The last stage of a step may pass
 * store_G = 0 (the next step's first stage has zeta = 0 and never reads it).
 * ---------------------------------------------------------------------------------------------- */
int swmhd_tendencies_rk3_f64(const double *const *q, double *const *qnew, double *const *Gn, const double *const *Gm,
                             int Nx, int Ny, int Hx, int Hy, int64_t stride_y, double dx, double dy,
                             double g, double f, int formulation, int lorentz,
                             double dt, double gamma, double zeta, int store_G,
                             int j_begin, int j_end, int flags, void *stream);
int swmhd_tendencies_rk3_f32(const float *const *q, float *const *qnew, float *const *Gn, const float *const *Gm,
                             int Nx, int Ny, int Hx, int Hy, int64_t stride_y, float dx, float dy,
                             float g, float f, int formulation, int lorentz,
                             float dt, float gamma, float zeta, int store_G,
                             int j_begin, int j_end, int flags, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Native step driver: `nsteps` complete RK3 time steps of the periodic single-GPU model, i.e. Oceananigans'
 * time_step!(model, dt) (timestepper = :RungeKutta3, SWMHD_example.jl:23,42 / divergence_sw_mhd.jl:20,39) repeated:
 *     3 x { swmhd_tendencies_rk3 (gamma, zeta of the stage) ; swap state sets ; swap G sets ; periodic halo fill }
 * Periodic grids only (SWMHD_BOUNDED_* -> SWMHD_ENOTSUP: drive the stages with swmhd_tendencies_rk3 + swmhd_fill_halo).
 * All 6*nsteps launches are enqueued on `stream` by this one call (capturable into a HIP graph).  With SWMHD_WRAP_X | SWMHD_WRAP_Y
 * the halo fills are dropped (3 launches per step) and the halos of the returned state are STALE: fill them
 * (swmhd_fill_halo_periodic_multi) before anything but a WRAP-flagged tendency call reads them.
 *   q      HOST array of 4 parents (u|uh, v|vh, h, A): the current state, halos filled
 *   q_alt  second set of 4 parents (scratch on entry)
 *   Ga,Gb  two sets of 4 tendency parents (scratch)
 * On return the state is in q if *state_in_alt == 0, in q_alt otherwise (an RK3 step swaps the sets three times, so
 * nsteps odd <=> state_in_alt = 1).  state_in_alt may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int swmhd_step_rk3_f64(double *const *q, double *const *q_alt, double *const *Ga, double *const *Gb,
                       int Nx, int Ny, int Hx, int Hy, int64_t stride_y, double dx, double dy,
                       double g, double f, int formulation, int lorentz, double dt, int nsteps,
                       int flags, int *state_in_alt, void *stream);
int swmhd_step_rk3_f32(float *const *q, float *const *q_alt, float *const *Ga, float *const *Gb,
                       int Nx, int Ny, int Hx, int Hy, int64_t stride_y, float dx, float dy,
                       float g, float f, int formulation, int lorentz, float dt, int nsteps,
                       int flags, int *state_in_alt, void *stream);

/* Launch geometry the fast tendency entry points use for an Nx x rows launch on the current device (introspection for
 * benchmarks: bench.py derives the kernel's fp64-VALU floor from it; nothing in the reference corresponds to it).
 * out[0] = kernel kind (1 LDS-tiled, 2 row-marching, 3 row-marching with two fp32 columns per lane), out[1] = threads per workgroup, out[2] = strips (workgroups along x),
 * out[3] = segments (workgroups along y), out[4] = rows per segment, out[5] = resident workgroups per CU the kernel is built
 * for, out[6] = halo lanes per strip side, out[7] = compute units of the device.  elem_size 8 (f64) or 4 (f32); flags as above. */
int swmhd_tendency_launch_geometry(int Nx, int rows, int formulation, int elem_size, int flags, int out[8]);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, the domain cut into y-slabs (rank r owns global rows [r*Ny, (r+1)*Ny), all x).
 * The reference is single-process -- its periodic y boundary is the in-memory halo copy of Oceananigans'
 * fill_halo_regions! (topology = (Periodic, Periodic, Flat), SWMHD_example.jl:16, divergence_sw_mhd.jl:14); here that
 * copy becomes a ring of RCCL sends/receives between y-neighbours.  A swmhd_ring owns the RCCL
 * communicator, a high-priority comm stream and two events.  RCCL is loaded at run time from `rccl_path` (NULL/"" =
 * "librccl.so.1" from the loader path); a host that already uses RCCL (PyTorch) passes the path of ITS copy so that one
 * instance serves both.  The 128-byte id from swmhd_ring_unique_id (call on ONE rank) must reach every rank out of band
 * (MPI_Bcast, torch.distributed.broadcast, a file ...) before swmhd_ring_create, which is collective over the ranks and
 * binds the communicator to the calling thread's current HIP device.
 * ---------------------------------------------------------------------------------------------- */
#define SWMHD_RING_ID_BYTES 128
typedef struct swmhd_ring swmhd_ring;
/* SWMHD_OK if the RCCL library can be loaded from rccl_path (SWMHD_ENOTSUP otherwise).  Creates nothing: lets the ranks agree that
 * every one of them can take the native path BEFORE any of them enters the collective swmhd_ring_create. */
int swmhd_ring_available(const char *rccl_path);
int swmhd_ring_unique_id(const char *rccl_path, void *id128);
int swmhd_ring_create(swmhd_ring **ring, const char *rccl_path, int nranks, int rank, const void *id128);
/* Loopback transport for rehearsals and tests on ONE GPU (RCCL refuses two ranks on one device): creates `nranks` rings in this
 * process, rings[k] = rank k with neighbours (k-1, k+1) mod nranks, whose exchanges copy the neighbours' edge rows device-to-device
 * on the comm streams with RCCL's rendezvous semantics (a receive waits for the sender's matching exchange, a send for the
 * receiver to have taken the rows).  Everything else -- swmhd_ring_step_rk3's two schedules, streams, events -- is the code the
 * RCCL ring runs.  Drive every ring from its OWN host thread (an exchange blocks on the host until the neighbours have enqueued
 * theirs, at most `timeout_s` seconds (<= 0: 60), then SWMHD_ECOMM).  Each ring is destroyed with swmhd_ring_destroy. */
int swmhd_ring_create_loopback(swmhd_ring **rings, int nranks, double timeout_s);
int swmhd_ring_destroy(swmhd_ring *ring);
const char *swmhd_ring_last_error(const swmhd_ring *ring);
void *swmhd_ring_comm_stream(const swmhd_ring *ring);   /* the ring's hipStream_t */

/* Fill the south/north halo rows of `nfields` parents from the ring neighbours: per field the Hy northern interior rows
 * go to the north neighbour's south halo and the Hy southern interior rows to the south neighbour's north halo, full padded
 * width (fill x halos first: corners travel with the rows).  Zero-copy (rows are contiguous), one grouped RCCL launch,
 * enqueued on `stream`.  With nranks == 1 every send goes to self: a periodic copy through RCCL. */
int swmhd_ring_exchange_y_f64(swmhd_ring *ring, double *const *fields, int nfields, int Nx, int Ny, int Hx, int Hy,
                              int64_t stride_y, void *stream);
int swmhd_ring_exchange_y_f32(swmhd_ring *ring, float *const *fields, int nfields, int Nx, int Ny, int Hx, int Hy,
                              int64_t stride_y, void *stream);

/* swmhd_step_rk3 for one slab of the ring: `nsteps` RK3 steps, every launch enqueued by this one call.  Per stage the
 * interior rows [Hy, Ny-Hy) run on `stream` while the neighbour exchange of the state they read is still in flight on
 * the ring's comm stream; the two Hy-row boundary strips are queued on the comm stream behind that exchange; then the
 * x halos of the new state are filled and its exchange is started.
 *   entry: halos of q current (x and y) -- or the exchange this ring left in flight for exactly this state
 *   exit : the y exchange of the final state is IN FLIGHT on the comm stream: call swmhd_ring_join(ring, stream) before
 *          anything but swmhd_ring_step_rk3 reads the y halos (or reuses the buffers) on `stream`
 * flags must not carry SWMHD_WRAP_Y (y images belong to the neighbours); with SWMHD_WRAP_X no x-halo kernel runs between a stage and
 * its exchange and the x halos of the returned state are stale.  Ny >= 2*Hy+1.  Other arguments and state_in_alt as swmhd_step_rk3.
 * Deep-halo schedule (taken when Hy >= 9, Ny >= 32 and SWMHD_WRAP_X is set): ONE exchange of Hy rows per step instead of one of 3
 * rows per stage.  The slab evaluates the rows of its neighbours that stages 2 and 3 need inside its own halo (18 redundant rows
 * per step): stage k computes interior rows [3,9,12][k] .. Ny - [3,9,12][k] on `stream` and the boundary zones
 * [-6,-3,0][k] .. [3,9,12][k] (and the mirror image at the top) on the comm stream behind the exchange.  `stream` waits for the comm
 * stream once per step instead of once per stage, and a thin slab is bound by its interior launches rather than by the chain
 * exchange -> strips -> exchange.  Entry / exit conditions are the same with "y halos" meaning all Hy rows.
 * Reproducibility against the single-domain run: SWMHD_STRICT slabs are bit-identical to it (every row is the same arithmetic
 * whoever computes it; tests/test_loopback_gpu.py, 2 and 3 slabs).  Fast builds are NOT: a row may be computed by a different kernel
 * variant than in the single-domain launch (interior launches pick the row-marching or the LDS-tiled kernel by slab size, the
 * boundary zones take the row-marching kernel whenever Nx >= 1024), and the deep-halo schedule's redundantly computed rows need not
 * equal the owner's copy in the last bits -- differences of the fast-kernel rounding (<= 1e-13 of the term scale per evaluation),
 * deterministic from run to run. */
int swmhd_ring_step_rk3_f64(swmhd_ring *ring, double *const *q, double *const *q_alt, double *const *Ga, double *const *Gb,
                            int Nx, int Ny, int Hx, int Hy, int64_t stride_y, double dx, double dy,
                            double g, double f, int formulation, int lorentz, double dt, int nsteps,
                            int flags, int *state_in_alt, void *stream);
int swmhd_ring_step_rk3_f32(swmhd_ring *ring, float *const *q, float *const *q_alt, float *const *Ga, float *const *Gb,
                            int Nx, int Ny, int Hx, int Hy, int64_t stride_y, float dx, float dy,
                            float g, float f, int formulation, int lorentz, float dt, int nsteps,
                            int flags, int *state_in_alt, void *stream);
int swmhd_ring_join(swmhd_ring *ring, void *stream);   /* order `stream` behind the exchange in flight (no-op if none) */

/* Measurement hook: record HIP events around the next `max_launches` interior launches of swmhd_ring_step_rk3 (0 = off);
 * swmhd_ring_launch_times waits for them and returns how many (ms, rows) pairs it wrote. */
int swmhd_ring_time_launches(swmhd_ring *ring, int max_launches);
int swmhd_ring_launch_times(swmhd_ring *ring, float *ms, int *rows, int capacity);

/* ------------------------------------------------------------------------------------------------
 * Energy and extrema diagnostics, one pass (the reference computes them every iteration:
 * kinetic/magnetic/potential_energy_func SWMHD_example.jl:67-77 / divergence_sw_mhd.jl:63-74 written by the
 * NetCDFOutputWriter :87-92, and max|u|, max|A|, min h in the progress callback :47-65).
 *   out[0..6] = { KE, ME, PE, max|u|, max|v|, max|A|, min h }  over rows j_begin+1..j_end of this (slab of the) domain,
 *   energies already multiplied by dx*dy (sum over ranks = the reference's mean(...)*Lx*Ly); always double.
 * `out` (7 doubles) and `workspace` (SWMHD_DIAG_WORKSPACE doubles) are DEVICE buffers; deterministic summation order.
 * Needs halo >= 1, filled.  For formulation SWMHD_CONSERVATIVE velocities are uh/ℑxᶠh, vh/ℑyᶠh.
 * ---------------------------------------------------------------------------------------------- */
#define SWMHD_DIAG_NOUT 7
#define SWMHD_DIAG_WORKSPACE (1024 * 7)
int swmhd_diagnostics_f64(const double *q1, const double *q2, const double *h, const double *A,
                          int Nx, int Ny, int Hx, int Hy, int64_t stride_y, double dx, double dy,
                          double g, double h_ref, int formulation, int j_begin, int j_end,
                          double *workspace, double *out, void *stream);
int swmhd_diagnostics_f32(const float *q1, const float *q2, const float *h, const float *A,
                          int Nx, int Ny, int Hx, int Hy, int64_t stride_y, float dx, float dy,
                          float g, float h_ref, int formulation, int j_begin, int j_end,
                          double *workspace, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SWMHD_H */
