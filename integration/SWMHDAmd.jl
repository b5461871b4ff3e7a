# SWMHDAmd.jl -- Julia binding of libswmhd.so (include/swmhd.h) for the reference's two driver scripts.
#
# TEXT DELIVERABLE: there is no Julia toolchain (and no Oceananigans) in the build image or on the GPU box, so this file has
# never been executed.  It is the glue a maintainer of writingindy/SWMHD would add; every ccall below matches a prototype in
# include/swmhd.h one to one, and the Python/ctypes binding swmhd_amd/_lib.py (which IS tested) makes the same calls.
#
# What it replaces in the reference:
#   jacobian_formulation/SWMHD_example.jl:30-31     forcing = (u = Forcing(lorentz_force_func_x, discrete_form=true), v = ...)
#   divergence_formulation/divergence_sw_mhd.jl:28-29  forcing = (uh = Forcing(div_lorentz_x, ...), vh = ...)
# The forcing keeps its discrete-form signature f(i, j, k, grid, clock, fields, parameters) and just reads an auxiliary field
# that one whole-field kernel launch fills (the per-cell functions of sw_mhd_jacobian_functions.jl / sw_mhd_divergence_functions.jl
# are evaluated for all cells at once on the GPU).
module SWMHDAmd

using Oceananigans, AMDGPU
using Oceananigans.Grids: topology

const libswmhd = get(ENV, "SWMHD_LIB", joinpath(@__DIR__, "..", "swmhd_amd", "libswmhd.so"))

const SWMHD_FAST, SWMHD_STRICT = Cint(0), Cint(1)
const SWMHD_WRAP_X, SWMHD_WRAP_Y = Cint(16), Cint(32)            # read periodic images instead of halo cells (no halo launch per stage)
const SWMHD_BOUNDED_X, SWMHD_BOUNDED_Y = Cint(256), Cint(512)    # Grids.topology(grid, d) == Bounded
const SWMHD_PERIODIC, SWMHD_BOUNDED = Cint(0), Cint(1)
const CONSERVATIVE, VECTOR_INVARIANT = Cint(0), Cint(1)
const LORENTZ_NONE, LORENTZ_JACOBIAN, LORENTZ_DIVERGENCE = Cint(0), Cint(1), Cint(2)

check(rc) = rc == 0 || error(unsafe_string(ccall((:swmhd_strerror, libswmhd), Cstring, (Cint,), rc)))

# parent pointer, extents and stride of a halo-padded Oceananigans field (column-major (Nx+2Hx, Ny+2Hy, 1) parent)
pp(f) = pointer(parent(f))
stride_y(f) = Int64(size(parent(f), 1))
hipstream() = AMDGPU.stream().stream

"Fill (Fx, Fy) with lorentz_force_func_x/y for every interior cell (sw_mhd_jacobian_functions.jl:20-26)."
function lorentz_jacobian!(Fx, Fy, A, h, grid; flags = SWMHD_FAST)
    check(ccall((:swmhd_lorentz_jacobian_f64, libswmhd), Cint,
                (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Cint, Cint, Int64, Float64, Float64, Cint, Ptr{Cvoid}),
                pp(A), pp(h), pp(Fx), pp(Fy), grid.Nx, grid.Ny, grid.Hx, grid.Hy, stride_y(A), grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, flags, hipstream()))
end

"Fill (Fx, Fy) with div_lorentz_x/y for every interior cell (sw_mhd_divergence_functions.jl:162-170)."
function lorentz_divergence!(Fx, Fy, A, h, grid; flags = SWMHD_FAST)
    check(ccall((:swmhd_lorentz_divergence_f64, libswmhd), Cint,
                (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Cint, Cint, Int64, Float64, Float64, Cint, Ptr{Cvoid}),
                pp(A), pp(h), pp(Fx), pp(Fy), grid.Nx, grid.Ny, grid.Hx, grid.Hy, stride_y(A), grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, flags, hipstream()))
end

# the forcing callbacks keep the reference's signature; `p` carries the auxiliary field
@inline lorentz_x(i, j, k, grid, clock, fields, p) = @inbounds p.Fx[i, j, k]
@inline lorentz_y(i, j, k, grid, clock, fields, p) = @inbounds p.Fy[i, j, k]

"""
    mhd_shallow_water_model(grid; formulation) -> (model, update!)

The model of SWMHD_example.jl:21-33 (VectorInvariantFormulation) or divergence_sw_mhd.jl:19-31 (ConservativeFormulation) with
the MHD forcing evaluated by libswmhd.  Call `update!(model)` after every halo fill (Oceananigans >= 0.80:
`add_callback!(simulation, update!, callsite = UpdateStateCallsite())`).
"""
function mhd_shallow_water_model(grid; formulation = VectorInvariantFormulation(), g = 9.81, f = 1)
    Fx, Fy = XFaceField(grid), YFaceField(grid)
    jac = formulation isa VectorInvariantFormulation
    names = jac ? (:u, :v) : (:uh, :vh)
    forcing = NamedTuple{names}((Forcing(lorentz_x, discrete_form = true, parameters = (; Fx)),
                                 Forcing(lorentz_y, discrete_form = true, parameters = (; Fy))))
    model = ShallowWaterModel(; grid, timestepper = :RungeKutta3, gravitational_acceleration = g, coriolis = FPlane(f = f),
                              momentum_advection = jac ? WENO5(vector_invariant = VelocityStencil()) : WENO5(),
                              mass_advection = WENO5(), tracer_advection = WENO5(), tracers = (:A,), forcing, formulation)
    update!(m) = jac ? lorentz_jacobian!(Fx, Fy, m.tracers.A, m.solution.h, grid) :
                       lorentz_divergence!(Fx, Fy, m.tracers.A, m.solution.h, grid)
    return model, update!
end

"""
    native_steps!(q, q_alt, Ga, Gb, grid, Δt, n; formulation, lorentz) -> state_in_alt::Bool

Hand `n` whole RK3 steps to the engine (swmhd_step_rk3_f64): q, q_alt are 4-tuples of fields (u|uh, v|vh, h, A) -- current state
and a scratch copy --, Ga, Gb two 4-tuples of tendency fields.  Periodic single-GPU grids.  With `flags = SWMHD_WRAP_X | SWMHD_WRAP_Y`
no halo launch runs between the stages (3 launches per step); the halos of the returned state are then stale: `fill_halos!` before
anything else reads them.  The base right-hand side inside is a
restatement of Oceananigans' scheme that could not be checked against the library (DESIGN.md section 3).
"""
function native_steps!(q, q_alt, Ga, Gb, grid, Δt, n; formulation = VECTOR_INVARIANT, lorentz = LORENTZ_JACOBIAN, g = 9.81, f = 1.0,
                       flags = SWMHD_FAST)
    ptrs(t) = Ptr{Float64}[pp(x) for x in t]
    swapped = Ref{Cint}(0)
    check(ccall((:swmhd_step_rk3_f64, libswmhd), Cint,
                (Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cint, Cint, Cint, Cint, Int64,
                 Float64, Float64, Float64, Float64, Cint, Cint, Float64, Cint, Cint, Ptr{Cint}, Ptr{Cvoid}),
                ptrs(q), ptrs(q_alt), ptrs(Ga), ptrs(Gb), grid.Nx, grid.Ny, grid.Hx, grid.Hy, stride_y(q[1]),
                grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, g, f, formulation, lorentz, Δt, n, flags, swapped, hipstream()))
    return swapped[] != 0
end

"Energies and extrema in one device pass: (KE, ME, PE, max|u|, max|v|, max|A|, min h), SWMHD_example.jl:47-77."
function diagnostics(q, grid; formulation = VECTOR_INVARIANT, g = 9.81, h_ref = 1.0)
    ws, out = AMDGPU.zeros(Float64, 1024 * 7), AMDGPU.zeros(Float64, 7)
    check(ccall((:swmhd_diagnostics_f64, libswmhd), Cint,
                (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Cint, Cint, Int64, Float64, Float64, Float64,
                 Float64, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                pp(q[1]), pp(q[2]), pp(q[3]), pp(q[4]), grid.Nx, grid.Ny, grid.Hx, grid.Hy, stride_y(q[1]), grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ,
                g, h_ref, formulation, 0, grid.Ny, pointer(ws), pointer(out), hipstream()))
    return Array(out)
end

"Topology flags of a grid for the tendency entry points (the reference tests `topology(grid, d) == Bounded`, sw_mhd_divergence_functions.jl:42)."
topology_flags(grid) = (topology(grid, 1) == Bounded ? SWMHD_BOUNDED_X : Cint(0)) | (topology(grid, 2) == Bounded ? SWMHD_BOUNDED_Y : Cint(0))

"""
    fill_halos!(q, grid; A_gradients = (NaN, NaN, NaN, NaN))

fill_halo_regions! of the four prognostic fields (u|uh, v|vh, h, A) for any (Periodic | Bounded) topology pair with Oceananigans' default
boundary conditions; `A_gradients` = (west, east, south, north) GradientBoundaryCondition values of A (NaN = default), e.g. the
reference's commented `A_bcs` (SWMHD_example.jl:18-19): `(NaN, NaN, -0.05, -0.05)`.  swmhd_fill_halo_f64.
"""
function fill_halos!(q, grid; A_gradients = (NaN, NaN, NaN, NaN))
    ptrs = Ptr{Float64}[pp(x) for x in q]
    grad = vcat(fill(NaN, 12), collect(Float64, A_gradients))            # fields 0..2 default, field 3 = A
    tx = topology(grid, 1) == Bounded ? SWMHD_BOUNDED : SWMHD_PERIODIC
    ty = topology(grid, 2) == Bounded ? SWMHD_BOUNDED : SWMHD_PERIODIC
    check(ccall((:swmhd_fill_halo_f64, libswmhd), Cint,
                (Ptr{Ptr{Float64}}, Cint, Cint, Cint, Cint, Cint, Int64, Cint, Cint, Cint, Cint, Ptr{Float64}, Float64, Float64, Ptr{Cvoid}),
                ptrs, 4, grid.Nx, grid.Ny, grid.Hx, grid.Hy, stride_y(q[1]), tx, ty,
                0b0001 #= field 0 (u|uh) is at Face in x =#, 0b0010 #= field 1 (v|vh) is at Face in y =#, grad,
                grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, hipstream()))
end

"""
    native_stage!(q, qnew, Gn, Gm, grid, Δt, γ, ζ, store_G; ...)

ONE fused RK3 stage (calculate_tendencies! + rk3_substep!, swmhd_tendencies_rk3_f64) on any topology: the caller swaps q <-> qnew and
Gn <-> Gm afterwards and, in Bounded directions, calls `fill_halos!`; Periodic directions need no halo fill when `flags` carry
SWMHD_WRAP_X / SWMHD_WRAP_Y (the kernels read the periodic images themselves).  `Gm === nothing` on the first stage.
"""
function native_stage!(q, qnew, Gn, Gm, grid, Δt, γ, ζ, store_G; formulation = VECTOR_INVARIANT, lorentz = LORENTZ_JACOBIAN, g = 9.81, f = 1.0,
                       flags = SWMHD_FAST)
    ptrs(t) = Ptr{Float64}[pp(x) for x in t]
    fl = flags | topology_flags(grid)
    topology(grid, 1) == Periodic && (fl |= SWMHD_WRAP_X)
    topology(grid, 2) == Periodic && (fl |= SWMHD_WRAP_Y)
    check(ccall((:swmhd_tendencies_rk3_f64, libswmhd), Cint,
                (Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cint, Cint, Cint, Cint, Int64,
                 Float64, Float64, Float64, Float64, Cint, Cint, Float64, Float64, Float64, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                ptrs(q), ptrs(qnew), ptrs(Gn), Gm === nothing ? C_NULL : ptrs(Gm), grid.Nx, grid.Ny, grid.Hx, grid.Hy, stride_y(q[1]),
                grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, g, f, formulation, lorentz, Δt, γ, ζ, store_G ? 1 : 0, 0, grid.Ny, fl, hipstream()))
end

# ---- several GPUs: one Julia process per GPU (e.g. MPI.jl), the domain cut into y-slabs -----------------------------------
# The reference is single-process; its periodic y boundary (fill_halo_regions! for (Periodic, Periodic, Flat),
# SWMHD_example.jl:16) becomes a ring of RCCL sends/receives between y-neighbours (swmhd_ring_*, include/swmhd.h).

"Create this rank's ring.  `bcast!(id::Vector{UInt8}, root)` is any out-of-band broadcast, e.g. `(id, r) -> MPI.Bcast!(id, r, comm)`."
function ring_create(nranks, rank, bcast!; rccl_path = C_NULL)
    id = zeros(UInt8, 128)
    rank == 0 && check(ccall((:swmhd_ring_unique_id, libswmhd), Cint, (Cstring, Ptr{UInt8}), rccl_path, id))
    bcast!(id, 0)
    ring = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:swmhd_ring_create, libswmhd), Cint, (Ptr{Ptr{Cvoid}}, Cstring, Cint, Cint, Ptr{UInt8}), ring, rccl_path, nranks, rank, id))
    return ring[]
end
ring_destroy(ring) = check(ccall((:swmhd_ring_destroy, libswmhd), Cint, (Ptr{Cvoid},), ring))
"Order the current HIP stream behind the halo exchange `ring_steps!` left in flight (before anything else reads the y halos)."
ring_join(ring) = check(ccall((:swmhd_ring_join, libswmhd), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ring, hipstream()))

"""
    ring_steps!(ring, q, q_alt, Ga, Gb, grid, Δt, n; ...) -> state_in_alt::Bool

`native_steps!` for one y-slab of the ring (`grid` is the LOCAL slab grid, halos of `q` filled on entry): the neighbour exchange of
every stage runs on the ring's stream while the interior rows of the next stage compute.  Build the slab grid with `halo = (3, 9)`
(and pass `flags = SWMHD_FAST | SWMHD_WRAP_X`) to get the deep-halo schedule: one exchange of 9 rows per RK3 step instead of 3 rows per stage, boundary
rows of stages 1-2 evaluated redundantly inside the halo (include/swmhd.h, swmhd_ring_step_rk3).
"""
function ring_steps!(ring, q, q_alt, Ga, Gb, grid, Δt, n; formulation = VECTOR_INVARIANT, lorentz = LORENTZ_JACOBIAN, g = 9.81, f = 1.0,
                     flags = SWMHD_FAST)
    ptrs(t) = Ptr{Float64}[pp(x) for x in t]
    swapped = Ref{Cint}(0)
    check(ccall((:swmhd_ring_step_rk3_f64, libswmhd), Cint,
                (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Cint, Cint, Cint, Cint, Int64,
                 Float64, Float64, Float64, Float64, Cint, Cint, Float64, Cint, Cint, Ptr{Cint}, Ptr{Cvoid}),
                ring, ptrs(q), ptrs(q_alt), ptrs(Ga), ptrs(Gb), grid.Nx, grid.Ny, grid.Hx, grid.Hy, stride_y(q[1]),
                grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, g, f, formulation, lorentz, Δt, n, flags, swapped, hipstream()))
    return swapped[] != 0
end

end # module
