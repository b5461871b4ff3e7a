# make_oceananigans_fixtures.jl -- the maintainer-side half of the A9 pinning kit (SURVEY.md 8(f)4, DESIGN.md section 3).
#
# TEXT DELIVERABLE: there is no Julia toolchain and no Oceananigans in the build image or on the GPU box (and no network to fetch
# them), so this script has never been executed here.  Run it wherever the reference itself runs:
#
#     julia --project=<the environment SWMHD_example.jl runs in> integration/make_oceananigans_fixtures.jl
#
# What it does: builds the two ShallowWaterModels EXACTLY as the reference's drivers do
#     jacobian_formulation/SWMHD_example.jl:21-33          VectorInvariantFormulation, WENO5(vector_invariant = VelocityStencil()),
#                                                           forcing = lorentz_force_func_x/y on (u, v)
#     divergence_formulation/divergence_sw_mhd.jl:19-31     ConservativeFormulation, WENO5, forcing = div_lorentz_x/y on (uh, vh)
# on the 48 x 40 periodic grid of tests/golden/model_48x40.npz, loads that file's prognostic parents (halos included) into the
# models' fields, and writes
#     tests/golden/oceananigans_<vi|cons>_G.npy        the four tendencies after ONE calculate_tendencies!   (4, Ny+2H, Nx+2H)
#     tests/golden/oceananigans_<vi|cons>_after2.npy   the four prognostic parents after TWO time_step!s    (4, Ny+2H, Nx+2H)
#     tests/golden/oceananigans_version.txt            Oceananigans / Julia versions (the reference pins neither)
# tests/test_reference_fixtures.py picks these files up: the moment they exist, the oracle AND the HIP engine are compared with
# them instead of reporting "parity unpinned".  Needs NPZ.jl besides what the reference's scripts already use.
#
# Layout: numpy (Ny+2H, Nx+2H) C-order == Julia (Nx+2H, Ny+2H) column-major parent, i.e. permutedims of what NPZ returns.

using Oceananigans
using Oceananigans.Models.ShallowWaterModels: VectorInvariantFormulation, ConservativeFormulation
using Oceananigans.Advection: VelocityStencil
using Oceananigans.Operators
using Oceananigans.Grids: AbstractGrid, topology
using Oceananigans.TimeSteppers: time_step!, update_state!
using NPZ, Pkg

const ROOT   = normpath(joinpath(@__DIR__, ".."))
const REF    = get(ENV, "SWMHD_REFERENCE", joinpath(ROOT, "..", "reference"))     # checkout of writingindy/SWMHD
const GOLDEN = joinpath(ROOT, "tests", "golden")

include(joinpath(REF, "jacobian_formulation", "sw_mhd_jacobian_functions.jl"))
include(joinpath(REF, "divergence_formulation", "sw_mhd_divergence_functions.jl"))

z = npzread(joinpath(GOLDEN, "model_48x40.npz"))
Nx, Ny, H, dt = Int(z["Nx"]), Int(z["Ny"]), Int(z["H"]), Float64(z["dt"])
Lx, Ly = Nx * Float64(z["dx"]), Ny * Float64(z["dy"])                              # 2π x 2π (tests/test_model_oracle.py)

grid = RectilinearGrid(size = (Nx, Ny), x = (0, Lx), y = (0, Ly), topology = (Periodic, Periodic, Flat), halo = (H, H))

function build(tag)
    if tag == "vi"       # SWMHD_example.jl:21-33
        ShallowWaterModel(grid = grid, timestepper = :RungeKutta3,
                          momentum_advection = WENO5(vector_invariant = VelocityStencil()),
                          mass_advection = WENO5(), tracer_advection = WENO5(),
                          gravitational_acceleration = 9.81, coriolis = FPlane(f = 1), tracers = (:A),
                          forcing = (u = Forcing(lorentz_force_func_x, discrete_form = true),
                                     v = Forcing(lorentz_force_func_y, discrete_form = true)),
                          formulation = VectorInvariantFormulation())
    else                 # divergence_sw_mhd.jl:19-31
        ShallowWaterModel(grid = grid, timestepper = :RungeKutta3,
                          momentum_advection = WENO5(), mass_advection = WENO5(), tracer_advection = WENO5(),
                          gravitational_acceleration = 9.81, coriolis = FPlane(f = 1), tracers = (:A),
                          forcing = (uh = Forcing(div_lorentz_x, discrete_form = true),
                                     vh = Forcing(div_lorentz_y, discrete_form = true)),
                          formulation = ConservativeFormulation())
    end
end

prognostic(model) = (model.solution[1], model.solution[2], model.solution.h, model.tracers.A)
tendencies(model) = (model.timestepper.Gⁿ[1], model.timestepper.Gⁿ[2], model.timestepper.Gⁿ.h, model.timestepper.Gⁿ.A)
to_numpy(fields) = permutedims(cat((Array(parent(f))[:, :, 1] for f in fields)...; dims = 3), (3, 2, 1))   # (4, Ny+2H, Nx+2H)

# Which call fills Gⁿ depends on the library vintage.  In the v0.7x releases the reference's `WENO5(vector_invariant = ...)` needs,
# `update_state!` only fills halos and auxiliary fields; the tendencies are computed inside `time_step!` by
# `calculate_tendencies!(model)` (Models.ShallowWaterModels).  Later releases renamed it `compute_tendencies!` and moved the call
# into `update_state!`.  Call it explicitly, whichever exists, and record which branch ran.
const SWM = Oceananigans.Models.ShallowWaterModels
function fill_tendencies!(model)
    update_state!(model)                                # fill_halo_regions! (a no-op on these already-periodic parents)
    if isdefined(SWM, :calculate_tendencies!)
        Base.invokelatest(getfield(SWM, :calculate_tendencies!), model)
        return "Models.ShallowWaterModels.calculate_tendencies!(model)"
    elseif isdefined(SWM, :compute_tendencies!)
        f = getfield(SWM, :compute_tendencies!)
        try
            Base.invokelatest(f, model, [])             # (model, callbacks) in the releases that have callbacks here
        catch err
            err isa MethodError || rethrow()
            Base.invokelatest(f, model)
        end
        return "Models.ShallowWaterModels.compute_tendencies!(model[, callbacks])"
    elseif isdefined(Oceananigans.TimeSteppers, :calculate_tendencies!)
        Base.invokelatest(getfield(Oceananigans.TimeSteppers, :calculate_tendencies!), model)
        return "TimeSteppers.calculate_tendencies!(model)"
    end
    error("no calculate_tendencies!/compute_tendencies! found in this Oceananigans: fill Gⁿ by hand and re-run")
end

for tag in ("vi", "cons")
    model = build(tag)
    q = z["$(tag)_q"]                                   # (4, Ny+2H, Nx+2H)
    for (f, k) in zip(prognostic(model), 1:4)
        size(parent(f))[1:2] == (Nx + 2H, Ny + 2H) || error("parent of field $k is $(size(parent(f))): halo or location mismatch")
        parent(f)[:, :, 1] .= permutedims(q[k, :, :])   # halos included: exactly the bytes the HIP engine was given
    end
    global tendency_call = fill_tendencies!(model)      # Gⁿ = tendencies of the initial state
    for (g, k) in zip(tendencies(model), 1:4)           # a G of zeros would "prove" the restatement wrong: refuse to write it
        any(!iszero, parent(g)) || error("tendency $k of the $tag model is identically zero after $tendency_call: " *
                                         "this Oceananigans computes tendencies elsewhere; nothing written")
    end
    npzwrite(joinpath(GOLDEN, "oceananigans_$(tag)_G.npy"), to_numpy(tendencies(model)))
    for _ in 1:2
        time_step!(model, dt)
    end
    npzwrite(joinpath(GOLDEN, "oceananigans_$(tag)_after2.npy"), to_numpy(prognostic(model)))
end

open(joinpath(GOLDEN, "oceananigans_version.txt"), "w") do io
    println(io, "julia ", VERSION)
    println(io, "tendencies filled by: ", tendency_call)
    for (_, p) in Pkg.dependencies()
        p.name in ("Oceananigans", "KernelAbstractions", "NPZ") && println(io, p.name, " ", p.version)
    end
end
