"""GPU parity of the Bounded-topology path through the C-ABI against the CPU oracle: boundary-condition halo fill
(swmhd_fill_halo_*), fused tendencies with SWMHD_BOUNDED_X / _Y (wall reconstructions + the reference's wall branches of the divergence
forcing, sw_mhd_divergence_functions.jl:42-53,66-77,90-101,114-125), and whole RK3 steps of a (Periodic, Bounded) model with the
reference's commented GradientBoundaryCondition on A (SWMHD_example.jl:18-19).  Bars as for the periodic engine: SWMHD_STRICT
bit-identical, fast <= 1e-12 max-norm (fp64)."""
import numpy as np
import pytest
import torch

from test_bounded_oracle import state, fill_all, LOC, G, F, P, B

pytestmark = pytest.mark.gpu
FORM = {0: "Conservative", 1: "VectorInvariant"}
TOPO = {P: "Periodic", B: "Bounded"}


def grid_for(S, Nx, Ny, topo, dx=0.1, dy=0.12):
    return S.RectilinearGrid(size=(Nx, Ny), x=(0, dx * Nx), y=(0, dy * Ny), topology=(TOPO[topo[0]], TOPO[topo[1]], "Flat"))


@pytest.mark.parametrize("topo", [(P, P), (P, B), (B, P), (B, B)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fill_halo_matches_oracle(swmhd, oracle, topo, dtype):
    S, O = swmhd, oracle
    Nx, Ny = 37, 21
    g = grid_for(S, Nx, Ny, topo)
    rng = np.random.default_rng(11)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    for loc, name in zip(LOC, ("u", "v", "h", "A")):
        base = np.ascontiguousarray(rng.standard_normal(g.parent_shape).astype(dtype))
        bc, grad = None, None
        if name == "A" and topo[1] == B:
            bc = S.FieldBoundaryConditions(north=S.GradientBoundaryCondition(-0.05), south=S.GradientBoundaryCondition(0.125))
            grad = (None, None, 0.125, -0.05)
        want = O.fill_halo(base.copy(), Nx, Ny, 3, 3, topo=topo, face=loc, grad=grad, dx=g.dx, dy=g.dy)
        f = S.Field(g, (S.Face if loc[0] else S.Center, S.Face if loc[1] else S.Center), tdt, data=torch.from_numpy(base.copy()).cuda())
        f.fill_halo_regions(boundary_conditions=bc)
        torch.cuda.synchronize()
        assert np.array_equal(want, f.numpy()), (name, topo)


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2), (1, 0), (0, 0)])
@pytest.mark.parametrize("topo", [(P, B), (B, P), (B, B)])
@pytest.mark.parametrize("shape", [(40, 36), (70, 9), (7, 8)])
def test_bounded_tendencies_strict_bitwise_fast_within_tolerance(swmhd, oracle, form, lor, topo, shape):
    S, O = swmhd, oracle
    Nx, Ny = shape
    g = grid_for(S, Nx, Ny, topo)
    q = fill_all(O, state(Nx, Ny, 17 + Nx, form), Nx, Ny, topo, dx=g.dx, dy=g.dy)
    want = O.tendencies(*q, Nx, Ny, 3, 3, g.dx, g.dy, form, lor, G, F, nthreads=8, topo=topo)
    I = g.interior
    for strict in (True, False):
        m = S.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=bool(lor), strict=strict)
        for f, a in zip(m._raw_fields, q):
            f.data.copy_(torch.from_numpy(a))
        m.calculate_tendencies(); torch.cuda.synchronize()
        for w, gf in zip(want, m.Gn):
            got = gf.numpy()
            if strict:
                assert np.array_equal(w[I], got[I]), f"strict differs by {np.abs(w[I] - got[I]).max()}"
            else:
                assert np.abs(w[I] - got[I]).max() <= 1e-12 * np.abs(w[I]).max()


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
@pytest.mark.parametrize("topo,with_bc", [((P, B), True), ((B, B), False), ((B, P), False)])
def test_bounded_time_steps_match_oracle(swmhd, oracle, form, lor, topo, with_bc):
    """Three RK3 steps (fused stage kernel + boundary-condition halo fill per stage) == the oracle's time_step, halos included, bit for
    bit in strict builds; the (Periodic, Bounded) case carries the reference's commented A_bcs (gradient -0.05 north and south)."""
    S, O = swmhd, oracle
    Nx, Ny, dt = 48, 40, 2e-3
    g = grid_for(S, Nx, Ny, topo, 0.1, 0.1)
    gradA = (None, None, -0.05, -0.05) if with_bc else None
    bcs = {"A": S.FieldBoundaryConditions(north=S.GradientBoundaryCondition(-0.05), south=S.GradientBoundaryCondition(-0.05))} if with_bc else None
    q = fill_all(O, state(Nx, Ny, 9, form), Nx, Ny, topo, gradA=gradA, dx=g.dx, dy=g.dy)
    qo = [a.copy() for a in q]
    ms = {}
    for strict in (True, False):
        m = S.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=True, strict=strict, boundary_conditions=bcs)
        for f, a in zip(m._raw_fields, q):
            f.data.copy_(torch.from_numpy(a))
        ms[strict] = m
    for _ in range(3):
        O.time_step(*qo, Nx, Ny, 3, 3, g.dx, g.dy, dt, form, lor, G, F, nthreads=8, topo=topo, gradA=gradA)
        for m in ms.values():
            m.time_step(dt)
    for m in ms.values():
        m.synchronize()
    for w, f in zip(qo, ms[True].fields):
        assert np.array_equal(w, f.numpy()), "strict Bounded time stepping (incl. halos) differs from the oracle"
    for w, f in zip(qo, ms[False].fields):
        assert np.abs(w - f.numpy()).max() <= 1e-12 * max(np.abs(w).max(), 1.0)
    # time_steps() takes the same path (the periodic C step driver refuses Bounded grids)
    m2 = S.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=True, strict=True, boundary_conditions=bcs)
    for f, a in zip(m2._raw_fields, q):
        f.data.copy_(torch.from_numpy(a))
    m2.time_steps(3, dt); m2.synchronize()
    for a, b in zip(ms[True].fields, m2.fields):
        assert torch.equal(a.data, b.data)


def test_bounded_refusals(swmhd):
    S = swmhd
    g = grid_for(S, 32, 32, (P, B))
    with pytest.raises(S._lib.SwmhdError):
        S.ShallowWaterModel(g, kernel="march").calculate_tendencies()          # walls: LDS-tiled kernel only
    with pytest.raises(S._lib.SwmhdError):
        S.ShallowWaterModel(g, decomp=S.SlabDecomposition(32, 1, 0, force_ring=True))
    gp = grid_for(S, 32, 32, (P, P))
    with pytest.raises(S._lib.SwmhdError):                                       # a boundary condition on a Periodic side
        S.ShallowWaterModel(gp, boundary_conditions={"A": S.FieldBoundaryConditions(north=S.GradientBoundaryCondition(-0.05))})


def test_bounded_large_grid_runs_on_tile_kernel(swmhd, oracle):
    """A (Periodic, Bounded) 1024 x 512 grid (above the marching threshold): still the tile kernel, fast vs oracle."""
    S, O = swmhd, oracle
    Nx, Ny, topo = 1024, 512, (P, B)
    g = grid_for(S, Nx, Ny, topo, 0.01, 0.01)
    q = fill_all(O, state(Nx, Ny, 2, 1), Nx, Ny, topo, dx=g.dx, dy=g.dy)
    want = O.tendencies(*q, Nx, Ny, 3, 3, g.dx, g.dy, 1, 1, G, F, nthreads=8, topo=topo)
    m = S.ShallowWaterModel(g, G, F, formulation="VectorInvariant")
    for f, a in zip(m._raw_fields, q):
        f.data.copy_(torch.from_numpy(a))
    m.calculate_tendencies(); torch.cuda.synchronize()
    I = g.interior
    for w, gf in zip(want, m.Gn):
        assert np.abs(w[I] - gf.numpy()[I]).max() <= 1e-12 * np.abs(w[I]).max()


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
@pytest.mark.parametrize("topo", [(P, B), (B, P), (B, B)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_large_bounded_grid_hybrid_launch_matches_oracle_and_tile_kernel(swmhd, oracle, form, lor, topo, dtype):
    """Bounded grids from ~0.3 Mcell on: the row-marching kernel computes every row with the periodic formulas (boundary-condition halos
    read from memory) and the LDS-tiled Bounded kernel overwrites a frame (8 rows along y walls, the outer 64-column tile columns
    along x walls).  Against the oracle within the fast tolerance everywhere -- a frame too narrow would leave O(1e-3) errors beside
    it -- and bitwise equal to the all-tile launch inside the frame; tendencies, and the state after one RK3 step within tolerance."""
    S, O = swmhd, oracle
    Nx, Ny = 700, 520
    g = grid_for(S, Nx, Ny, topo)
    q = fill_all(O, state(Nx, Ny, 5, form), Nx, Ny, topo, dx=g.dx, dy=g.dy)
    q = [np.ascontiguousarray(a.astype(dtype)) for a in q]
    want = O.tendencies(*[a.astype(np.float64) for a in q], Nx, Ny, 3, 3, g.dx, g.dy, form, lor, G, F, nthreads=8, topo=topo)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    I = g.interior
    out = {}
    for kern in ("auto", "tile"):
        m = S.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=bool(lor), kernel=kern, dtype=tdt)
        for f, a in zip(m._raw_fields, q):
            f.data.copy_(torch.from_numpy(a))
        m.calculate_tendencies(); torch.cuda.synchronize()
        out[kern] = [gf.numpy().copy() for gf in m.Gn]
        m.time_step(1e-3); m.synchronize()
        out[kern + "_state"] = [f.numpy().copy() for f in m.fields]
    tol = 1e-12 if dtype == np.float64 else 1e-4
    scale = max(np.abs(w[I]).max() for w in want)
    for w, a in zip(want, out["auto"]):
        assert np.abs(w[I] - a[I]).max() <= tol * scale
    frame = np.zeros((Ny, Nx), dtype=bool)
    if topo[1] == B:
        frame[:8] = True; frame[-8:] = True
    if topo[0] == B:
        frame[:, :64] = True; frame[:, 64 * ((Nx - 1) // 64):] = True
    for key in ("auto", "auto_state"):
        for a, t in zip(out[key], out[key.replace("auto", "tile")]):
            if key == "auto":   # (after a whole step the frame has read its rounding-different neighbours: tolerance only)
                assert np.array_equal(a[I][frame], t[I][frame])                        # the frame IS the tile kernel's output
            assert np.abs(a[I] - t[I]).max() <= tol * max(np.abs(t[I]).max(), scale)  # elsewhere: rounding only
    if form == 1 and dtype == np.float64:   # (the vector-invariant marching kernel rounds differently from the tile kernel: proof that it ran)
        assert any(not np.array_equal(a[I], t[I]) for a, t in zip(out["auto"], out["tile"]))


def test_bounded_model_graph_replay_equals_eager(swmhd):
    """HIP-graph replay (capture_graph / time_steps) of a Bounded model on the reference's own grid size: the step is 3 fused stages
    + 3 boundary-condition fills, all plain launches on the caller's stream; bit-identical to eager stepping, odd leftovers included."""
    S = swmhd
    g = S.RectilinearGrid(size=(64, 64), x=(-5, 5), y=(-5, 5), topology=("Periodic", "Bounded", "Flat"))
    bcs = {"A": S.FieldBoundaryConditions(north=S.GradientBoundaryCondition(-0.05), south=S.GradientBoundaryCondition(-0.05))}
    ms = []
    for _ in range(2):
        m = S.ShallowWaterModel(g, 9.81, 1.0, formulation="VectorInvariant", boundary_conditions=bcs)
        m.set(u=lambda X, Y: 0.1 * np.exp(-(X ** 2 + Y ** 2)), v=lambda X, Y: 0 * X, h=lambda X, Y: 1 + 0.01 * np.cos(0.3 * X),
              A=lambda X, Y: 0.05 * np.abs(Y))
        ms.append(m)
    a, b = ms
    dt = 2e-3
    b.capture_graph(dt)
    for n in (7, 2, 5):
        b.time_steps(n, dt)
        for _ in range(n):
            a.time_step(dt)
        a.synchronize(); b.synchronize()
        for fa, fb in zip(a.fields, b.fields):
            assert torch.equal(fa.data, fb.data)
    assert a.iteration == b.iteration == 14
