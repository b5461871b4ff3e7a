"""CPU tests of the oracle's Bounded-topology restatement: wall reconstructions of the base RHS (Oceananigans'
topologically conditional interpolation, PARITY UNPINNED), the reference's wall branches of the divergence forcing
(sw_mhd_divergence_functions.jl:42-53,66-77,90-101,114-125) inside the fused tendencies, and the boundary-condition halo fill incl.
the reference's commented GradientBoundaryCondition on A (SWMHD_example.jl:18-19).  The reference holds no numeric output for any of
this; what is asserted are properties that must hold whatever the library's last bits are."""
import numpy as np
import pytest

import helpers as Hh

G, F = 9.81, 1.0
P, B = 0, 1
LOC = [(True, False), (False, True), (False, False), (False, False)]   # (face_x, face_y) of (u|uh, v|vh, h, A)


def state(Nx, Ny, seed, form, smooth=True):
    rng = np.random.default_rng(seed)
    H = 3
    shp = (Ny + 2 * H, Nx + 2 * H)
    y, x = np.meshgrid(np.arange(shp[0]) / Ny, np.arange(shp[1]) / Nx, indexing="ij")
    h = 1.0 + 0.2 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 0.02 * rng.random(shp)
    u = 0.4 * np.cos(2 * np.pi * (x + y)) + 0.05 * rng.standard_normal(shp)
    v = 0.3 * np.sin(2 * np.pi * x) * np.sin(4 * np.pi * y) + 0.05 * rng.standard_normal(shp)
    A = 0.5 * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y) + 0.05 * rng.standard_normal(shp)
    q = [u, v, h, A] if form == 1 else [h * u, h * v, h, A]
    return [np.ascontiguousarray(a) for a in q]


def fill_all(O, q, Nx, Ny, topo, gradA=None, dx=1.0, dy=1.0):
    for a, loc, gr in zip(q, LOC, (None, None, None, gradA)):
        O.fill_halo(a, Nx, Ny, 3, 3, topo=topo, face=loc, grad=gr, dx=dx, dy=dy)
    return q


def test_fill_halo_semantics(oracle):
    O = oracle
    Nx, Ny, H = 9, 7, 3
    rng = np.random.default_rng(0)
    base = rng.standard_normal((Ny + 6, Nx + 6))
    I = (slice(H, H + Ny), slice(H, H + Nx))
    # periodic/periodic == the periodic fill
    a = O.fill_halo(base.copy(), Nx, Ny, H, H)
    assert np.array_equal(a, Hh.fill_halo_periodic(base, Nx, Ny, H, H))
    # centre field, Bounded y: mirror of the interior rows, x still periodic (corners come from the y pass over the padded width)
    a = O.fill_halo(base.copy(), Nx, Ny, H, H, topo=(P, B))
    for m in range(1, H + 1):
        assert np.array_equal(a[H - m, :], a[H + m - 1, :]) and np.array_equal(a[H + Ny + m - 1, :], a[H + Ny - m, :])
    assert np.array_equal(a[I], base[I]) and np.array_equal(a[H:H + Ny, :H], a[H:H + Ny, Nx:Nx + H])
    # wall-normal velocity, Bounded y: zero ON the walls (Julia j = 1 and Ny+1) and beyond them (what the library never writes keeps
    # its allocation zeros there; the engine sets it)
    a = O.fill_halo(base.copy(), Nx, Ny, H, H, topo=(P, B), face=(False, True))
    assert np.all(a[:H + 1, :] == 0) and np.all(a[H + Ny:, :] == 0)
    assert np.array_equal(a[H + 1:H + Ny, H:H + Nx], base[H + 1:H + Ny, H:H + Nx])
    # GradientBoundaryCondition: the first halo point linearly extrapolated (the reference's A_bcs: -0.05 north and south), zeros beyond
    g, dy = -0.05, 0.3
    a = O.fill_halo(base.copy(), Nx, Ny, H, H, topo=(P, B), grad=(None, None, g, g), dy=dy)
    assert np.array_equal(a[H - 1, :], a[H, :] - g * dy) and np.array_equal(a[H + Ny, :], a[H + Ny - 1, :] + g * dy)
    assert np.all(a[:H - 1, :] == 0) and np.all(a[H + Ny + 1:, :] == 0)
    # a linear profile A = g*y is reproduced exactly in that first halo point
    yc = (np.arange(-H, Ny + H) + 0.5) * dy
    lin = np.repeat((g * yc)[:, None], Nx + 6, axis=1)
    b = O.fill_halo(lin.copy(), Nx, Ny, H, H, topo=(P, B), grad=(None, None, g, g), dy=dy)
    assert np.allclose(b[H - 1], lin[H - 1], atol=1e-15) and np.allclose(b[H + Ny], lin[H + Ny], atol=1e-15)


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2), (1, 0), (0, 0)])
@pytest.mark.parametrize("topo", [(P, B), (B, P), (B, B)])
def test_bounded_equals_periodic_away_from_walls(oracle, form, lor, topo):
    """The boundary buffers are at most 2 interpolation points deep and the widest stencil reaches 3 cells (+1 for flux differences,
    +3 for the divergence forcing's face quantities): seven cells from a wall the Bounded tendencies are the periodic ones, bit for bit."""
    O = oracle
    Nx, Ny = 40, 36
    q = state(Nx, Ny, 3, form)
    qp = fill_all(O, [a.copy() for a in q], Nx, Ny, (P, P))
    qb = fill_all(O, [a.copy() for a in q], Nx, Ny, topo)
    Gp = O.tendencies(*qp, Nx, Ny, 3, 3, 0.1, 0.12, form, lor, G, F, nthreads=4)
    Gb = O.tendencies(*qb, Nx, Ny, 3, 3, 0.1, 0.12, form, lor, G, F, nthreads=4, topo=topo)
    mx = 7 if topo[0] == B else 0
    my = 7 if topo[1] == B else 0
    far = (slice(3 + my, 3 + Ny - my), slice(3 + mx, 3 + Nx - mx))
    for a, b in zip(Gp, Gb):
        assert np.array_equal(a[far], b[far])
    near = (slice(3, 3 + Ny), slice(3, 3 + Nx))
    assert any(not np.array_equal(a[near], b[near]) for a, b in zip(Gp, Gb)), "walls changed nothing?"
    assert all(np.isfinite(b[near]).all() for b in Gb)


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
@pytest.mark.parametrize("topo", [(P, B), (B, B)])
def test_walls_conserve_mass_and_stay_impenetrable(oracle, form, lor, topo):
    """Five RK3 steps in a box: the wall-normal velocity (transport) is exactly zero on the walls after every step, so no mass leaves:
    sum(h) is conserved to rounding; a state of rest with uniform h and A stays exactly at rest."""
    O = oracle
    Nx, Ny, dx, dy, dt = 32, 28, 0.1, 0.1, 2e-3
    q = fill_all(O, state(Nx, Ny, 5, form), Nx, Ny, topo, dx=dx, dy=dy)
    I = (slice(3, 3 + Ny), slice(3, 3 + Nx))
    m0 = q[2][I].sum()
    for _ in range(5):
        O.time_step(*q, Nx, Ny, 3, 3, dx, dy, dt, form, lor, G, F, nthreads=4, topo=topo)
        if topo[0] == B:
            assert np.all(q[0][3:3 + Ny, 3] == 0) and np.all(q[0][3:3 + Ny, 3 + Nx] == 0)
        assert np.all(q[1][3, :] == 0) and np.all(q[1][3 + Ny, :] == 0)
    assert all(np.isfinite(a).all() for a in q)
    assert abs(q[2][I].sum() - m0) <= 1e-12 * m0
    rest = [np.zeros_like(q[0]), np.zeros_like(q[0]), np.full_like(q[0], 1.3), np.full_like(q[0], 0.7)]
    fill_all(O, rest, Nx, Ny, topo, dx=dx, dy=dy)
    O.time_step(*rest, Nx, Ny, 3, 3, dx, dy, dt, form, lor, G, F, nthreads=4, topo=topo)
    assert np.all(rest[0] == 0) and np.all(rest[1] == 0) and np.all(rest[2] == 1.3) and np.all(rest[3] == 0.7)


def test_wall_orders(oracle):
    """A linear profile is reproduced by every order; a quadratic one tells WENO5/third order from first order: the tracer flux next to
    a wall differs from the periodic-stencil value exactly where the buffers say it should (faces 1, 2 and N, N+1 in a Bounded x)."""
    O = oracle
    Nx, Ny = 24, 8
    H = 3
    x = (np.arange(-H, Nx + H) + 0.5)
    h = np.repeat((1.0 + 0.01 * x ** 2)[None, :], Ny + 6, axis=0)
    u = np.ones_like(h); v = np.zeros_like(h); A = np.zeros_like(h)
    q = [u, v, np.ascontiguousarray(h), A]
    Gp = O.tendencies(*q, Nx, Ny, 3, 3, 1.0, 1.0, 1, 0, G, F)                 # halos hold the analytic extension: "no walls"
    Gb = O.tendencies(*q, Nx, Ny, 3, 3, 1.0, 1.0, 1, 0, G, F, topo=(B, P))
    dGh = np.abs(Gp[2] - Gb[2])[3 + 2, 3:3 + Nx]
    changed = np.nonzero(dGh > 1e-13)[0] + 1                                  # 1-based cell indices whose G_h saw a lower-order face
    # u > 0: left-biased values; WENO5 left needs 3 <= i <= N-1, third order 2 <= i <= N: faces 1, 2 (cells 1, 2) and N, N+1 (cells N-1, N)
    assert set(changed) <= {1, 2, Nx - 1, Nx} and 1 in changed and Nx in changed
