"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs.

Bars:
  SWMHD_STRICT  -> bit-identical to the oracle (same operation order, IEEE divides, no FMA contraction)
  fast (default)-> max|dF| <= 1e-13 * max|F| (fp64),  <= 2e-5 * max|F| (fp32)
"""
import os

import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = {np.float64: 1e-13, np.float32: 2e-5}


def _fields(S, Nx, Ny, H, A, h, d=(0.37, 0.41)):
    g = S.RectilinearGrid(size=(Nx, Ny), x=(0, d[0] * Nx), y=(0, d[1] * Ny), halo=(H, H))
    tdt = torch.float64 if A.dtype == np.float64 else torch.float32
    fa = S.Field(g, dtype=tdt, data=torch.from_numpy(A).cuda())
    fh = S.Field(g, dtype=tdt, data=torch.from_numpy(h).cuda())
    return g, {"A": fa, "h": fh}


def _run(S, form, g, fields, **kw):
    fn = S.lorentz_force_func if form == "jacobian" else S.div_lorentz
    Fx, Fy = fn(g, fields, **kw)
    torch.cuda.synchronize()
    return Fx.numpy(), Fy.numpy()


def _oracle(O, form, A, h, g):
    fn = O.lorentz_jacobian if form == "jacobian" else O.lorentz_divergence
    return fn(A, h, g.Nx, g.Ny, g.Hx, g.Hy, g.dx, g.dy, nthreads=8)


SHAPES = [(64, 16), (100, 37), (5, 4), (130, 70), (256, 256), (513, 33), (1, 1), (252, 40), (253, 3), (700, 129)]


@pytest.mark.parametrize("form", ["jacobian", "divergence"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_strict_is_bitwise_and_fast_within_tolerance(swmhd, oracle, form, dtype, shape):
    Nx, Ny = shape
    H = 3
    if Nx < H or Ny < H:   # periodic fill needs H <= N; tiny grids get non-periodic random halos instead
        A, h = Hh.random_case(Nx, Ny, H, H, 100 + Nx, dtype, periodic=False)
    else:
        A, h = Hh.random_case(Nx, Ny, H, H, 100 + Nx, dtype)
    g, f = _fields(swmhd, Nx, Ny, H, A, h)
    want = _oracle(oracle, form, A, h, g)
    got = _run(swmhd, form, g, f, strict=True)
    I = g.interior
    for w, q in zip(want, got):
        assert np.array_equal(w[I], q[I]), f"strict {form} differs from oracle: max {np.abs(w[I]-q[I]).max()}"
    for tile in (False, True):   # both fast kernels: row-marching and LDS-tiled
        fast = _run(swmhd, form, g, f, strict=False, kernel=("tile" if tile else "march"))
        for w, q in zip(want, fast):
            assert np.abs(w[I] - q[I]).max() <= TOL[dtype] * max(np.abs(w[I]).max(), 1e-300), ("tile" if tile else "march")


@pytest.mark.parametrize("form", ["jacobian", "divergence"])
def test_outputs_outside_interior_untouched_and_row_ranges(swmhd, oracle, form):
    Nx, Ny, H = 70, 50, 3
    A, h = Hh.random_case(Nx, Ny, H, H, 9)
    g, f = _fields(swmhd, Nx, Ny, H, A, h)
    want = _oracle(oracle, form, A, h, g)
    sentinel = -777.25
    out = (swmhd.Field(g), swmhd.Field(g))
    for o in out:
        o.data.fill_(sentinel)
    got = _run(swmhd, form, g, f, out=out, strict=True, rows=(13, 31))
    for w, q in zip(want, got):
        assert np.array_equal(q[H + 13:H + 31, H:H + Nx], w[H + 13:H + 31, H:H + Nx])
        mask = np.ones_like(q, dtype=bool)
        mask[H + 13:H + 31, H:H + Nx] = False
        assert np.all(q[mask] == sentinel), "kernel wrote outside the requested rows / into halos"


@pytest.mark.parametrize("H", [2, 3])
@pytest.mark.parametrize("Nx", [16, 250, 252, 254, 505, 1300])
def test_fp32_jacobian_marching_operator_at_strip_boundaries(swmhd, oracle, H, Nx):
    """The fp32 row-marching Jacobian operator (252 output columns per workgroup): widths around the strip boundaries, the minimal halo
    the Jacobian form accepts, non-periodic random halos (so that a wrong halo column shows), a row sub-range, and nothing written
    outside it."""
    Ny = 23
    A, h = Hh.random_case(Nx, Ny, H, H, 300 + Nx, np.float32, periodic=False)
    g, f = _fields(swmhd, Nx, Ny, H, A, h)
    want = _oracle(oracle, "jacobian", A, h, g)
    I = g.interior
    got = _run(swmhd, "jacobian", g, f, strict=False, kernel="march")
    for w, q in zip(want, got):
        assert np.abs(w[I] - q[I]).max() <= TOL[np.float32] * np.abs(w[I]).max()
    sentinel = -777.25
    out = (swmhd.Field(g, dtype=torch.float32), swmhd.Field(g, dtype=torch.float32))
    for o in out:
        o.data.fill_(sentinel)
    part = _run(swmhd, "jacobian", g, f, out=out, strict=False, kernel="march", rows=(5, 17))
    for q, full in zip(part, got):
        assert np.array_equal(q[H + 5:H + 17, H:H + Nx], full[H + 5:H + 17, H:H + Nx])
        mask = np.ones_like(q, dtype=bool)
        mask[H + 5:H + 17, H:H + Nx] = False
        assert np.all(q[mask] == sentinel), "kernel wrote outside the requested rows / into halos"


@pytest.mark.parametrize("name", ["gaussian_128", "two_gaussians_64"])
def test_golden_fixtures(swmhd, name):
    """Committed fixtures (restatement-generated, tests/golden/make_golden.py) -- no oracle call here."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    A, h, d = z["A"], z["h"], float(z["d"])
    N = A.shape[0] - 6
    g, f = _fields(swmhd, N, N, 3, np.ascontiguousarray(A), np.ascontiguousarray(h), d=(d, d))
    Jx, Jy = _run(swmhd, "jacobian", g, f, strict=True)
    Dx, Dy = _run(swmhd, "divergence", g, f, strict=True)
    I = g.interior
    for got, key in ((Jx, "Jx"), (Jy, "Jy"), (Dx, "Dx"), (Dy, "Dy")):
        assert np.array_equal(got[I], z[key][I]), key


def test_analytic_pin_through_the_hip_path(swmhd):
    """test_formulations.jl:14-15,205-210 evaluated by the HIP kernels: 2nd-order convergence to (-4x,-4y)exp(-2r^2)."""
    errs = []
    for N in (64, 128, 256, 512):
        A, h, d, ex, ey = Hh.gaussian_case(N, 3)
        g, f = _fields(swmhd, N, N, 3, A, h, d=(d, d))
        I = g.interior
        Jx, Jy = _run(swmhd, "jacobian", g, f)
        Dx, Dy = _run(swmhd, "divergence", g, f)
        errs.append([np.abs(Jx - ex)[I].max(), np.abs(Jy - ey)[I].max(), np.abs(Dx - ex)[I].max(), np.abs(Dy - ey)[I].max()])
    E = np.array(errs)
    for c in range(4):
        slope = -np.polyfit(np.log10([64, 128, 256, 512]), np.log10(E[:, c]), 1)[0]
        assert 1.9 < slope < 2.1
    assert abs(E[0, 0] - 6.554741068160697e-2) < 1e-12 and abs(E[0, 2] - 8.998312934496977e-2) < 1e-12


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_periodic_halo_fill(swmhd, dtype):
    Nx, Ny, H = 37, 21, 3
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 1), y=(0, 1), halo=(H, H))
    rng = np.random.default_rng(0)
    a = rng.standard_normal(g.parent_shape)
    f = swmhd.Field(g, dtype=dtype, data=torch.from_numpy(a).to(dtype).cuda())
    f.fill_halo_regions()
    torch.cuda.synchronize()
    want = Hh.fill_halo_periodic(a.astype(np.float64 if dtype == torch.float64 else np.float32), Nx, Ny, H, H)
    assert np.array_equal(f.numpy(), want)


@pytest.mark.parametrize("form", ["jacobian", "divergence"])
def test_full_size_properties_4096(swmhd, form):
    """BASELINE size (4096^2, too big for the oracle in seconds): size-independent properties.
    (1) quadratic scaling: F(2A) == 4 F(A) bitwise (power-of-two scaling commutes with every rounding);
    (2) periodic shift equivariance, bitwise; (3) fast vs strict within tolerance."""
    N, H = 4096, 3
    g = swmhd.RectilinearGrid(size=(N, N), x=(-np.pi, np.pi), y=(-10, 10), halo=(H, H))
    gen = torch.Generator(device="cuda").manual_seed(1234)
    A = swmhd.Field(g)
    h = swmhd.Field(g)
    A.data.normal_(generator=gen)
    h.data.uniform_(1.0, 1.3, generator=gen)
    A.fill_halo_regions(); h.fill_halo_regions()
    fn = swmhd.lorentz_force_func if form == "jacobian" else swmhd.div_lorentz
    Fx, Fy = fn(g, {"A": A, "h": h}, strict=True)
    A2 = swmhd.Field(g, data=A.data * 2)
    Gx, Gy = fn(g, {"A": A2, "h": h}, strict=True)
    I = g.interior
    assert torch.equal(Gx.data[I], 4 * Fx.data[I]) and torch.equal(Gy.data[I], 4 * Fy.data[I])
    sy_, sx_ = 1000, 37
    As = swmhd.Field(g); hs = swmhd.Field(g)
    As.data[I] = torch.roll(A.data[I], (sy_, sx_), (0, 1)); hs.data[I] = torch.roll(h.data[I], (sy_, sx_), (0, 1))
    As.fill_halo_regions(); hs.fill_halo_regions()
    Sx, Sy = fn(g, {"A": As, "h": hs}, strict=True)
    assert torch.equal(Sx.data[I], torch.roll(Fx.data[I], (sy_, sx_), (0, 1)))
    assert torch.equal(Sy.data[I], torch.roll(Fy.data[I], (sy_, sx_), (0, 1)))
    Qx, Qy = fn(g, {"A": A, "h": h}, strict=False)
    for s, q in ((Fx, Qx), (Fy, Qy)):
        assert (s.data[I] - q.data[I]).abs().max().item() <= 1e-13 * s.data[I].abs().max().item()
    assert torch.isfinite(Fx.data[I]).all() and Fx.data[I].abs().max().item() > 0


@pytest.mark.parametrize("topo", [("Bounded", "Bounded"), ("Bounded", "Periodic"), ("Periodic", "Bounded")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(70, 50), (20, 18), (4, 3), (130, 9)])
def test_bounded_wall_branches(swmhd, oracle, topo, dtype, shape):
    """sw_mhd_divergence_functions.jl:42-53,66-77,90-101,114-125: `topology(grid, d) == Bounded` branches of the four fluxes.
    Halos hold whatever the caller's boundary conditions put there (random here); strict = bitwise, fast within tolerance."""
    Nx, Ny = shape
    H = 3
    A, h = Hh.random_case(Nx, Ny, H, H, 77 + Nx, dtype, periodic=False)
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 0.37 * Nx), y=(0, 0.41 * Ny), halo=(H, H), topology=(topo[0], topo[1], "Flat"))
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    f = {"A": swmhd.Field(g, dtype=tdt, data=torch.from_numpy(A).cuda()), "h": swmhd.Field(g, dtype=tdt, data=torch.from_numpy(h).cuda())}
    code = {"Periodic": oracle.PERIODIC, "Bounded": oracle.BOUNDED}
    want = oracle.lorentz_divergence(A, h, Nx, Ny, H, H, g.dx, g.dy, topo=(code[topo[0]], code[topo[1]]), nthreads=4)
    periodic = oracle.lorentz_divergence(A, h, Nx, Ny, H, H, g.dx, g.dy, nthreads=4)
    I = g.interior
    got = _run(swmhd, "divergence", g, f, strict=True)
    fast = _run(swmhd, "divergence", g, f, strict=False)
    for w, q, r, p in zip(want, got, fast, periodic):
        assert np.array_equal(w[I], q[I])
        assert np.abs(w[I] - r[I]).max() <= TOL[dtype] * np.abs(w[I]).max()
        assert not np.array_equal(w[I], p[I])      # the wall branches really are exercised


@pytest.mark.parametrize("topo", [("Bounded", "Bounded"), ("Bounded", "Periodic"), ("Periodic", "Bounded")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bounded_wall_branches_large_grid_hybrid_launch(swmhd, oracle, topo, dtype):
    """From ~0.9 Mcell on a Bounded grid takes the marching kernel (periodic branch) over every row and the LDS-tiled kernel with
    the wall branches over the frame near the walls: against the oracle within the fast tolerance everywhere, and bitwise the
    all-tile launch inside the frame."""
    Nx, Ny, H = 1300, 800, 3
    A, h = Hh.random_case(Nx, Ny, H, H, 4242, dtype, periodic=False)
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 0.37 * Nx), y=(0, 0.41 * Ny), halo=(H, H), topology=(topo[0], topo[1], "Flat"))
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    f = {"A": swmhd.Field(g, dtype=tdt, data=torch.from_numpy(A).cuda()), "h": swmhd.Field(g, dtype=tdt, data=torch.from_numpy(h).cuda())}
    code = {"Periodic": oracle.PERIODIC, "Bounded": oracle.BOUNDED}
    want = oracle.lorentz_divergence(A, h, Nx, Ny, H, H, g.dx, g.dy, topo=(code[topo[0]], code[topo[1]]), nthreads=8)
    I = g.interior
    auto = _run(swmhd, "divergence", g, f, strict=False)
    tile = _run(swmhd, "divergence", g, f, strict=False, kernel="tile")
    frame = np.zeros((Ny, Nx), dtype=bool)
    if topo[1] == "Bounded":
        frame[:8] = True; frame[-8:] = True
    if topo[0] == "Bounded":
        frame[:, :64] = True; frame[:, 64 * ((Nx - 1) // 64):] = True
    for w, a, t in zip(want, auto, tile):
        assert np.abs(w[I] - a[I]).max() <= TOL[dtype] * np.abs(w[I]).max()
        assert np.array_equal(a[I][frame], t[I][frame])


def test_error_codes_on_device_pointers(swmhd):
    g = swmhd.RectilinearGrid(size=(16, 16), x=(0, 1), y=(0, 1), halo=(2, 2))
    A, h = swmhd.Field(g), swmhd.Field(g)
    h.data.fill_(1.0)
    swmhd.lorentz_force_func(g, {"A": A, "h": h})          # halo 2 is enough for the Jacobian form
    with pytest.raises(swmhd._lib.SwmhdError, match="halo"):
        swmhd.div_lorentz(g, {"A": A, "h": h})             # ... but not for the divergence form
    torch.cuda.synchronize()


@pytest.mark.parametrize("form", ["jacobian", "divergence"])
@pytest.mark.parametrize("kernel", ["tile", "march"])
def test_zero_depth_propagates_non_finite_only_through_its_stencil(swmhd, oracle, form, kernel):
    """Error behaviour of the reference's forcing functions: none -- a dry cell (h -> 0) just propagates Inf/NaN
    (SURVEY.md 8(b)).  Same here: no trap, no hang; cells outside the stencil footprint of the dry cell are untouched and still
    match the oracle; inside it both produce non-finite values (which of Inf / NaN differs: Newton reciprocal vs IEEE divide)."""
    Nx, Ny, H = 300, 60, 3
    A, h = Hh.random_case(Nx, Ny, H, H, 5)
    h[H + 30, H + 150] = 0.0
    h[H + 30, H + 151] = 0.0      # two neighbours, so that the face-averaged depth of the divergence form vanishes too
    g, f = _fields(swmhd, Nx, Ny, H, A, h)
    want = _oracle(oracle, form, A, h, g)
    got = _run(swmhd, form, g, f, strict=False, kernel=kernel)
    I = g.interior
    bad_hip = bad_oracle = False
    for w, q in zip(want, got):
        far = np.ones((Ny, Nx), dtype=bool)
        far[30 - 4:30 + 5, 150 - 4:151 + 5] = False
        wi, qi = w[I], q[I]
        assert np.isfinite(qi[far]).all()
        assert np.abs(wi[far] - qi[far]).max() <= 1e-13 * np.abs(wi[far]).max()
        bad_hip |= bool((~np.isfinite(qi)).any())
        bad_oracle |= bool((~np.isfinite(wi)).any())
    assert bad_hip and bad_oracle      # both propagate Inf/NaN inside the footprint (in at least one component)
