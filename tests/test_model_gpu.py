"""GPU parity tests of the fused tendency engine (HIP, through the C-ABI) against the CPU oracle.

Bars: SWMHD_STRICT bit-identical to the oracle (which for the base RHS is a PARITY-UNPINNED restatement of Oceananigans'
scheme, and for the forcing restates the reference's own code); fast path max|dG| <= 1e-12 * max|G| (fp64) per RHS
evaluation, <= 1e-4 * max|G| (fp32: WENO smoothness weights amplify fp32 rounding)."""
import numpy as np
import pytest
import torch

import helpers as Hh
from test_model_oracle import staggered_fields, G, F

pytestmark = pytest.mark.gpu
CASES = [(1, 1), (1, 0), (0, 2), (0, 0)]   # (formulation, lorentz)
FORM = {0: "Conservative", 1: "VectorInvariant"}


def random_state(Nx, Ny, H, seed, form, dtype=np.float64):
    rng = np.random.default_rng(seed)
    shp = (Ny + 2 * H, Nx + 2 * H)
    h = 1.0 + 0.3 * rng.random(shp)
    u, v = 0.5 * rng.standard_normal(shp), 0.5 * rng.standard_normal(shp)
    A = rng.standard_normal(shp)
    q1, q2 = (u, v) if form == 1 else (h * u, h * v)
    return [np.ascontiguousarray(Hh.fill_halo_periodic(a, Nx, Ny, H, H).astype(dtype)) for a in (q1, q2, h, A)]


def make_model(S, Nx, Ny, form, lor, q, dx, dy, strict, dtype=torch.float64, fused=True):
    g = S.RectilinearGrid(size=(Nx, Ny), x=(0, dx * Nx), y=(0, dy * Ny), halo=(3, 3))
    m = S.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=bool(lor), dtype=dtype, strict=strict, fused=fused)
    for f, a in zip(m.fields, q):
        f.data.copy_(torch.from_numpy(a))
    return m


@pytest.mark.parametrize("form,lor", CASES)
@pytest.mark.parametrize("shape", [(64, 8), (100, 37), (7, 5), (130, 70), (256, 64)])
def test_tendencies_strict_bitwise_fast_within_tolerance(swmhd, oracle, form, lor, shape):
    Nx, Ny = shape
    q = random_state(Nx, Ny, 3, 7 + Nx, form)
    dx, dy = 0.11, 0.13
    want = oracle.tendencies(*q, Nx, Ny, 3, 3, dx, dy, form, lor, G, F, nthreads=8)
    m = make_model(swmhd, Nx, Ny, form, lor, q, dx, dy, strict=True)
    m.calculate_tendencies(); torch.cuda.synchronize()
    I = m.grid.interior
    for w, gf in zip(want, m.Gn):
        got = gf.numpy()
        assert np.array_equal(w[I], got[I]), f"strict differs: max {np.abs(w[I] - got[I]).max()} of {np.abs(w[I]).max()}"
    m2 = make_model(swmhd, Nx, Ny, form, lor, q, dx, dy, strict=False)
    m2.calculate_tendencies(); torch.cuda.synchronize()
    for w, gf in zip(want, m2.Gn):
        assert np.abs(w[I] - gf.numpy()[I]).max() <= 1e-12 * np.abs(w[I]).max()


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
def test_smooth_fields_and_pde_convergence_on_gpu(swmhd, form, lor):
    """The HIP engine itself converges to the continuous equations (no oracle involved)."""
    from test_model_oracle import pde_rhs, Lx, Ly
    errs = []
    for N in (64, 128):
        q, (cc, fc, cf), dx, dy = staggered_fields(N, form)
        m = make_model(swmhd, N, N, form, lor, q, dx, dy, strict=False)
        m.calculate_tendencies(); torch.cuda.synchronize()
        I = m.grid.interior
        r1 = pde_rhs(fc[0][I], fc[1][I], form, lor)[0]
        rc = pde_rhs(cc[0][I], cc[1][I], form, lor)
        errs.append([np.abs(m.Gn[0].numpy()[I] - r1).max() / np.abs(r1).max(), np.abs(m.Gn[3].numpy()[I] - rc[3]).max() / np.abs(rc[3]).max()])
    E = np.array(errs)
    assert np.all(np.log2(E[0] / E[1]) > 1.9)


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
def test_fp32_tendencies(swmhd, oracle, form, lor):
    N = 96
    q, _, dx, dy = staggered_fields(N, form)
    q32 = [a.astype(np.float32) for a in q]
    want = oracle.tendencies(*q32, N, N, 3, 3, dx, dy, form, lor, G, F, nthreads=8)
    ref64 = oracle.tendencies(*q, N, N, 3, 3, dx, dy, form, lor, G, F, nthreads=8)
    m = make_model(swmhd, N, N, form, lor, q32, dx, dy, strict=True, dtype=torch.float32)
    m.calculate_tendencies(); torch.cuda.synchronize()
    I = m.grid.interior
    for w, gf in zip(want, m.Gn):
        assert np.array_equal(w[I], gf.numpy()[I])
    m2 = make_model(swmhd, N, N, form, lor, q32, dx, dy, strict=False, dtype=torch.float32)
    m2.calculate_tendencies(); torch.cuda.synchronize()
    for w, gf in zip(ref64, m2.Gn):
        assert np.abs(w[I] - gf.numpy()[I]).max() <= 1e-4 * np.abs(w[I]).max()


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
def test_row_ranges_compose(swmhd, form, lor):
    """Interior rows + two boundary strips (the overlap split of SURVEY.md 8(e)) == one full launch, bitwise."""
    Nx, Ny = 90, 40
    q = random_state(Nx, Ny, 3, 3, form)
    m = make_model(swmhd, Nx, Ny, form, lor, q, 0.1, 0.1, strict=False)
    m.calculate_tendencies(); torch.cuda.synchronize()
    full = [g_.data.clone() for g_ in m.Gn]
    for g_ in m.Gn:
        g_.data.fill_(-1.5)
    m.calculate_tendencies(rows=(3, Ny - 3)); m.calculate_tendencies(rows=(0, 3)); m.calculate_tendencies(rows=(Ny - 3, Ny))
    torch.cuda.synchronize()
    I = m.grid.interior
    for a, b in zip(full, m.Gn):
        assert torch.equal(a[I], b.data[I])
        halo = b.data.clone(); halo[I] = -1.5
        assert torch.all(halo == -1.5)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
def test_time_steps_match_oracle(swmhd, oracle, form, lor, fused):
    """Three full RK3 steps (9 tendency evaluations, 9 substeps, 9 halo fills): strict bitwise vs the oracle's time_step,
    for the fused one-kernel-per-stage path (ping-ponged state) and for the separate tendencies + rk3_substep path."""
    N = 48
    q, _, dx, dy = staggered_fields(N, form)
    q = [oracle.fill_halo_periodic(a, N, N, 3, 3) for a in q]
    m = make_model(swmhd, N, N, form, lor, q, dx, dy, strict=True, fused=fused)
    mf = make_model(swmhd, N, N, form, lor, q, dx, dy, strict=False, fused=fused)
    dt = 0.002
    qo = [a.copy() for a in q]
    for _ in range(3):
        oracle.time_step(*qo, N, N, 3, 3, dx, dy, dt, form, lor, G, F, nthreads=8)
        m.time_step(dt); mf.time_step(dt)
    m.synchronize()
    for w, f in zip(qo, m.fields):
        assert np.array_equal(w, f.numpy()), "strict time stepping (incl. halos) differs from the oracle"
    for w, f in zip(qo, mf.fields):
        assert np.abs(w - f.numpy()).max() <= 1e-12 * np.abs(w).max()
    assert m.iteration == 3 and abs(m.clock_time - 3 * dt) < 1e-15


@pytest.mark.parametrize("form,lor", [(1, 1), (1, 0), (0, 2), (0, 0)])
@pytest.mark.parametrize("shape", [(250, 32), (251, 33), (600, 70), (40, 9), (1024, 256)])
def test_marching_kernel_agrees_with_tile_kernel(swmhd, oracle, form, lor, shape):
    """The row-marching kernels (default fast path of both formulations) vs the LDS-tiled kernel on the same
    inputs: same arithmetic up to rounding, incl. strips that end mid-workgroup and segments shorter than LY."""
    Nx, Ny = shape
    q = random_state(Nx, Ny, 3, 21 + Nx, form)
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 0.11 * Nx), y=(0, 0.13 * Ny), halo=(3, 3))
    out = []
    for kern in ("march", "tile"):
        m = swmhd.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=bool(lor), kernel=kern)
        for f, a in zip(m.fields, q):
            f.data.copy_(torch.from_numpy(a))
        for g_ in m.Gn:
            g_.data.fill_(-3.25)
        m.calculate_tendencies(); torch.cuda.synchronize()
        out.append([g_.data.clone() for g_ in m.Gn])
    I = g.interior
    for other in out[2:] + [out[0]]:          # marching (and wave-specialised) kernels vs the tile kernel (out[1])
        for a, b in zip(other, out[1]):
            assert (a[I] - b[I]).abs().max().item() <= 1e-12 * b[I].abs().max().item()
            halo = a.clone(); halo[I] = -3.25
            assert torch.all(halo == -3.25), "marching kernel wrote outside the interior"
    if Nx * Ny <= 600 * 70:
        want = oracle.tendencies(*q, Nx, Ny, 3, 3, g.dx, g.dy, form, lor, G, F, nthreads=8)
        for w, a in zip(want, out[0]):
            assert np.abs(w[I] - a.cpu().numpy()[I]).max() <= 1e-12 * np.abs(w[I]).max()


def test_halo_multi_matches_numpy(swmhd):
    Nx, Ny, H = 33, 17, 3
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 1), y=(0, 1), halo=(H, H))
    m = swmhd.ShallowWaterModel(g)
    rng = np.random.default_rng(5)
    raw = [rng.standard_normal(g.parent_shape) for _ in range(4)]
    for f, a in zip(m.fields, raw):
        f.data.copy_(torch.from_numpy(a))
    m.update_state(); torch.cuda.synchronize()
    for f, a in zip(m.fields, raw):
        assert np.array_equal(f.numpy(), Hh.fill_halo_periodic(a, Nx, Ny, H, H))


def test_model_rejects_bad_arguments(swmhd):
    L = swmhd._lib.lib()
    g = swmhd.RectilinearGrid(size=(16, 16), x=(0, 1), y=(0, 1), halo=(3, 3))
    m = swmhd.ShallowWaterModel(g)
    q = m.fields
    f = L.swmhd_tendencies_f64
    args = lambda form, lor, H: (q[0].ptr, q[1].ptr, q[2].ptr, q[3].ptr, m.Gn[0].ptr, m.Gn[1].ptr, m.Gn[2].ptr, m.Gn[3].ptr,
                                 16, 16, H, H, q[0].stride_y, 0.1, 0.1, 9.81, 1.0, form, lor, 0, 16, 0, None)
    assert f(*args(0, 1, 3)) == 1      # Jacobian forcing with the conservative formulation
    assert f(*args(1, 2, 3)) == 1      # divergence forcing with the vector-invariant formulation
    assert f(*args(1, 1, 2)) == 2      # halo 2 < 3
    assert f(*args(1, 1, 3)) == 0
    torch.cuda.synchronize()


def test_checkpoint_round_trip_is_bitwise(swmhd, tmp_path):
    """Dump/restore incl. halos: 4 steps == 2 steps + save/load into a fresh model + 2 steps."""
    N = 40
    q, _, dx, dy = staggered_fields(N, 1)
    q = [Hh.fill_halo_periodic(a, N, N, 3, 3) for a in q]
    a = make_model(swmhd, N, N, 1, 1, q, dx, dy, strict=False)
    b = make_model(swmhd, N, N, 1, 1, q, dx, dy, strict=False)
    for _ in range(4):
        a.time_step(0.002)
    for _ in range(2):
        b.time_step(0.002)
    b.save_checkpoint(tmp_path / "ck.npz")
    c = make_model(swmhd, N, N, 1, 1, q, dx, dy, strict=False).load_checkpoint(tmp_path / "ck.npz")
    for _ in range(2):
        c.time_step(0.002)
    a.synchronize(); c.synchronize()
    for fa, fc in zip(a.fields, c.fields):
        assert torch.equal(fa.data, fc.data)
    assert c.iteration == 4
    fa = a.fields[2]
    fa.save(tmp_path / "h.npy")
    fresh = swmhd.Field(a.grid).load(tmp_path / "h.npy")
    assert torch.equal(fresh.data, fa.data)


@pytest.mark.parametrize("form,shape", [("VectorInvariant", (16384, 2048)), ("Conservative", (8192, 1024))])
def test_baseline_slab_sizes_size_independent_properties(swmhd, form, shape):
    """BASELINE configs 4 / 5 per-GPU slab sizes (8192x1024, 16384x2048: far beyond what the oracle finishes in seconds):
    size-independent properties of the fused engine -- mass is conserved to rounding by the flux form, the advected tracer's
    extrema do not grow, x-translation of the initial state translates the result (periodic equivariance), everything finite."""
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    Nx, Ny = shape
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, Lx), y=(0, Ly))
    outs = []
    for shift in (0, 517):
        sx = shift * g.dx
        m = swmhd.ShallowWaterModel(g, 9.81, 1.0, formulation=form)
        h_ = lambda X, Y: hf(X - sx, Y)
        if form == "VectorInvariant":
            m.set(u=lambda X, Y: uf(X - sx, Y), v=lambda X, Y: vf(X - sx, Y), h=h_, A=lambda X, Y: Af(X - sx, Y))
        else:
            m.set(uh=lambda X, Y: h_(X, Y) * uf(X - sx, Y), vh=lambda X, Y: h_(X, Y) * vf(X - sx, Y), h=h_, A=lambda X, Y: Af(X - sx, Y))
        d0 = m.diagnostics()
        h_sum0 = m.solution["h"].data[g.interior].sum().item()
        for _ in range(2):
            m.time_step(2e-5)
        m.synchronize()
        d1 = m.diagnostics()
        assert all(np.isfinite(v) for v in d1.values())
        assert abs(m.solution["h"].data[g.interior].sum().item() - h_sum0) <= 1e-12 * abs(h_sum0)
        assert d1["max_abs_A"] <= d0["max_abs_A"] * (1 + 1e-6)     # sampled maximum of a smooth advected field
        assert abs(d1["total_energy"] - d0["total_energy"]) <= 1e-4 * d0["total_energy"]
        outs.append([f.data[g.interior].clone() for f in m.fields])
        del m
    for a, b in zip(*outs):
        ref = torch.roll(a, 517, 1)
        assert (ref - b).abs().max().item() <= 1e-11 * ref.abs().max().item()


@pytest.mark.parametrize("form", [1, 0])
def test_graph_replay_equals_eager(swmhd, form):
    """HIP-graph replay of the step (capture_graph / time_steps) is bit-identical to eager stepping, on the reference's own
    grid size (64 x 64, SWMHD_example.jl:11)."""
    N = 64
    q, _, dx, dy = staggered_fields(N, form)
    q = [Hh.fill_halo_periodic(a, N, N, 3, 3) for a in q]
    a = make_model(swmhd, N, N, form, 2 - form, q, dx, dy, strict=False)
    b = make_model(swmhd, N, N, form, 2 - form, q, dx, dy, strict=False)
    dt = 0.002
    a.time_step(dt); b.time_step(dt)            # iteration 0 takes the eager path in both
    b.capture_graph(dt)
    for _ in range(9):
        a.time_step(dt)
    b.time_steps(9, dt)                          # 4 replays + 1 eager step
    a.synchronize(); b.synchronize()
    for fa, fb in zip(a.fields, b.fields):
        assert torch.equal(fa.data, fb.data)
    assert a.iteration == b.iteration == 10 and abs(a.clock_time - b.clock_time) < 1e-15


@pytest.mark.parametrize("form", [1, 0])
def test_graph_replay_after_odd_step_counts(swmhd, form):
    """The captured graph has one role assignment of the ping-pong buffers baked in; every odd number of eager steps flips the
    roles.  Mixed sequences (odd leftovers, plain time_step calls, capture at iteration 0) must stay bit-identical to eager
    stepping: capture, time_steps(9), time_steps(4), time_step(), time_steps(2), time_steps(5), time_steps(5)."""
    N = 64
    q, _, dx, dy = staggered_fields(N, form)
    q = [Hh.fill_halo_periodic(a, N, N, 3, 3) for a in q]
    a = make_model(swmhd, N, N, form, 2 - form, q, dx, dy, strict=False)
    b = make_model(swmhd, N, N, form, 2 - form, q, dx, dy, strict=False)
    dt = 0.002
    b.capture_graph(dt)                          # at iteration 0
    total = 0
    for n, plain in ((9, False), (4, False), (1, True), (2, False), (5, False), (5, False), (1, False), (3, False)):
        if plain:
            b.time_step(dt)
        else:
            b.time_steps(n, dt)
        for _ in range(n):
            a.time_step(dt)
        total += n
        a.synchronize(); b.synchronize()
        for fa, fb in zip(a.fields, b.fields):
            assert torch.equal(fa.data, fb.data), f"graph path diverged after {total} steps"
    assert a.iteration == b.iteration == total


@pytest.mark.parametrize("form", [1, 0])
@pytest.mark.parametrize("nsteps", [1, 4])
def test_native_step_driver_equals_python_driven_stages(swmhd, form, nsteps):
    """swmhd_step_rk3_* (one C call enqueues 3 fused stages + halo fills per step) == ShallowWaterModel.time_step, bitwise,
    for odd and even step counts (state ends in the alternate / original buffers)."""
    N = 72
    q, _, dx, dy = staggered_fields(N, form)
    q = [Hh.fill_halo_periodic(a, N, N, 3, 3) for a in q]
    for strict in (True, False):
        a = make_model(swmhd, N, N, form, 2 - form, q, dx, dy, strict=strict)
        b = make_model(swmhd, N, N, form, 2 - form, q, dx, dy, strict=strict)
        for _ in range(nsteps):
            a.time_step(0.002)
        b.time_steps(nsteps, 0.002)
        a.synchronize(); b.synchronize()
        for fa, fb in zip(a.fields, b.fields):
            assert torch.equal(fa.data, fb.data)
        a.time_step(0.002); b.time_step(0.002)      # G- bookkeeping stays consistent afterwards
        a.synchronize(); b.synchronize()
        for fa, fb in zip(a.fields, b.fields):
            assert torch.equal(fa.data, fb.data)
        assert b.iteration == nsteps + 1


@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
def test_marching_kernels_are_deterministic_and_stable_at_4096(swmhd, form):
    """Race detector by proxy: the marching kernels hand data between waves through LDS rings guarded by
    two barriers per row; a missed hazard would show up as run-to-run differences.  Two models, same initial state, 12 RK3 steps at
    the BASELINE 4096^2 size: bit-identical; and a 60-step run stays finite with total energy within 1e-3 (the current sheets of the
    benchmark initial condition are dissipated by the upwinding)."""
    from swmhd_amd import configs
    N = 4096
    cfg = configs.config3_bickley() if form == "VectorInvariant" else configs.config4_two_gaussians()
    g = swmhd.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
    ms = []
    for _ in range(2):
        m = swmhd.ShallowWaterModel(g, 9.81, 1.0, formulation=form)
        n1, n2 = m.names[:2]
        m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
        ms.append(m)
    e0 = ms[0].diagnostics()["total_energy"]
    for m in ms:
        for _ in range(12):
            m.time_step(1e-4)
        m.synchronize()
    for fa, fb in zip(ms[0].fields, ms[1].fields):
        assert torch.equal(fa.data, fb.data)
    ms[0].time_steps(48, 1e-4)
    ms[0].synchronize()
    d = ms[0].diagnostics()
    assert all(np.isfinite(v) for v in d.values())
    assert abs(d["total_energy"] - e0) <= 1e-3 * abs(e0)


@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
@pytest.mark.parametrize("halo", [(4, 5), (5, 3)])
def test_wider_halos_and_pitched_rows(swmhd, oracle, form, lor, halo):
    """Hx != Hy > 3 and a row pitch larger than Nx+2Hx (the C-ABI's stride_y): every fast kernel and the strict kernel, Lorentz
    operators and fused tendencies, against the oracle evaluated on a dense copy."""
    Nx, Ny = 300, 41
    Hx, Hy = halo
    rng = np.random.default_rng(17)
    shp = (Ny + 2 * Hy, Nx + 2 * Hx)
    h = 1.0 + 0.3 * rng.random(shp); u = 0.5 * rng.standard_normal(shp); v = 0.5 * rng.standard_normal(shp); A = rng.standard_normal(shp)
    q = [u, v, h, A] if form == 1 else [h * u, h * v, h, A]
    q = [np.ascontiguousarray(Hh.fill_halo_periodic(a, Nx, Ny, Hx, Hy)) for a in q]
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 0.11 * Nx), y=(0, 0.13 * Ny), halo=(Hx, Hy))
    W, pitch = shp[1], shp[1] + 7

    def pitched(a=None):
        buf = torch.full((shp[0], pitch), -9.5, dtype=torch.float64, device="cuda")
        view = buf[:, :W]
        if a is not None:
            view.copy_(torch.from_numpy(a))
        return swmhd.Field(g, data=view), buf

    want = oracle.tendencies(*q, Nx, Ny, Hx, Hy, g.dx, g.dy, form, lor, G, F, nthreads=8)
    opw = (oracle.lorentz_jacobian if form == 1 else oracle.lorentz_divergence)(q[3], q[2], Nx, Ny, Hx, Hy, g.dx, g.dy, nthreads=8)
    I = g.interior
    for kern, strict in (("tile", True), ("tile", False), ("march", False)):
        m = swmhd.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=True, strict=strict, kernel=kern, fuse_halo=False)
        bufs = []
        for name, a in zip(m.names, q):
            f, b = pitched(a); m.solution[name] = f; bufs.append(b)
        for k in range(4):
            m.Gn[k], b = pitched(); bufs.append(b)
        assert m.fields[0].stride_y == pitch
        m.calculate_tendencies(); torch.cuda.synchronize()
        for w, gf in zip(want, m.Gn):
            got = gf.data.cpu().numpy()
            if strict:
                assert np.array_equal(w[I], got[I])
            else:
                assert np.abs(w[I] - got[I]).max() <= 1e-12 * np.abs(w[I]).max(), kern
        for b in bufs:
            assert torch.all(b[:, W:] == -9.5), "kernel wrote into the row padding"
        op = swmhd.lorentz_force_func if form == 1 else swmhd.div_lorentz
        out = (pitched()[0], pitched()[0])
        op(g, {"A": m.solution["A"], "h": m.solution["h"]}, out=out, strict=strict, kernel=kern)
        torch.cuda.synchronize()
        for w, of in zip(opw, out):
            got = of.data.cpu().numpy()
            assert np.array_equal(w[I], got[I]) if strict else np.abs(w[I] - got[I]).max() <= 1e-13 * np.abs(w[I]).max()


@pytest.mark.parametrize("tag,form,lor", [("vi", 1, 1), ("cons", 0, 2)])
def test_committed_model_fixture(swmhd, tag, form, lor):
    """tests/golden/model_48x40.npz (restatement-generated, A9 parity unpinned): the HIP engine alone -- no oracle call --
    reproduces the committed tendencies and the state after two RK3 steps bit for bit (strict) / within tolerance (fast)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_48x40.npz"))
    Nx, Ny, dx, dy, dt = int(z["Nx"]), int(z["Ny"]), float(z["dx"]), float(z["dy"]), float(z["dt"])
    q = [np.ascontiguousarray(a) for a in z[f"{tag}_q"]]
    for strict in (True, False):
        m = make_model(swmhd, Nx, Ny, form, lor, q, dx, dy, strict=strict)
        m.calculate_tendencies(); torch.cuda.synchronize()
        I = m.grid.interior
        for w, gf in zip(z[f"{tag}_G"], m.Gn):
            if strict:
                assert np.array_equal(w[I], gf.numpy()[I])
            else:
                assert np.abs(w[I] - gf.numpy()[I]).max() <= 1e-12 * np.abs(w[I]).max()
        m.time_step(dt); m.time_step(dt); m.synchronize()
        for w, f in zip(z[f"{tag}_after2"], m.fields):
            if strict:
                assert np.array_equal(w, f.numpy())          # halos included
            else:
                assert np.abs(w - f.numpy()).max() <= 1e-12 * max(np.abs(w).max(), 1.0)


@pytest.mark.parametrize("form", [1, 0])
def test_byte_offsets_beyond_2GiB(swmhd, form):
    """Parents of 2.16 GiB (16384 x 16400 fp64): the marching kernels address with 32-bit BYTE offsets and store through buffer
    descriptors whose num_records exceeds 2^31 -- every top row of the field lies beyond offset 2^31.  Fused stage (reads G-,
    stores G and the new state) through the marching kernel vs the LDS-tiled kernel (64-bit addressing), compared on the GPU;
    inputs generated on the GPU (no multi-GiB host arrays)."""
    Nx, Ny, H = 16384, 16400, 3
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 2 * np.pi), y=(0, 2 * np.pi), halo=(H, H))
    assert (Ny + 2 * H) * (Nx + 2 * H) * 8 > 2 ** 31
    gen = torch.Generator(device="cuda").manual_seed(5)
    yy = torch.arange(Ny + 2 * H, device="cuda", dtype=torch.float64).reshape(-1, 1) * (2 * np.pi / Ny)
    xx = torch.arange(Nx + 2 * H, device="cuda", dtype=torch.float64).reshape(1, -1) * (2 * np.pi / Nx)
    base = [0.4 * torch.sin(xx) * torch.cos(2 * yy) + 0.2, 0.3 * torch.cos(2 * xx) * torch.sin(yy) - 0.1,
            1.0 + 0.2 * torch.sin(xx + 0.3) * torch.cos(yy), 0.3 * torch.sin(xx) * torch.sin(yy - 0.5)]
    res = []
    for kern in ("march", "tile"):
        m = swmhd.ShallowWaterModel(g, 9.81, 1.0, formulation=FORM[form], kernel=kern, fuse_halo=False)
        for f, b in zip(m.fields, base):
            f.data.copy_(b)
        if form == 0:
            m.fields[0].data.mul_(m.fields[2].data); m.fields[1].data.mul_(m.fields[2].data)
        for f in m.Gm:
            f.data.copy_(0.01 * torch.cos(xx + yy))
        m.update_state()
        m._stage_fused(1e-5, 1)            # stage 2 of RK3: reads G-, stores G and the new state
        torch.cuda.synchronize()
        I = g.interior
        res.append([f.data[I][-64:].clone() for f in m.Gn] + [m._alt[n].data[I][-64:].clone() for n in m.names]
                   + [f.data[I][:64].clone() for f in m.Gn])
        # nothing outside the interior of the outputs was touched (dropped stores really are dropped)
        for f in m.Gn:
            assert f.data[:H].abs().max().item() == 0 and f.data[-H:].abs().max().item() == 0
            assert f.data[:, :H].abs().max().item() == 0 and f.data[:, -H:].abs().max().item() == 0
        del m
        torch.cuda.empty_cache()
    for a, b in zip(*res):      # (dx = 3.8e-4 amplifies rounding in the gradients: an addressing error would be O(1), not 1e-11)
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() <= 2e-11 * max(b.abs().max().item(), 1.0)


@pytest.mark.parametrize("lor", [1, 0])
@pytest.mark.parametrize("shape", [(512, 40), (1000, 64), (1024, 256), (600, 70), (2048, 33)])
def test_packed_fp32_marching_kernel(swmhd, oracle, lor, shape):
    """Config 5's fp32 leg runs on k_tendency_vi_march_pk (two columns per lane, v_pk_* arithmetic; taken when x is read with periodic
    wrapping and Nx is even).  Same bars as the unpacked fp32 kernel: tendencies within 1e-4 max-norm of the fp32 oracle; and it agrees
    with the unpacked kernel (model with fuse_halo=False: no SWMHD_WRAP_X, so the launcher keeps one column per lane) to fp32 rounding,
    for a tendency evaluation and for three fused RK3 steps (all stage variants: first / with G- / last)."""
    Nx, Ny = shape
    q = random_state(Nx, Ny, 3, 5 + Nx, 1, dtype=np.float32)
    dx, dy = 0.11, 0.13
    want = oracle.tendencies(*q, Nx, Ny, 3, 3, dx, dy, 1, lor, G, F, nthreads=8)
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, dx * Nx), y=(0, dy * Ny), halo=(3, 3))
    assert swmhd._lib.tendency_launch_geometry(Nx, Ny, 1, 4, swmhd._lib.MARCH_KERNEL | swmhd._lib.WRAP_X)["kind"] == 3
    assert swmhd._lib.tendency_launch_geometry(Nx, Ny, 1, 4, swmhd._lib.MARCH_KERNEL)["kind"] == 2
    ms = []
    for fuse_halo in (True, False):
        m = swmhd.ShallowWaterModel(g, G, F, formulation="VectorInvariant", lorentz_forcing=bool(lor), dtype=torch.float32,
                                    kernel="march", fuse_halo=fuse_halo)
        for f, a in zip(m._raw_fields, q):
            f.data.copy_(torch.from_numpy(a))
        for g_ in m.Gn:
            g_.data.fill_(-3.25)
        m.calculate_tendencies(); torch.cuda.synchronize()
        ms.append(m)
    I = g.interior
    for w, gp, gu in zip(want, ms[0].Gn, ms[1].Gn):
        ref = np.abs(w[I]).max()
        assert np.abs(w[I] - gp.numpy()[I]).max() <= 1e-4 * ref
        assert np.abs(gu.numpy()[I] - gp.numpy()[I]).max() <= 2e-5 * ref
        halo = gp.data.clone(); halo[I] = -3.25
        assert torch.all(halo == -3.25), "packed kernel wrote outside the interior"
    for m in ms:
        for _ in range(3):
            m.time_step(1e-3)
        m.synchronize()
    for a, b in zip(ms[0].fields, ms[1].fields):
        assert (a.data[I] - b.data[I]).abs().max().item() <= 2e-5 * max(b.data[I].abs().max().item(), 1.0)
        assert torch.isfinite(a.data).all()


def test_timing_events_without_system_fence(swmhd):
    """swmhd_event_*: the timing events bench.py brackets every stage launch with (hipEventDisableSystemFence)."""
    g = swmhd.RectilinearGrid(size=(512, 512), x=(-5, 5), y=(-5, 5))
    m = swmhd.ShallowWaterModel(g, formulation="VectorInvariant")
    m.set(u=lambda X, Y: 0.01 * Y, v=lambda X, Y: 0 * X, h=lambda X, Y: 1 + 0 * X, A=lambda X, Y: np.exp(-(X ** 2 + Y ** 2)))
    m.tendency_events = []
    m.time_step(1e-3)
    m.synchronize()
    assert len(m.tendency_events) == 3
    for a, b, rows in m.tendency_events:
        assert rows == 512 and 0.0 < a.elapsed_time(b) < 50.0      # ms
    e0, e1 = swmhd._lib.TimingEvent(), swmhd._lib.TimingEvent()
    e0.record(); e1.record()
    assert 0.0 <= e0.elapsed_time(e1) < 10.0


@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
@pytest.mark.parametrize("kernel", ["strict", "tile", "march"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_rows_reaching_into_a_deep_halo(swmhd, form, kernel, dtype):
    """swmhd_tendencies_rk3 with Hy = 9: the row range may reach 6 rows into the y halo; those rows get what their periodic images
    get.  Compared with the same call over the interior only: extended rows == the periodic images of interior rows (bitwise: the
    same kernel and arithmetic), rows further out untouched, and out-of-range requests are refused."""
    import ctypes
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    L = swmhd._lib
    Nx, Ny, H = (600, 80, 9) if kernel == "march" else (130, 40, 9)
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, Lx), y=(0, Ly), halo=(3, H))
    m = swmhd.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=(kernel == "strict"), dtype=dtype)
    if form == "VectorInvariant":
        m.set(u=uf, v=vf, h=hf, A=Af)
    else:
        m.set(uh=lambda X, Y: hf(X, Y) * uf(X, Y), vh=lambda X, Y: hf(X, Y) * vf(X, Y), h=hf, A=Af)
    q = [f for f in m.fields]                      # halos current (x and y, depth 9)
    sfx = "f64" if dtype == torch.float64 else "f32"
    fn = getattr(L.lib(), f"swmhd_tendencies_rk3_{sfx}")
    flags = {"strict": L.STRICT, "tile": L.TILE_KERNEL, "march": L.MARCH_KERNEL}[kernel]
    sentinel = -777.25

    def run(j0, j1, fl=flags):
        new = [swmhd.Field(g, dtype=dtype) for _ in range(4)]
        G = [swmhd.Field(g, dtype=dtype) for _ in range(4)]
        for f in new + G:
            f.data.fill_(sentinel)
        rc = fn(L.ptr_array([f.ptr for f in q]), L.ptr_array([f.ptr for f in new]), L.ptr_array([f.ptr for f in G]), None,
                Nx, Ny, 3, H, q[0].stride_y, g.dx, g.dy, 9.81, 1.0, m.form_code, m.lorentz_code, 1e-3, 8.0 / 15.0, 0.0, 1, j0, j1,
                fl, None)
        torch.cuda.synchronize()
        return rc, [f.numpy() for f in new], [f.numpy() for f in G]

    rc, new_i, G_i = run(0, Ny)
    assert rc == 0
    rc, new_e, G_e = run(-6, Ny + 6)
    assert rc == 0
    # (the fast marching kernel carries y-fluxes from row to row and forms them directly in a segment's prologue: which of the two
    #  a row gets depends on where the segments start, so there the comparison is to rounding, not bitwise)
    tol = 0.0 if kernel != "march" else (1e-12 if dtype == torch.float64 else 2e-5)
    same = lambda a, b: np.array_equal(a, b) if tol == 0.0 else np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1.0)
    for full, ext in zip(new_i + G_i, new_e + G_e):
        I = (slice(H, H + Ny), slice(3, 3 + Nx))
        assert same(full[I], ext[I])
        assert same(ext[H - 6:H, 3:3 + Nx], full[H + Ny - 6:H + Ny, 3:3 + Nx])      # south halo rows = images of the top rows
        assert same(ext[H + Ny:H + Ny + 6, 3:3 + Nx], full[H:H + 6, 3:3 + Nx])
        assert np.all(ext[:H - 6] == sentinel) and np.all(ext[H + Ny + 6:] == sentinel)
        assert np.all(ext[:, :3] == sentinel) and np.all(ext[:, 3 + Nx:] == sentinel)
    EINVAL = 1
    assert run(-7, Ny)[0] == EINVAL and run(0, Ny + 7)[0] == EINVAL
    assert run(-1, Ny, flags | L.WRAP_Y)[0] == EINVAL              # wrapped y: no rows outside the interior


@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("N", [96, 1024])
def test_second_stage_with_the_previous_state_as_operand(swmhd, form, dtype, N):
    """SWMHD_GM_IS_PREV_STATE (swmhd.h): the second RK3 stage given U0 instead of G0 -- U1 = U0 + dt gamma1 G0, zeta passed as
    zeta2 / gamma1, U0 living in the buffer the stage writes U2 to -- equals the classic call within an ulp of the state, on the
    LDS-tiled (N = 96) and the row-marching / packed-fp32 (N = 1024) kernels; and it is refused for strict and Bounded calls."""
    import ctypes
    S, L = swmhd, swmhd._lib
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    g = S.RectilinearGrid(size=(N, N), x=(0, Lx), y=(0, Ly))
    m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, dtype=dtype)
    if form == "VectorInvariant":
        m.set(u=uf, v=vf, h=hf, A=Af)
    else:
        m.set(uh=lambda X, Y: hf(X, Y) * uf(X, Y), vh=lambda X, Y: hf(X, Y) * vf(X, Y), h=hf, A=Af)
    dt = 2e-4 if N > 512 else 2e-3
    sfx = m.sfx
    f = getattr(L.lib(), f"swmhd_tendencies_rk3_{sfx}")
    mk = lambda: [S.Field(g, dtype=dtype) for _ in range(4)]
    U0 = [x for x in m.fields]
    U1, G0, G1a, G1b, U2a = mk(), mk(), mk(), mk(), mk()
    P = lambda fl: L.ptr_array([x.ptr for x in fl])
    flags = L.WRAP_X | L.WRAP_Y
    args = (g.Nx, g.Ny, g.Hx, g.Hy, U0[0].stride_y, g.dx, g.dy, 9.81, 1.0, m.form_code, m.lorentz_code)
    g1, g2, z2 = 8.0 / 15.0, 5.0 / 12.0, -17.0 / 60.0
    L.check(f(P(U0), P(U1), P(G0), None, *args, dt, g1, 0.0, 1, 0, g.Ny, flags, None), "stage 1")
    L.check(f(P(U1), P(U2a), P(G1a), P(G0), *args, dt, g2, z2, 1, 0, g.Ny, flags, None), "stage 2, classic")
    # previous-state form: the new state goes INTO the buffer that holds U0
    U2b = [S.Field(g, dtype=dtype, data=x.data.clone()) for x in U0]
    L.check(f(P(U1), P(U2b), P(G1b), P(U2b), *args, dt, g2, z2 / g1, 1, 0, g.Ny, flags | L.GM_IS_PREV_STATE, None), "stage 2, previous state")
    torch.cuda.synchronize()
    I = g.interior
    eps = np.finfo(np.float64 if dtype == torch.float64 else np.float32).eps
    for a, b, ga, gb in zip(U2a, U2b, G1a, G1b):
        A_, B_ = a.numpy()[I].astype(np.float64), b.numpy()[I].astype(np.float64)
        assert np.isfinite(B_).all() and np.abs(A_ - B_).max() <= 4 * eps * np.abs(A_).max()
        assert np.array_equal(ga.numpy()[I], gb.numpy()[I])                 # the tendencies themselves do not depend on the operand
    assert f(P(U1), P(U2b), P(G1b), P(U2b), *args, dt, g2, z2 / g1, 1, 0, g.Ny, flags | L.GM_IS_PREV_STATE | L.STRICT, None) == 3     # SWMHD_ENOTSUP
    assert f(P(U1), P(U2b), P(G1b), None, *args, dt, g2, z2 / g1, 1, 0, g.Ny, flags | L.GM_IS_PREV_STATE, None) == 1                  # no operand: SWMHD_EINVAL
