"""GPU test of the NATIVE multi-GPU path (swmhd_ring_*: RCCL communicator + comm stream + overlapped step driver) on the one
card a test box has: a ring of ONE rank, where every ncclSend goes to the rank itself.  That exercises everything the 8-GPU
run uses -- id broadcast, ncclCommInitRank, the grouped zero-copy edge-row exchange, interior/strip split across the two
streams, the exchange left in flight between calls -- except the xGMI links themselves.  (RCCL refuses two ranks on one
device, so the 2-rank rehearsals in test_distributed_gpu.py go through gloo and the torch p2p path instead.)

A slab run through the ring must reproduce the plain single-GPU periodic model: bit for bit with the strict kernels,
within the fast-kernel tolerance otherwise (interior rows and 3-row strips take different kernel variants)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DT = 0.002


@pytest.fixture(scope="module")
def rccl_world_of_one():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    torch.cuda.set_device(0)
    import socket
    with socket.socket() as sk:      # a free port, not a fixed one (another job on the box may hold it)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _model(S, form, N, strict, ring, dtype=torch.float64, **kw):
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    dec = S.SlabDecomposition(N, 1, 0, force_ring=ring)
    g = dec.local_grid(S.RectilinearGrid, N, x=(0, Lx), y=(0, Ly))
    m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=strict, decomp=dec, dtype=dtype, **kw)
    if form == "VectorInvariant":
        m.set(u=uf, v=vf, h=hf, A=Af)
    else:
        m.set(uh=lambda X, Y: hf(X, Y) * uf(X, Y), vh=lambda X, Y: hf(X, Y) * vf(X, Y), h=hf, A=Af)
    return m


def _state(m):
    m.synchronize()
    return np.stack([f.numpy() for f in m.fields])      # parents: halos included


@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_ring_of_one_is_the_periodic_model_bitwise(rccl_world_of_one, form, dtype):
    import swmhd_amd as S
    N = 96
    ref = _model(S, form, N, True, ring=False, dtype=dtype)
    rng = _model(S, form, N, True, ring=True, dtype=dtype)
    assert rng._ring is not None, "native ring was not created under the nccl backend"
    a0, b0 = _state(ref), _state(rng)
    assert np.array_equal(a0, b0), "halo rows filled through RCCL differ from the local periodic copy"
    for _ in range(3):
        ref.time_step(DT); rng.time_step(DT)           # one C call per step: the exchange stays in flight between calls
    assert np.array_equal(_state(ref), _state(rng))
    ref.time_steps(4, DT); rng.time_steps(4, DT)       # one C call for 4 steps
    a, b = _state(ref), _state(rng)
    assert np.isfinite(a).all() and np.array_equal(a, b)
    da, db = ref.diagnostics(), rng.diagnostics()
    assert da == db


def _interior(m):
    m.synchronize()
    return np.stack([f.numpy()[m.grid.interior] for f in m.fields])


@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_deep_halo_schedule_is_the_periodic_model_bitwise(rccl_world_of_one, form, dtype):
    """Slab grids with Hy = 9 (SlabDecomposition.ring_halo) take swmhd_ring_step_rk3's deep-halo schedule: one exchange per step,
    boundary rows of stages 1-2 evaluated redundantly inside the halo.  Strict build: every row is the same arithmetic whoever
    computes it, so the interiors equal the plain periodic model's bit for bit -- single steps (the exchange stays in flight
    between calls), several steps per call, and after a join + diagnostics in between."""
    import swmhd_amd as S
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    N = 96
    ref = _model(S, form, N, True, ring=False, dtype=dtype)
    dec = S.SlabDecomposition(N, 1, 0, force_ring=True)
    assert dec.ring_halo() == (3, 9)
    g = dec.local_grid(S.RectilinearGrid, N, x=(0, Lx), y=(0, Ly), halo=dec.ring_halo())
    rng = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=True, decomp=dec, dtype=dtype)
    if form == "VectorInvariant":
        rng.set(u=uf, v=vf, h=hf, A=Af)
    else:
        rng.set(uh=lambda X, Y: hf(X, Y) * uf(X, Y), vh=lambda X, Y: hf(X, Y) * vf(X, Y), h=hf, A=Af)
    assert rng._ring is not None and g.Hy == 9
    assert np.array_equal(_interior(ref), _interior(rng))
    for _ in range(3):
        ref.time_step(DT); rng.time_step(DT)
    assert np.array_equal(_interior(ref), _interior(rng))
    assert ref.diagnostics() == rng.diagnostics()
    ref.time_steps(4, DT); rng.time_steps(4, DT)
    a, b = _interior(ref), _interior(rng)
    assert np.isfinite(a).all() and np.array_equal(a, b)
    ref.time_step(DT); rng.time_step(DT)
    assert np.array_equal(_interior(ref), _interior(rng))
    # the y halos of the returned state are the periodic images (the exchange of the final state has been joined by _interior)
    for f in rng.fields:
        p = f.numpy()
        assert np.array_equal(p[:9, 3:-3], p[N:N + 9, 3:-3]) and np.array_equal(p[N + 9:, 3:-3], p[9:18, 3:-3])


@pytest.mark.parametrize("form,dtype", [("VectorInvariant", torch.float64), ("Conservative", torch.float64), ("VectorInvariant", torch.float32)])
def test_deep_halo_fast_kernels_thin_and_tall_slabs(rccl_world_of_one, form, dtype):
    """Fast build, deep-halo schedule: interior rows AND (Nx >= 1024) both boundary zones per launch on the row-marching kernels --
    vector-invariant, conservative and the packed fp32 kernel with their second row range; a thin slab (4096 x 64: the boundary zones
    almost meet) and a taller one against the plain periodic model within the fast tolerance."""
    import swmhd_amd as S
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    for Nx, Ny in ((4096, 64), (2048, 1024)):
        out = []
        for ring in (False, True):
            dec = S.SlabDecomposition(Ny, 1, 0, force_ring=ring)
            g = dec.local_grid(S.RectilinearGrid, Nx, x=(0, Lx), y=(0, Ly), halo=dec.ring_halo())
            m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, decomp=dec, dtype=dtype)
            if form == "VectorInvariant":
                m.set(u=uf, v=vf, h=hf, A=Af)
            else:
                m.set(uh=lambda X, Y: hf(X, Y) * uf(X, Y), vh=lambda X, Y: hf(X, Y) * vf(X, Y), h=hf, A=Af)
            m.time_steps(2, 1e-4); m.time_step(1e-4)
            out.append(_interior(m))
            m.close()
        a, b = out
        assert np.isfinite(a).all()
        for k in range(4):
            assert np.abs(a[k] - b[k]).max() <= tol * max(np.abs(a[k]).max(), 1.0), (Nx, Ny, k)


@pytest.mark.parametrize("Nx,Ny", [(100, 32), (130, 33), (1100, 45), (1101, 45)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_deep_halo_schedule_awkward_sizes(rccl_world_of_one, Nx, Ny, dtype):
    """Smallest slab the deep-halo schedule takes (32 rows: the last stage's interior is 8 rows), odd heights and widths, widths
    on either side of the 1024 columns from which the boundary zones go to the row-marching kernel (odd width: the unpacked fp32
    kernel); strict build bitwise against the periodic model, fast build within tolerance."""
    import swmhd_amd as S
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    for strict in (True, False):
        out = []
        for ring in (False, True):
            dec = S.SlabDecomposition(Ny, 1, 0, force_ring=ring)
            g = dec.local_grid(S.RectilinearGrid, Nx, x=(0, Lx), y=(0, Ly), halo=dec.ring_halo())
            assert (g.Hy == 9) == ring
            m = S.ShallowWaterModel(g, 9.81, 1.0, formulation="VectorInvariant", strict=strict, decomp=dec, dtype=dtype)
            m.set(u=uf, v=vf, h=hf, A=Af)
            m.time_steps(2, 1e-4); m.time_step(1e-4)
            out.append(_interior(m))
            m.close()
        a, b = out
        assert np.isfinite(a).all()
        if strict:
            assert np.array_equal(a, b)
        else:
            tol = 1e-12 if dtype == torch.float64 else 2e-5
            for k in range(4):
                assert np.abs(a[k] - b[k]).max() <= tol * max(np.abs(a[k]).max(), 1.0), k


def test_deep_halo_schedule_under_real_concurrency(rccl_world_of_one):
    """Hazard check of the deep-halo schedule with launches long enough to overlap on the chip (the 96^2 case above is over before
    the other stream starts): (a) strict kernels, 2048 x 40 and 2048 x 64 slabs, 60 steps in calls of 1 and 20 -- bitwise equal to the
    periodic model; (b) fast kernels, 4096 x 512, two independent runs of 60 steps -- bitwise equal to each other (a race between
    the interior launches, the boundary launches and the exchange would show as a difference)."""
    import swmhd_amd as S
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly

    def build(Nx, Ny, ring, strict):
        dec = S.SlabDecomposition(Ny, 1, 0, force_ring=ring)
        g = dec.local_grid(S.RectilinearGrid, Nx, x=(0, Lx), y=(0, Ly), halo=dec.ring_halo())
        m = S.ShallowWaterModel(g, 9.81, 1.0, formulation="VectorInvariant", strict=strict, decomp=dec)
        m.set(u=uf, v=vf, h=hf, A=Af)
        return m

    for Ny in (40, 64):
        ref, rng = build(2048, Ny, False, True), build(2048, Ny, True, True)
        for m in (ref, rng):
            for _ in range(20):
                m.time_step(2e-5)
            m.time_steps(20, 2e-5); m.time_steps(20, 2e-5)
        a, b = _interior(ref), _interior(rng)
        assert np.isfinite(a).all() and np.array_equal(a, b), Ny
        rng.close()
    runs = []
    for _ in range(2):
        m = build(4096, 512, True, False)
        m.time_steps(30, 2e-5)
        for _ in range(30):
            m.time_step(2e-5)
        runs.append(_interior(m))
        m.close()
    assert np.isfinite(runs[0]).all() and np.array_equal(runs[0], runs[1])


def test_ring_fast_kernels_large_slab(rccl_world_of_one):
    """2048 x 1024 slab: interior rows take the row-marching kernel, the strips the tile kernel; fast build tolerance."""
    import swmhd_amd as S
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    Nx, Ny = 2048, 1024
    ms = []
    for ring in (False, True):
        dec = S.SlabDecomposition(Ny, 1, 0, force_ring=ring)
        g = dec.local_grid(S.RectilinearGrid, Nx, x=(0, Lx), y=(0, Ly))
        m = S.ShallowWaterModel(g, 9.81, 1.0, formulation="VectorInvariant", decomp=dec)
        m.set(u=uf, v=vf, h=hf, A=Af)
        m.time_steps(3, 1e-4)
        ms.append(_state(m))
    a, b = ms
    assert np.isfinite(a).all()
    for k in range(4):
        assert np.abs(a[k] - b[k]).max() <= 1e-12 * max(np.abs(a[k]).max(), 1.0)


def test_ring_matches_torch_p2p_path(rccl_world_of_one):
    """The native driver and the Python-driven stages over torch.distributed p2p (same RCCL underneath) agree bitwise."""
    import swmhd_amd as S
    N = 128
    nat = _model(S, "VectorInvariant", N, True, ring=True)
    py = _model(S, "VectorInvariant", N, True, ring=True, native_ring=False)
    assert nat._ring is not None and py._ring is None
    for _ in range(3):
        nat.time_step(DT); py.time_step(DT)
    assert np.array_equal(_state(nat), _state(py))


def test_ring_exchange_entry_point(rccl_world_of_one):
    """swmhd_ring_exchange_y_* alone == the y part of swmhd_fill_halo_periodic on one rank; rejects bad arguments."""
    import ctypes
    import swmhd_amd as S
    from swmhd_amd import _lib
    L = _lib.lib()
    m = _model(S, "VectorInvariant", 64, True, ring=True)
    g = m.grid
    rngen = torch.Generator(device="cpu").manual_seed(7)
    fields = [torch.randn(g.Ny + 2 * g.Hy, g.Nx + 2 * g.Hx, generator=rngen, dtype=torch.float64).cuda() for _ in range(3)]
    want = [f.clone() for f in fields]
    for w in want:
        w[:g.Hy] = w[g.Ny:g.Ny + g.Hy]
        w[g.Ny + g.Hy:] = w[g.Hy:2 * g.Hy]
    ptrs = _lib.ptr_array([f.data_ptr() for f in fields])
    rc = L.swmhd_ring_exchange_y_f64(m._ring, ptrs, 3, g.Nx, g.Ny, g.Hx, g.Hy, g.Nx + 2 * g.Hx, None)
    assert rc == 0
    torch.cuda.synchronize()
    for f, w in zip(fields, want):
        assert torch.equal(f, w)
    assert L.swmhd_ring_exchange_y_f64(m._ring, ptrs, 0, g.Nx, g.Ny, g.Hx, g.Hy, g.Nx + 2 * g.Hx, None) == 1
    assert L.swmhd_ring_exchange_y_f64(None, ptrs, 3, g.Nx, g.Ny, g.Hx, g.Hy, g.Nx + 2 * g.Hx, None) == 1
    assert L.swmhd_ring_exchange_y_f64(m._ring, ptrs, 3, g.Nx, g.Ny, g.Hx, g.Hy, g.Nx, None) == 1   # pitch < padded width
    # the step driver refuses wrap flags (y images belong to the neighbours) and slabs without interior rows
    q = _lib.ptr_array([f.ptr for f in m.fields]); qa = _lib.ptr_array([m._alt[n].ptr for n in m.names])
    Ga = _lib.ptr_array([f.ptr for f in m.Gn]); Gb = _lib.ptr_array([f.ptr for f in m.Gm])
    args = (g.Nx, g.Ny, g.Hx, g.Hy, m.fields[0].stride_y, g.dx, g.dy, 9.81, 1.0, 1, 1, 1e-3, 1)
    assert L.swmhd_ring_step_rk3_f64(m._ring, q, qa, Ga, Gb, *args, _lib.WRAP_Y, None, None) == 1
    assert L.swmhd_ring_step_rk3_f64(m._ring, q, qa, Ga, Gb, g.Nx, 6, *args[2:], 0, None, None) == 1
