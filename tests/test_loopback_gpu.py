"""The native slab driver (swmhd_amd/csrc/ring.hip: swmhd_ring_step_rk3's deep-halo and per-stage schedules, the exchange left in
flight between calls, two streams per slab) run with DISTINCT neighbours on the one GPU a test box has.

RCCL refuses two ranks on one device, so tests/test_ring_gpu.py can only run a ring of one (north == south == self).  Here the
rings use the in-process loopback transport (swmhd_ring_create_loopback): 2-3 slabs of one periodic domain in one process, each
driven from its own host thread on its own pair of streams, exchanging edge rows by device-to-device copies with RCCL's rendezvous
semantics.  The driver code, its stream/event choreography and the kernels are exactly what the multi-GPU run uses; only the
transport differs.  What the single periodic copy of the reference does (jacobian_formulation/SWMHD_example.jl:16) must come out:
strict builds bit for bit equal to the single-domain model."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DT = 0.002


def _ics(form):
    from test_model_oracle import hf, uf, vf, Af
    if form == "VectorInvariant":
        return dict(u=uf, v=vf, h=hf, A=Af)
    return dict(uh=lambda X, Y: hf(X, Y) * uf(X, Y), vh=lambda X, Y: hf(X, Y) * vf(X, Y), h=hf, A=Af)


def _single(S, form, Nx, Ny, strict, dtype, plan, DT=DT):
    from test_model_oracle import Lx, Ly
    g = S.RectilinearGrid(size=(Nx, Ny), x=(0, Lx), y=(0, Ly))
    m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=strict, dtype=dtype)
    m.set(**_ics(form))
    for n in plan:
        m.time_step(DT) if n == 1 else m.time_steps(n, DT)
    m.synchronize()
    return np.stack([f.numpy()[g.interior] for f in m.fields])


def _slabs(S, form, Nx, Ny_local, world, strict, dtype, plan, deep=True, timeout=60.0, DT=DT, **kw):
    """every slab in its own thread + stream; returns the global interior assembled from the slabs, and the slabs' halo rows"""
    from test_model_oracle import Lx, Ly
    rings = S.loopback_rings(world, timeout)
    out, errs = [None] * world, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                dec = S.SlabDecomposition(Ny_local * world, world, r)
                halo = dec.ring_halo() if deep else (3, 3)
                g = dec.local_grid(S.RectilinearGrid, Nx, x=(0, Lx), y=(0, Ly), halo=halo)
                m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=strict, dtype=dtype, decomp=dec, ring=rings[r], **kw)
                m.set(**_ics(form))
                for n in plan:
                    m.time_step(DT) if n == 1 else m.time_steps(n, DT)
                m.synchronize()
                out[r] = np.stack([f.numpy() for f in m.fields])
                m.close()
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    Hy = 9 if (deep and Ny_local >= 32) else 3
    glob = np.concatenate([p[:, Hy:Hy + Ny_local, 3:3 + Nx] for p in out], axis=1)
    return glob, out, Hy


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
@pytest.mark.parametrize("Ny_local", [32, 33, 45])
def test_deep_halo_driver_with_distinct_neighbours_is_the_single_domain_bitwise(swmhd, world, form, Ny_local):
    plan = (1, 1, 3, 1)       # single steps (the exchange stays in flight between C calls), several steps per call
    want = _single(swmhd, form, 96, Ny_local * world, True, torch.float64, plan)
    got, parents, Hy = _slabs(swmhd, form, 96, Ny_local, world, True, torch.float64, plan)
    assert Hy == 9
    assert np.isfinite(got).all() and np.array_equal(got, want), np.abs(got - want).max()
    # the y halos every slab ends with are its neighbours' edge rows
    for r, p in enumerate(parents):
        south, north = parents[(r - 1) % world], parents[(r + 1) % world]
        assert np.array_equal(p[:, :9, 3:-3], south[:, Ny_local:Ny_local + 9, 3:-3])
        assert np.array_equal(p[:, Ny_local + 9:, 3:-3], north[:, 9:18, 3:-3])


@pytest.mark.parametrize("world,Ny_local", [(2, 8), (3, 7), (2, 20)])
@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
def test_per_stage_driver_with_distinct_neighbours_is_the_single_domain_bitwise(swmhd, world, Ny_local, form):
    plan = (1, 2, 1)
    want = _single(swmhd, form, 80, Ny_local * world, True, torch.float64, plan)
    got, _, Hy = _slabs(swmhd, form, 80, Ny_local, world, True, torch.float64, plan, deep=False)
    assert Hy == 3 and np.array_equal(got, want), np.abs(got - want).max()


def test_per_stage_driver_with_x_halos_in_memory(swmhd):
    """fuse_halo=False: x halos are filled by the halo kernel between the stages (the other branch of the per-stage schedule)"""
    plan = (1, 2)
    want = _single(swmhd, "VectorInvariant", 64, 36, True, torch.float64, plan)
    got, _, _ = _slabs(swmhd, "VectorInvariant", 64, 12, 3, True, torch.float64, plan, deep=False, fuse_halo=False)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32])
@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
def test_fp32_strict_bitwise(swmhd, form, dtype):
    plan = (2, 1)
    want = _single(swmhd, form, 96, 99, True, dtype, plan)
    got, _, _ = _slabs(swmhd, form, 96, 33, 3, True, dtype, plan)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("form,dtype", [("VectorInvariant", torch.float64), ("Conservative", torch.float64), ("VectorInvariant", torch.float32)])
def test_fast_kernels_50_steps_under_real_concurrency_deterministic(swmhd, form, dtype):
    """Fast build on a wide slab (Nx >= 1024: interior rows AND boundary zones on the row-marching kernels, packed fp32 included), 50
    steps, three slabs concurrently on their six streams: within the fast tolerance of the single-domain run (redundantly computed
    boundary rows may come from a different kernel variant than the owner's: last bits, see swmhd.h) and bit-identical run to run."""
    plan = (1, 24, 25)
    Nx, Ny_local, world = 1024, 45, 3
    dt = 2e-4          # dx = 2 pi / 1024: the suite's 2e-3 would be a gravity-wave Courant number of 1.3
    want = _single(swmhd, form, Nx, Ny_local * world, False, dtype, plan, DT=dt)
    a, _, _ = _slabs(swmhd, form, Nx, Ny_local, world, False, dtype, plan, DT=dt)
    b, _, _ = _slabs(swmhd, form, Nx, Ny_local, world, False, dtype, plan, DT=dt)
    assert np.array_equal(a, b), "two runs of the same slabs differ: a race between the streams"
    tol = 1e-11 if dtype == torch.float64 else 2e-4
    scale = np.abs(want).max(axis=(1, 2), keepdims=True)
    assert np.isfinite(a).all() and (np.abs(a - want) / scale).max() <= tol, (np.abs(a - want) / scale).max(axis=(1, 2))


def test_a_neighbour_that_never_arrives_is_an_error_not_a_hang(swmhd):
    """rings driven from ONE thread: rank 0's first exchange finds no neighbour -- SWMHD_ECOMM after the hub's timeout, with a message"""
    from test_model_oracle import Lx, Ly
    S = swmhd
    rings = S.loopback_rings(2, 1.5)
    dec = S.SlabDecomposition(64, 2, 0)
    g = dec.local_grid(S.RectilinearGrid, 64, x=(0, Lx), y=(0, Ly), halo=dec.ring_halo())
    m = S.ShallowWaterModel(g, 9.81, 1.0, decomp=dec, ring=rings[0])
    with pytest.raises(S._lib.SwmhdError, match="neighbour did not reach"):
        m.set(**_ics("VectorInvariant"))
    m.close()
    S._lib.lib().swmhd_ring_destroy(rings[1])
