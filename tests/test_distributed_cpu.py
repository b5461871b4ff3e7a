"""CPU, world_size 2, gloo: the y-slab decomposition + ring halo exchange reproduce the single-domain result.
The compute on each slab is done by the ORACLE here (the product's compute is GPU-only); what is under test is the host
logic that the multi-GPU path uses unchanged: SlabDecomposition, local grids, exchange_y_halos."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, form, lor, nsteps, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from swmhd_amd import SlabDecomposition, exchange_y_halos, RectilinearGrid
        from test_model_oracle import staggered_fields, G, F
        N, H = 12 * world, 3
        q, _, dx, dy = staggered_fields(N, form)
        q = [O.fill_halo_periodic(a, N, N, H, H) for a in q]
        dec = SlabDecomposition(N, world, rank)
        g = dec.local_grid(RectilinearGrid, N, x=(0, N * dx), y=(0, N * dy))
        assert g.Ny == N // world and g.dy == dy
        # my slab = my rows of the global parents (+ halos, to be overwritten by the exchange)
        sl = slice(dec.j_offset, dec.j_offset + g.Ny + 2 * H)
        loc = [np.ascontiguousarray(a[sl]) for a in q]
        for a in loc:   # poison y halos: they must come from the neighbours
            a[:H] = np.nan; a[g.Ny + H:] = np.nan
        ts = [torch.from_numpy(a) for a in loc]
        exchange_y_halos(ts, g.Ny, H, dec)
        for a, full in zip(loc, q):
            assert np.array_equal(a, full[sl]), "halo exchange did not reproduce the periodic global halos"
        dt = 0.002
        gam, zet = (8 / 15, 5 / 12, 3 / 4), (0.0, -17 / 60, -5 / 12)
        Gm = None
        for _ in range(nsteps):
            for s in range(3):
                Gn = O.tendencies(*loc, N, g.Ny, H, H, dx, dy, form, lor, G, F)
                for a, gn, k in zip(loc, Gn, range(4)):
                    I = (slice(H, H + g.Ny), slice(H, H + N))
                    a[I] += dt * gam[s] * gn[I] if s == 0 else dt * (gam[s] * gn[I] + zet[s] * Gm[k][I])
                    a[:, :H] = a[:, N:N + H]; a[:, N + H:] = a[:, H:2 * H]          # local periodic x fill
                Gm = Gn
                exchange_y_halos(ts, g.Ny, H, dec)
        np.save(os.path.join(out, f"rank{rank}.npy"), np.stack([a[H:H + g.Ny] for a in loc]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
def test_slab_run_equals_single_domain(oracle, tmp_path, form, lor, world):
    """world 2: both ring neighbours are the same peer (two sends + two recvs to one rank in one batch);
    world 3: distinct north / south peers."""
    from test_model_oracle import staggered_fields, G, F
    nsteps = 2
    mp.spawn(_worker, args=(world, _free_port(), form, lor, nsteps, str(tmp_path)), nprocs=world, join=True)
    N, H = 12 * world, 3
    q, _, dx, dy = staggered_fields(N, form)
    q = [oracle.fill_halo_periodic(a, N, N, H, H) for a in q]
    for _ in range(nsteps):
        oracle.time_step(*q, N, N, H, H, dx, dy, 0.002, form, lor, G, F)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)], axis=1)
    want = np.stack([a[H:H + N] for a in q])
    assert np.array_equal(got, want), np.abs(got - want).max()


def _worker_deep(rank, world, port, form, lor, nsteps, out):
    """The deep-halo schedule of swmhd_ring_step_rk3 restated with the oracle: ONE exchange of 9 rows per RK3 step; stage k evaluates
    rows [-(6 - 3k), Ny + 6 - 3k) of the slab (its neighbours' edge rows included), all from its own halo."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from swmhd_amd import SlabDecomposition, exchange_y_halos, RectilinearGrid
        from test_model_oracle import staggered_fields, G, F
        N, Hx, Hy = 16 * world, 3, 9
        q, _, dx, dy = staggered_fields(N, form)
        glob = [np.zeros((N + 2 * Hy, N + 2 * Hx)) for _ in q]
        for gl, a in zip(glob, q):
            gl[Hy:Hy + N, Hx:Hx + N] = a[3:3 + N, 3:3 + N]
            gl[:, :Hx] = gl[:, N:N + Hx]; gl[:, N + Hx:] = gl[:, Hx:2 * Hx]
        dec = SlabDecomposition(N, world, rank)
        g = dec.local_grid(RectilinearGrid, N, x=(0, N * dx), y=(0, N * dy), halo=(Hx, Hy))
        Ny = g.Ny
        sl = slice(dec.j_offset, dec.j_offset + Ny + 2 * Hy)
        loc = [np.ascontiguousarray(a[sl]) for a in glob]
        for a in loc:
            a[:Hy] = np.nan; a[Ny + Hy:] = np.nan
        ts = [torch.from_numpy(a) for a in loc]
        exchange_y_halos(ts, Ny, Hy, dec)
        dt = 0.002
        gam, zet = (8 / 15, 5 / 12, 3 / 4), (0.0, -17 / 60, -5 / 12)
        for _ in range(nsteps):
            Gm = None
            for s in range(3):
                e = 6 - 3 * s                                   # rows [-e, Ny + e): a sub-grid of Ny + 2e rows with the stencil's halo of 3
                sub = slice(Hy - e - 3, Hy + Ny + e + 3)
                views = [a[sub] for a in loc]
                Gn = O.tendencies(*[np.ascontiguousarray(v) for v in views], N, Ny + 2 * e, Hx, 3, dx, dy, form, lor, G, F)
                I = (slice(3, 3 + Ny + 2 * e), slice(Hx, Hx + N))
                for k, (v, gn) in enumerate(zip(views, Gn)):
                    if s == 0:
                        v[I] += dt * gam[s] * gn[I]
                    else:
                        v[I] += dt * (gam[s] * gn[I] + zet[s] * Gm[k][3 + 3:3 + 3 + Ny + 2 * e, Hx:Hx + N])   # G- lives on the previous, 3-rows-wider sub-grid
                    v[:, :Hx] = v[:, N:N + Hx]; v[:, N + Hx:] = v[:, Hx:2 * Hx]
                Gm = Gn
            exchange_y_halos(ts, Ny, Hy, dec)                   # once per step, all 9 rows
        np.save(os.path.join(out, f"rank{rank}.npy"), np.stack([a[Hy:Hy + Ny, Hx:Hx + N] for a in loc]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("form,lor", [(1, 1), (0, 2)])
def test_one_exchange_per_step_with_a_nine_row_halo_equals_single_domain(oracle, tmp_path, form, lor, world):
    """The arithmetic of the deep-halo schedule across REAL neighbours (the GPU tests can only run a ring of one rank): slabs that
    evaluate their neighbours' edge rows redundantly and exchange once per step reproduce the single-domain run bit for bit."""
    from test_model_oracle import staggered_fields, G, F
    nsteps = 2
    mp.spawn(_worker_deep, args=(world, _free_port(), form, lor, nsteps, str(tmp_path)), nprocs=world, join=True)
    N, H = 16 * world, 3
    q, _, dx, dy = staggered_fields(N, form)
    q = [oracle.fill_halo_periodic(a, N, N, H, H) for a in q]
    for _ in range(nsteps):
        oracle.time_step(*q, N, N, H, H, dx, dy, 0.002, form, lor, G, F)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)], axis=1)
    want = np.stack([a[H:H + N, H:H + N] for a in q])
    assert np.array_equal(got, want), np.abs(got - want).max()


def test_decomposition_bookkeeping():
    sys.path.insert(0, ROOT)
    from swmhd_amd import SlabDecomposition, RectilinearGrid
    d = SlabDecomposition(4096, 8, 3)
    assert (d.Ny_local, d.j_offset, d.south, d.north) == (512, 1536, 2, 4)
    assert SlabDecomposition(64, 4, 0).south == 3 and SlabDecomposition(64, 4, 3).north == 0
    with pytest.raises(ValueError):
        SlabDecomposition(100, 8, 0)
    g = d.local_grid(RectilinearGrid, 4096, x=(-1, 1), y=(-10, 10))
    assert g.Ny == 512 and g.dy == 20 / 4096 and abs(g.yc[3] - (-10 + (1536 + 0.5) * g.dy)) < 1e-12


def _worker_agree(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from swmhd_amd.distributed import agree_rc
        got = [agree_rc(0), agree_rc(4 if rank == 1 else 0), agree_rc(rank)]
        with open(os.path.join(out, f"agree{rank}.txt"), "w") as f:
            f.write(" ".join(map(str, got)))
    finally:
        dist.destroy_process_group()


def test_return_codes_are_agreed_across_ranks(tmp_path):
    """swmhd_ring_create on N ranks: a failure on ONE rank must be seen by all of them (they destroy their communicators and raise
    together instead of blocking in the first exchange; swmhd_amd/model.py::_create_ring) -- the agreement primitive, over gloo."""
    world = 3
    mp.spawn(_worker_agree, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"agree{r}.txt").read() == "0 4 2"
    from swmhd_amd.distributed import agree_rc
    assert agree_rc(7) == 7          # no process group: the rank's own code
