"""GPU test of the multi-rank path on ONE card: 2 ranks share cuda:0, backend gloo (halo rows staged through the host --
the rehearsal mode of swmhd_amd.distributed; the 8-GPU run uses nccl == RCCL with the same code path otherwise).
The slab-decomposed model, with interior/boundary overlap on two streams, must reproduce the single-domain run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, NSTEPS, DT = 96, 3, 0.002


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _ics():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_model_oracle import hf, uf, vf, Af, Lx, Ly
    return dict(h=hf, u=uf, v=vf, A=Af), Lx, Ly


def _build(S, form, dec, strict, overlap=True, group=None, deep=False):
    ics, Lx, Ly = _ics()
    g = dec.local_grid(S.RectilinearGrid, N, x=(0, Lx), y=(0, Ly), halo=dec.ring_halo() if deep else (3, 3))
    m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=strict, decomp=dec, group=group, overlap=overlap)
    if form == "VectorInvariant":
        m.set(u=ics["u"], v=ics["v"], h=ics["h"], A=ics["A"])
    else:
        m.set(uh=lambda X, Y: ics["h"](X, Y) * ics["u"](X, Y), vh=lambda X, Y: ics["h"](X, Y) * ics["v"](X, Y), h=ics["h"], A=ics["A"])
    return m


def _worker(rank, world, port, form, strict, out, deep=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import swmhd_amd as S
        dec = S.SlabDecomposition(N, world, rank)
        m = _build(S, form, dec, strict, deep=deep)
        assert m.grid.Hy == (9 if deep else 3)
        for _ in range(NSTEPS):
            m.time_step(DT)
        m.synchronize()
        d = m.diagnostics()
        I = m.grid.interior
        np.save(os.path.join(out, f"rank{rank}.npy"), np.stack([f.numpy()[I] for f in m.fields]))
        if rank == 0:
            np.save(os.path.join(out, "diag.npy"), np.array([d["total_energy"], d["max_abs_A"], d["min_h"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("form,strict,deep", [("VectorInvariant", True, False), ("VectorInvariant", False, False), ("Conservative", True, False),
                                              ("VectorInvariant", True, True)])
def test_two_ranks_on_one_gpu_match_single_domain(swmhd, tmp_path, form, strict, deep):
    """deep: slab grids with the 9-row y halo bench.py gives them (SlabDecomposition.ring_halo) on the torch.distributed p2p path, which
    keeps the per-stage schedule and simply exchanges all 9 rows."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), form, strict, str(tmp_path), deep), nprocs=world, join=True)
    m = _build(swmhd, form, swmhd.SlabDecomposition(N, 1, 0), strict)
    for _ in range(NSTEPS):
        m.time_step(DT)
    m.synchronize()
    want = np.stack([f.numpy()[m.grid.interior] for f in m.fields])
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)], axis=1)
    if strict:
        assert np.array_equal(got, want), np.abs(got - want).max()
    else:   # marching kernel: segment boundaries differ between the slab and the full domain -> rounding-level differences only
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
    d = m.diagnostics()
    dd = np.load(tmp_path / "diag.npy")
    assert abs(dd[0] - d["total_energy"]) <= 1e-12 * abs(d["total_energy"]) and dd[1] == d["max_abs_A"] and dd[2] == d["min_h"]


def test_bench_line_for_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2 --backend gloo` on the one-GPU box: bench.py launches its two workers itself, both on cuda:0, halo rows
    staged through the host (rehearsal mode) -- everything of the N > 1 bench path except RCCL runs: strong-scaled 4096^2 headline,
    the weak-scaled `companion` run, max over ranks, one JSON line from rank 0."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "4096^2" in d["metric"] and d["config"]["finite"]
    assert "4096x2048 cells per GPU" in d["config"]["workload"] and d["value"] > 0
    c = d["companion"]
    assert c["scaling"] == "weak" and c["finite"] and "4096x8192" in c["workload"] and c["value"] > 0
    assert d["roofline"]["cells_per_launch"] < 4096 * 2048       # rank 0's interior launches
