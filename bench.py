#!/usr/bin/env python3
"""bench.py -- headline benchmark of the shallow-water-MHD tendency engine on MI355X.

Contract (see the task statement):  python bench.py --gpus N --steps K --warmup W  prints ONE JSON line on rank 0.
Metric: BASELINE.json's "Mcell-steps/sec (fp64) on 4096^2 periodic grid".  Workload: BASELINE config 3
(4096x4096, Jacobian formulation, Bickley-jet-style h/u, fp64), inputs resident in HBM before the timed region.

Round-1 state of the "step": one evaluation of the Jacobian-form Lorentz-force operator over the whole grid
(the reference's hot path, sw_mhd_jacobian_functions.jl:1-26) -- `config.step` says so.  Weak scaling for N > 1:
every rank owns a 4096 x 4096 y-slab of a 4096 x (4096 N) periodic domain.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
OP_BYTES_PER_CELL = 32         # read A,h + write Fx,Fy, fp64


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--n", type=int, default=4096, help="grid edge (default: BASELINE config 3)")
    p.add_argument("--form", default="jacobian", choices=["jacobian", "divergence"])
    p.add_argument("--strict", action="store_true", help="time the reference-order (bitwise) kernels instead of the fast ones")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    return p.parse_args()


def cpu_baseline(args, cfg, g_full):
    """Oracle (C restatement of the reference's per-cell functions, same cost structure) on the host cores, on a
    bounded sample of the same workload: the first rows of the 4096^2 grid, repeated until the time budget is used."""
    from oracle import oracle as O
    import swmhd_amd as S
    cores = os.cpu_count() or 1
    Ny_s = 256
    g = S.RectilinearGrid(size=(args.n, Ny_s), x=cfg["domain"]["x"], y=cfg["domain"]["y"], halo=(3, 3),
                          j_offset=args.n // 2 - Ny_s // 2, Ny_global=args.n)
    X, Y = g.nodes(("Center", "Center"))
    A = np.ascontiguousarray(cfg["A"](X, Y)); h = np.ascontiguousarray(cfg["h"](X, Y) + 0 * X)
    fn = O.lorentz_jacobian if args.form == "jacobian" else O.lorentz_divergence
    fn(A, h, g.Nx, g.Ny, 3, 3, g.dx, g.dy, nthreads=cores)  # warm
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        fn(A, h, g.Nx, g.Ny, 3, 3, g.dx, g.dy, nthreads=cores); reps += 1
    dt = time.perf_counter() - t0
    return {"value": g.Nx * g.Ny * reps / dt / 1e6, "unit": "Mcell-steps/s", "cores": cores, "kind": "port",
            "sample": f"{args.n}x{Ny_s} centre rows of the {args.n}^2 workload, {reps} reps, C oracle with the "
                      f"reference's unshared per-cell composition, OpenMP over {cores} threads"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import swmhd_amd as S
    from swmhd_amd import configs
    cfg = configs.config3_bickley()
    N = args.n
    # weak scaling: rank r owns rows [r*N, (r+1)*N) of an N x (N*world) domain (y extent scales with world)
    y0, y1 = cfg["domain"]["y"]
    g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=(y0, y0 + (y1 - y0) * world), halo=(3, 3),
                          j_offset=rank * N, Ny_global=N * world)
    A, h = S.Field(g), S.Field(g)
    A.set(cfg["A"]); h.set(lambda X, Y: cfg["h"](X, Y) + 0 * X)
    A.fill_halo_regions(); h.fill_halo_regions()     # x wrap (+ y wrap; at N>1 the slab halos come from set())
    out = (S.Field(g, (S.Face, S.Center)), S.Field(g, (S.Center, S.Face)))
    fn = S.lorentz_force_func if args.form == "jacobian" else S.div_lorentz
    fields = {"A": A, "h": h}

    def step():
        fn(g, fields, out=out, strict=args.strict)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = t.item()
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # HIP events on the launch stream, live, over the timed region

    if rank == 0:
        cells = N * N
        value = cells * world * args.steps / wall / 1e6
        achieved = OP_BYTES_PER_CELL * cells / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "Mcell-steps/sec (fp64) on 4096^2 periodic grid", "value": value, "unit": "Mcell-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N}x{N} periodic, {args.form} formulation, Bickley-jet h/u + current-sheet A (BASELINE config 3)",
                       "step": f"one whole-grid evaluation of the {args.form}-form Lorentz force ({'strict' if args.strict else 'fast'} kernel)",
                       "decomposition": f"y-slabs x{world}, {N}x{N} cells per GPU"},
            "roofline": {"bound": "hbm", "kernel": f"k_lorentz_{args.form}", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": OP_BYTES_PER_CELL * cells, "avg_launch_ms": kern_ms},
        }
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args, cfg, g)
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
