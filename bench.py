#!/usr/bin/env python3
"""bench.py -- headline benchmark of the shallow-water-MHD tendency engine on MI355X.

Contract:  python bench.py --gpus N --steps K --warmup W  prints ONE JSON line on rank 0.
Metric (BASELINE.json): Mcell-steps/sec (fp64) on a 4096^2 periodic grid.  One step = one full RK3 time step of the
model the reference builds (SWMHD_example.jl:21-42): 3 x {fused tendency evaluation incl. the Jacobian-form Lorentz
force, RK3 substep of 4 fields, halo fill}.  Workload: BASELINE config 3 (4096x4096, Jacobian formulation = vector-
invariant u,v,h + tracer A, Bickley-jet-style h/u, fp64), synthetic initial condition resident in HBM before timing.
Weak scaling for N > 1: every rank owns a 4096 x 4096 y-slab of a 4096 x (4096 N) periodic domain; halo rows move by
RCCL send/recv between ring neighbours, overlapped with the interior rows on a second stream.
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
TEND_BYTES_PER_CELL = 64       # read u,v,h,A + write 4 tendencies, fp64
STEP_BYTES_PER_CELL = 544      # 3*64 + 96 + 128 + 128


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--size", dest="n", type=int, default=4096, help="grid edge per GPU (default: BASELINE config 3)")
    p.add_argument("--formulation", default="VectorInvariant", choices=["VectorInvariant", "Conservative"])
    p.add_argument("--strict", action="store_true", help="time the oracle-order (bitwise) kernels instead of the fast ones")
    p.add_argument("--dt", type=float, default=1e-4)
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    p.add_argument("--backend", default="nccl")
    p.add_argument("--torch-ring", action="store_true", help="multi-GPU: exchange through torch.distributed p2p instead of the native ring")
    p.add_argument("--force-ring", action="store_true",
                   help="N=1 rehearsal of the multi-GPU step: y halos through the RCCL ring exchange (sends to self) + overlap")
    return p.parse_args()


def build_model(S, args, rank, world, n, ny_local=None):
    from swmhd_amd import configs
    cfg = configs.config3_bickley() if args.formulation == "VectorInvariant" else configs.config4_two_gaussians()
    y0, y1 = cfg["domain"]["y"]
    dec = S.SlabDecomposition(n * world, world, rank, force_ring=args.force_ring)
    g = dec.local_grid(S.RectilinearGrid, n, x=cfg["domain"]["x"], y=(y0, y0 + (y1 - y0) * world))
    return cfg, dec, g


def valu_floor(Nx, rows, kern_ms, args):
    """Secondary bound (SURVEY 8(d) asks for it): the fp64 vector-ALU floor of the vector-invariant marching kernel.  585 VALU
    instructions per wave per row (static count of the steady loop in the ISA, cross-checked with SQ_INSTS_VALU,
    profiles/r01/tendency_pmc_sq*.json) x wave-rows / 1024 SIMDs x 2.05 ns per wave-instruction (what one SIMD sustains on fp64,
    tools/valu_probe.hip)."""
    if args.formulation != "VectorInvariant" or args.strict:
        return {}
    nstrips = -(-Nx // 250)
    nseg = max(1, (768 // nstrips))
    wave_rows = nstrips * 4 * (rows + 6 * nseg)
    floor_ms = 585 * wave_rows / 1024 * 2.05e-6
    return {"valu_floor_ms": floor_ms, "frac_of_valu_floor": floor_ms / kern_ms}


def cpu_baseline(args, cfg):
    """The oracle's RK3 step (C restatement: the reference's per-cell Lorentz functions with their unshared composition +
    the restated Oceananigans RHS) on the host cores, on a bounded sample of the same workload: a 512 x 256 periodic
    block cut from the centre of the configuration, stepped until the time budget is used."""
    from oracle import oracle as O
    import swmhd_amd as S
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)   # the box's CPU share for one GPU
    Nx, Ny = 512, 256
    x0, x1 = cfg["domain"]["x"]; y0, y1 = cfg["domain"]["y"]
    dx, dy = (x1 - x0) / args.n, (y1 - y0) / args.n
    g = S.RectilinearGrid(size=(Nx, Ny), x=(-Nx * dx / 2, Nx * dx / 2), y=(-Ny * dy / 2, Ny * dy / 2))
    form = 1 if args.formulation == "VectorInvariant" else 0
    names = [("u", ("Face", "Center")), ("v", ("Center", "Face")), ("h", ("Center", "Center")), ("A", ("Center", "Center"))]
    q = []
    for nm, loc in names:
        X, Y = g.nodes(loc)
        q.append(O.fill_halo_periodic(np.ascontiguousarray(cfg[nm](X, Y) + 0 * X), Nx, Ny, 3, 3))
    work = O.time_step(*q, Nx, Ny, 3, 3, dx, dy, args.dt, form, 2 - form, nthreads=cores)   # warm
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        O.time_step(*q, Nx, Ny, 3, 3, dx, dy, args.dt, form, 2 - form, nthreads=cores, work=work); reps += 1
    dt = time.perf_counter() - t0
    return {"value": Nx * Ny * reps / dt / 1e6, "unit": "Mcell-steps/s", "cores": cores, "kind": "port",
            "sample": f"{Nx}x{Ny} periodic block at the workload's dx,dy and fields, {reps} RK3 steps; C oracle (reference's "
                      f"unshared per-cell Lorentz composition + restated Oceananigans RHS), OpenMP over {cores} threads; "
                      "Julia/Oceananigans are not available on the box"}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: everything else any library prints there (RCCL announces its version on stdout
    # when a communicator is created) is sent to stderr.
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dist = None
    if world > 1 or args.force_ring:
        import torch.distributed as dist
        kw = {"device_id": torch.device("cuda", torch.cuda.current_device())} if args.backend == "nccl" else {}
        if "RANK" not in os.environ:   # --force-ring started without a launcher
            kw.update(init_method=f"tcp://127.0.0.1:{29500 + os.getpid() % 2000}", rank=0, world_size=1)
        dist.init_process_group(args.backend, **kw)

    import swmhd_amd as S
    N = args.n
    cfg, dec, g = build_model(S, args, rank, world, N)
    m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=args.formulation, strict=args.strict, decomp=dec, native_ring=not args.torch_ring)
    n1, n2 = m.names[:2]
    if args.formulation == "VectorInvariant":
        m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
    else:
        m.set(**{n1: cfg["u"], n2: cfg["v"], "h": cfg["h"], "A": cfg["A"]})

    # The device needs ~30 ms of sustained load before its clocks settle (a step measured right after start-up is 6-10 % slower
    # than the same step a hundred steps later): spin the step 40 times before the W warm-up steps the contract counts.
    for _ in range(40 + args.warmup):
        m.time_step(args.dt)
    m.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    m.tendency_events = []
    if m._ring is not None:
        m.ring_time_launches(3 * args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.time_step(args.dt)
    m.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = t.item()
    finite = all(torch.isfinite(f.data).all().item() for f in m.fields)

    if rank == 0:
        cells = N * N
        value = cells * world * args.steps / wall / 1e6
        line = {
            "metric": "Mcell-steps/sec (fp64) on 4096^2 periodic grid", "value": value, "unit": "Mcell-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N}x{N} cells per GPU, periodic, {args.formulation} formulation + "
                                   f"{'Jacobian' if args.formulation == 'VectorInvariant' else 'divergence'}-form Lorentz forcing, "
                                   "Bickley-jet h/u + current-sheet A (BASELINE config 3)" if args.formulation == "VectorInvariant"
                                   else f"{N}x{N} cells per GPU, periodic, Conservative formulation + divergence-form Lorentz forcing, two-Gaussian A (BASELINE config 4 ICs)",
                       "step": "one RK3 time step = 3 x (fused tendency+substep kernel, halo fill of 4 fields)",
                       "kernels": "strict (oracle-order)" if args.strict else "fast",
                       "spin_up": "40 untimed steps before the warm-up steps (device clocks settle after ~30 ms of load)",
                       "decomposition": f"y-slabs x{world} (ring halo exchange, backend {args.backend}, overlapped; " + ("native swmhd_ring driver" if m._ring is not None else "torch.distributed p2p") + ")" if dec.ring else "single GPU",
                       "dt": args.dt, "finite": finite},
        }
        launches = [(a.elapsed_time(b), r) for a, b, r in m.tendency_events]
        if m._ring is not None:
            launches = m.ring_launch_times()
        if launches:
            ms = [t for t, _ in launches]
            kern_ms = float(np.mean(ms))
            kcells = N * float(np.mean([r for _, r in launches]))   # rank 0's launches (slab interior rows when N>1)
            achieved = TEND_BYTES_PER_CELL * kcells / (kern_ms * 1e-3) / 1e9
            # PMC traffic cannot be collected inside this process; the per-launch figure measured with rocprofv3 --pmc on this
            # same command (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 note) is committed
            # under profiles/ and echoed here when the workload matches
            traffic = None
            tp = os.path.join(ROOT, "profiles", "r01", "tendency_pmc_traffic.json")
            if os.path.exists(tp) and N == 4096 and args.formulation == "VectorInvariant" and not args.strict and not dec.ring:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch_corrected")
            # the launch also performs the fused RK3 substep: besides the 64 B/cell of SURVEY 8(d) it reads G- and writes the
            # new state (96 / 128 / 96 B/cell in stages 1 / 2 / 3); `achieved` stays on the conservative 64 B/cell figure
            fused_bytes = (96 + 128 + 96) / 3.0 * kcells
            line["roofline"] = {"bound": "hbm", "kernel": ("k_tendency_vi_march" if args.formulation == "VectorInvariant" else "k_tendency_cons_march")
                                          + " (fused RHS of the 4 prognostic fields incl. Lorentz force + RK3 substep)",
                                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                "traffic": traffic, "algorithmic_bytes_per_launch": TEND_BYTES_PER_CELL * kcells,
                                "avg_launch_ms": kern_ms, "launches_timed": len(ms),
                                "fused_substep_bytes_per_launch": fused_bytes, "achieved_incl_fused_substep": fused_bytes / (kern_ms * 1e-3) / 1e9,
                                "whole_step_GBps_on_544B": STEP_BYTES_PER_CELL * cells * args.steps / wall / 1e9,
                                **valu_floor(N, kcells / N, kern_ms, args),
                                "note": "fp64 WENO5 makes this kernel VALU-bound: 625 VALU instructions per wave-row put its fp64-VALU floor at ~370 us per launch (DESIGN.md 4.1)"}
        if world == 1:
            # the reference's own hot-path kernel (whole-field Lorentz force, 32 B/cell: read A,h, write Fx,Fy) on the same fields,
            # timed with HIP events outside the step's timed region
            op = S.lorentz_force_func if args.formulation == "VectorInvariant" else S.div_lorentz
            fld = {"A": m.solution["A"], "h": m.solution["h"]}
            out = (S.Field(g), S.Field(g))
            for _ in range(300):      # (back to settled clocks after the host-side pause above)
                op(g, fld, out=out, strict=args.strict)
            evs = []
            for _ in range(30):     # an event pair around every launch, as for the tendency kernel
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); op(g, fld, out=out, strict=args.strict); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            op_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
            op_bw = 32 * cells / (op_ms * 1e-3) / 1e9
            line["lorentz_operator"] = {"kernel": "k_lorentz_jacobian_march" if args.formulation == "VectorInvariant" else "k_lorentz_divergence_march",
                                        "bound": "hbm", "avg_launch_ms": op_ms, "achieved": op_bw, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": op_bw / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": 32 * cells}
        if world == 1:
            # the launch SURVEY 8(d)'s 64-B figure describes literally: tendencies only (read 4 fields, write 4 tendencies), no fused
            # substep -- same kernel template, MODE 4 -- timed like the operator above, outside the step's timed region
            m.tendency_events = None
            for _ in range(100):      # (back to settled clocks after the host-side pause above)
                m.calculate_tendencies()
            evs = []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); m.calculate_tendencies(); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            t_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
            t_bw = TEND_BYTES_PER_CELL * cells / (t_ms * 1e-3) / 1e9
            line["tendency_only_launch"] = {"what": "calculate_tendencies! alone (no fused substep): exactly 64 B/cell", "bound": "hbm",
                                            "avg_launch_ms": t_ms, "achieved": t_bw, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": t_bw / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": TEND_BYTES_PER_CELL * cells}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args, cfg)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    m.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
