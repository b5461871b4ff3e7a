#!/usr/bin/env python3
"""bench.py -- headline benchmark of the shallow-water-MHD tendency engine on MI355X.

Contract:  python bench.py --gpus N --steps K --warmup W  prints ONE JSON line (rank 0, stdout).
Metric (BASELINE.json): Mcell-steps/sec (fp64) on a 4096^2 periodic grid.  One step = one full RK3 time step of the model the
reference builds (SWMHD_example.jl:21-42): 3 x {fused tendency evaluation incl. the Lorentz force, RK3 substep of 4 fields, halo
fill}.  Default workload: BASELINE config 3 (4096 x 4096, Jacobian formulation = vector-invariant u,v,h + tracer A, Bickley-jet
h/u, fp64), synthetic initial condition resident in HBM before timing.

  --config 3|4|5   3: 4096^2 Jacobian (default); 4: 8192^2 divergence formulation; 5: 16384^2 Jacobian (2048 rows per GPU)
  --scaling weak|strong   N > 1: weak = every rank owns a full per-GPU slab (config 3: 4096 x 4096 per GPU, config 5: 16384 x 2048
                   per GPU, config 4: 8192 x 1024 per GPU); strong = the fixed global grid cut into N y-slabs (config 3: 4096 x 4096/N,
                   config 4: 8192 x 8192/N).  Defaults as BASELINE.json words them: strong for config 3 ("4096^2 periodic grid at 1/2/4/8")
                   and config 4 ("8192 x 8192 ... 8 x MI355X"), weak for config 5 ("weak-scaled across 8").  When --scaling is not given
                   and N > 1, the OTHER scaling is measured too, after the headline, same K and W, and reported in the same line as
                   "companion" (--no-companion skips it): one driver run yields both curves.
  N > 1 without a launcher (RANK unset): bench.py starts N worker processes itself (one per GPU) and forwards rank 0's line.

Halo rows move by RCCL send/recv between ring neighbours (native swmhd_ring driver), overlapped with the interior rows on a second
stream.  Every number in the `roofline` block is derived by the formula printed next to it from a value measured in this run or from
a file under profiles/ that is named in the block together with the kernel-source hash it was collected at.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0          # the same guide's measured copy rate: every fraction is quoted against both (BASELINE.md sections 3-4)
TEND_BYTES_PER_CELL = 64       # read u,v,h,A + write 4 tendencies, fp64
STAGE_BYTES_PER_CELL = (64, 128, 96)   # what the three FUSED stage launches move (fast builds): stage 1 reads U0, writes U1 -- no tendency store:
                                       # stage 2 forms G0 = (U1 - U0)/(dt gamma1) from the two states (swmhd.h SWMHD_GM_IS_PREV_STATE); stage 2
                                       # reads U1, U0 and writes U2, G1; stage 3 reads U2, G1 and writes U3.  288 B/cell-step (strict: 96/128/96)
PROFILE_DIR = os.path.join(ROOT, "profiles", "r03")
VALU_NS_PER_WAVE_INST = 2.05   # one fp64 wave-instruction per 2.05 ns per SIMD (tools/valu_probe.hip: 4 cycles at ~1.95 GHz under load)

CONFIGS = {
    3: dict(builder="config3_bickley", Nx=4096, Ny=4096, slab=4096, form="VectorInvariant", default_scaling="strong",
            text="Bickley-jet h/u + current-sheet A (BASELINE config 3)"),
    4: dict(builder="config4_two_gaussians", Nx=8192, Ny=8192, slab=1024, form="Conservative", default_scaling="strong",
            text="two-Gaussian A, h = 1, uh = vh = 0 (BASELINE config 4)"),
    5: dict(builder="config3_bickley", Nx=16384, Ny=16384, slab=2048, form="VectorInvariant", default_scaling="weak",
            text="Bickley-jet h/u + current-sheet A at 16384^2 (BASELINE config 5)"),
}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    p.add_argument("--scaling", default=None, choices=["weak", "strong"])
    p.add_argument("--size", dest="n", type=int, default=None, help="override the grid edge Nx (and Ny) of the configuration")
    p.add_argument("--formulation", default=None, choices=["VectorInvariant", "Conservative"], help="override the configuration's formulation")
    p.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    p.add_argument("--strict", action="store_true", help="time the oracle-order (bitwise) kernels instead of the fast ones")
    p.add_argument("--dt", type=float, default=None)
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    p.add_argument("--backend", default="nccl")
    p.add_argument("--torch-ring", action="store_true", help="multi-GPU: exchange through torch.distributed p2p instead of the native ring")
    p.add_argument("--force-ring", action="store_true",
                   help="N=1 rehearsal of the multi-GPU step: y halos through the RCCL ring exchange (sends to self) + overlap")
    p.add_argument("--no-companion", action="store_true", help="N > 1: do not also measure the other scaling mode")
    p.add_argument("--launch-timeout", type=float, default=540.0,
                   help="self-launcher (N > 1 without torchrun): wall-clock deadline in seconds for the whole run; on expiry the workers "
                        "are stopped, the ranks still alive and the tails of their stderr are printed, exit code 124")
    p.add_argument("--rendezvous-only", action="store_true",
                   help="create the process group, barrier, print {'launcher': 'ok', ...} and exit (tests the launcher without a GPU)")
    return p.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no torchrun around it
# ------------------------------------------------------------------------------------------------------------------------------
def _tail(path, n=15):
    try:
        with open(path, errors="replace") as f:
            return "".join(f.readlines()[-n:]).rstrip()
    except OSError:
        return ""


def self_launch(args):
    """Start N fresh worker processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set) BEFORE anything in this process touches the GPU,
    forward rank 0's stdout (the JSON line), fail if any worker fails.  Never re-execs or restarts a GPU-initialised process.
    A wall-clock deadline (--launch-timeout) bounds the run: a rank stuck in communicator creation or in its first exchange would
    otherwise block the launcher until the driver's own kill, with nothing written.  Worker stderr goes to per-rank files whose
    tails are printed on failure (rank 0's is forwarded whole on success)."""
    import socket, tempfile
    N = args.gpus
    if args.backend == "nccl":
        import torch   # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < N:
            print(f"bench.py: --gpus {N} requested but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, errs = [], []
    logdir = tempfile.mkdtemp(prefix="swmhd_bench_")
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(os.path.join(logdir, f"rank{r}.stderr"))
        with open(errs[-1], "wb") as ef:
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef))
    rc, why = 0, ""
    deadline = time.monotonic() + args.launch_timeout
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = bad[0][1], f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            alive = [r for r, c in enumerate(codes) if c is None]
            rc, why = 124, f"--launch-timeout {args.launch_timeout:g} s expired with rank(s) {alive} still running"
            break
        time.sleep(0.05)
    if rc != 0:   # a failed or stuck rank leaves the others waiting in a collective: stop exactly the processes started here
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()
    out = procs[0].stdout.read().decode()
    for p in procs:
        try:
            p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill(); p.wait()
    if rc == 0:
        sys.stderr.write(open(errs[0], errors="replace").read())
        sys.stdout.write(out); sys.stdout.flush()
    else:
        print(f"bench.py: {why}; no result line.  Last stderr lines per rank:", file=sys.stderr)
        for r, e in enumerate(errs):
            print(f"--- rank {r} (exit {procs[r].returncode}) ---\n{_tail(e)}", file=sys.stderr)
    for e in errs:
        try:
            os.remove(e)
        except OSError:
            pass
    try:
        os.rmdir(logdir)
    except OSError:
        pass
    return rc


# ------------------------------------------------------------------------------------------------------------------------------
# workload
# ------------------------------------------------------------------------------------------------------------------------------
def workload(args, world):
    """(cfg dict, Nx, Ny_global, Ny_local, formulation, scaling, y-domain) of this run."""
    from swmhd_amd import configs
    c = CONFIGS[args.config]
    cfg = getattr(configs, c["builder"])()
    form = args.formulation or c["form"]
    if args.formulation and args.formulation != c["form"]:
        cfg = configs.config3_bickley() if form == "VectorInvariant" else configs.config4_two_gaussians()
    scaling = args.scaling or c["default_scaling"]
    Nx = args.n or c["Nx"]
    full_Ny = args.n or c["Ny"]                  # the configuration's global grid
    slab = min(args.n or c["slab"], full_Ny)     # rows per GPU when weak-scaled
    y0, y1 = cfg["domain"]["y"]
    yc, Ly = 0.5 * (y0 + y1), (y1 - y0)
    if scaling == "strong":
        Ny_global = full_Ny
        ydom = (y0, y1)
    else:
        # weak: every GPU owns `slab` rows at the configuration's dy; the N slabs are centred on the configuration's y axis (config 3:
        # N = 1 is exactly the 4096^2 Bickley domain; configs 4/5 at N = 8 are exactly the 8192^2 / 16384^2 domains)
        Ny_global = slab * world
        ext = Ly * Ny_global / full_Ny
        ydom = (yc - ext / 2, yc + ext / 2)
    if Ny_global % world:
        raise SystemExit(f"bench.py: Ny={Ny_global} is not divisible by --gpus {world}")
    return cfg, Nx, Ny_global, Ny_global // world, form, scaling, ydom


def companion_run(args, world, rank, dist, scaling):
    """The same K timed steps (after 40 + W untimed ones) on the workload of the OTHER scaling mode; all ranks take part.  Returns
    the entries of the "companion" object (rank 0 prints them)."""
    import copy
    import torch
    import swmhd_amd as S
    a2 = copy.copy(args); a2.scaling = scaling
    cfg, Nx, Ny_global, Ny_local, form, _, ydom = workload(a2, world)
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    dec = S.SlabDecomposition(Ny_global, world, rank, force_ring=args.force_ring)
    g = dec.local_grid(S.RectilinearGrid, Nx, x=cfg["domain"]["x"], y=ydom, halo=dec.ring_halo())
    dt = args.dt if args.dt is not None else 0.2 * min(g.dx, g.dy) / 4.2
    m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=args.strict, decomp=dec, native_ring=not args.torch_ring, dtype=dtype)
    n1, n2 = m.names[:2]
    m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
    for _ in range(40 + args.warmup):
        m.time_step(dt)
    m.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.time_step(dt)
    m.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = t.item()
    finite = all(torch.isfinite(f.data).all().item() for f in m.fields)
    m.close()
    return {"scaling": scaling, "value": Nx * Ny_global * args.steps / wall / 1e6, "unit": "Mcell-steps/s", "ms_per_step": wall / args.steps * 1e3,
            "steps": args.steps, "warmup": args.warmup, "workload": f"global grid {Nx}x{Ny_global} ({Nx}x{Ny_local} cells per GPU)",
            "finite": finite, "timing": "as the headline: barrier + synchronize on both sides, max over ranks"}


def box_probe(torch, _lib):
    """What THIS box sustains right now, measured in-process right after the timed region: the HBM rate of a plain two-array copy
    (torch's copy kernel, 1 GiB read + 1 GiB written) and the fp64 issue rate of a SIMD under load (swmhd_probe_fp64_issue).  The boxes
    of the pool differ by several per cent in both; kernel fractions are quoted against these as well as against the spec figures."""
    import ctypes
    out = {}
    src = torch.empty(1 << 27, dtype=torch.float64, device="cuda").normal_()
    dst = torch.empty_like(src)
    for _ in range(20):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    out["hbm_copy_GBps"] = 2 * src.numel() * 8 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    out["hbm_copy_what"] = "torch copy of a 1-GiB fp64 array (bytes read + written / time), 20 launches in one event pair"
    gb = ctypes.c_float(0)
    if _lib.lib().swmhd_probe_copy(dst.data_ptr(), src.data_ptr(), src.numel() * 8, 20, ctypes.byref(gb), torch.cuda.current_stream().cuda_stream) == 0:
        out["hbm_copy_oneshot_GBps"] = gb.value
        out["hbm_copy_oneshot_what"] = ("swmhd_probe_copy: one 16-byte element per thread, workgroups in address order -- the best case of the box; "
                                        "persistent kernels (torch's copy, hipMemcpy, anything that carries state along a direction) reach ~20 % less")
    ns = ctypes.c_float(0)
    scratch = torch.zeros(8, dtype=torch.float64, device="cuda")
    rc = _lib.lib().swmhd_probe_fp64_issue(scratch.data_ptr(), ctypes.byref(ns), torch.cuda.current_stream().cuda_stream)
    if rc == 0 and ns.value > 0:
        out["fp64_ns_per_wave_instruction"] = ns.value
        out["fp64_shader_MHz_under_load"] = 4000.0 / ns.value
        out["fp64_what"] = "swmhd_probe_fp64_issue: 3 waves per SIMD of independent fma chains; 4 cycles per wave-instruction on a 16-lane SIMD"
    del src, dst
    return out


def committed(name):
    """A JSON file under profiles/r03 (or None) plus whether it was collected at the kernel sources that are running now."""
    from swmhd_amd import _lib
    path = os.path.join(PROFILE_DIR, name)
    if not os.path.exists(path):
        return None, {"file": os.path.relpath(path, ROOT), "status": "absent"}
    d = json.load(open(path))
    now = _lib.source_hash()
    src = {"file": os.path.relpath(path, ROOT), "kernel_source_hash": d.get("kernel_source_hash"), "git_head": d.get("git_head"),
           "running_kernel_source_hash": now}
    src["status"] = "current" if d.get("kernel_source_hash") == now else "stale: kernel sources changed since it was collected"
    return (d if src["status"] == "current" else None), src


def parity_block(args):
    """Achieved errors of the running kernels against the oracle / the reference's plots, from the files the GPU tests write
    (tests/test_fullsize_gpu.py -> fullsize_parity.json, tests/test_reference_plots.py -> plot_parity.jsonl), echoed only while
    they were collected at the running kernel sources."""
    out = {"bar": "strict kernels: bit-identical to the CPU oracle (every entry point, fp64 and fp32, whole RK3 steps); fast kernels: "
                  "max|dG| <= 1e-13 (fp64) / 1e-4 (fp32) x max(max|G|, S), S = the largest term summed into the output "
                  "(include/swmhd.h, 'Tolerances'); oracle pinned to the reference's 12 energy plots at reading accuracy",
           "oracle_pin": "tests/test_reference_plots.py + tests/golden/plot_readings.json"}
    fp, src = committed("fullsize_parity.json")
    out["fullsize_source"] = src
    if fp:
        key = {3: "config3", 4: "config4_slab", 5: "config5_slab"}.get(args.config, "config3") + "/" + args.dtype
        rows = {k: v for k, v in fp.items() if k.startswith(key) and isinstance(v, dict)}
        worst_scale = max((f["err_over_scale"] for v in rows.values() for f in v.values() if isinstance(f, dict) and "err_over_scale" in f), default=None)
        worst_maxg = max((f["err_over_maxG"] for v in rows.values() for f in v.values() if isinstance(f, dict) and "err_over_maxG" in f), default=None)
        out["fast_vs_oracle_this_workload"] = {"cases": sorted(rows), "worst_err_over_term_scale": worst_scale, "worst_err_over_max_abs_G": worst_maxg}
    pp, psrc = committed("plot_parity.json")
    out["plots_source"] = psrc
    if pp:
        out["plots"] = {"runs": pp.get("runs"), "worst_ratio_to_reading_tolerance": pp.get("worst_ratio"), "slack_allowed": pp.get("slack")}
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, cfg, form, dx, dy, dt):
    """The oracle's RK3 step (C restatement: the reference's per-cell Lorentz functions with their unshared composition + the
    restated Oceananigans RHS) on the host cores, on a bounded sample of the same workload: a 2048 x 1024 periodic block at the
    workload's dx, dy and fields (2.1 Mcell, 16 arrays of 17 MB: not cache-resident, like the workload), stepped until the time
    budget is used -- on all cores of the box's share, then on ONE thread."""
    import numpy as np
    from oracle import oracle as O
    import swmhd_amd as S
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)   # the box's CPU share for one GPU
    Nx, Ny = 2048, 1024
    g = S.RectilinearGrid(size=(Nx, Ny), x=(-Nx * dx / 2, Nx * dx / 2), y=(-Ny * dy / 2, Ny * dy / 2))
    fcode = 1 if form == "VectorInvariant" else 0
    names = [("u", ("Face", "Center")), ("v", ("Center", "Face")), ("h", ("Center", "Center")), ("A", ("Center", "Center"))]
    q = []
    for nm, loc in names:
        X, Y = g.nodes(loc)
        q.append(O.fill_halo_periodic(np.ascontiguousarray(cfg[nm](X, Y) + 0 * X), Nx, Ny, 3, 3))

    def rate(nthreads, budget):
        work = O.time_step(*q, Nx, Ny, 3, 3, dx, dy, dt, fcode, 2 - fcode, nthreads=nthreads)   # warm
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < budget or reps < 2:
            O.time_step(*q, Nx, Ny, 3, 3, dx, dy, dt, fcode, 2 - fcode, nthreads=nthreads, work=work); reps += 1
        return Nx * Ny * reps / (time.perf_counter() - t0) / 1e6, reps

    v_all, reps_all = rate(cores, args.cpu_seconds * 2 / 3)
    v_one, reps_one = rate(1, args.cpu_seconds / 3)
    return {"value": v_all, "unit": "Mcell-steps/s", "cores": cores, "kind": "port",
            "value_1_thread": v_one, "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
            "sample": f"{Nx}x{Ny} periodic block at the workload's dx,dy and fields; {reps_all} RK3 steps on {cores} OpenMP threads, "
                      f"{reps_one} on 1 thread; C oracle (reference's unshared per-cell Lorentz composition + restated Oceananigans "
                      "RHS, gcc -O3 -ffp-contract=off); Julia/Oceananigans are not available on the box"}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE line, the JSON: everything else any library prints there (RCCL announces its version on stdout
    # when a communicator is created) is sent to stderr.
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    if os.environ.get("SWMHD_BENCH_TEST_HANG_RANK") == str(rank):      # tests/test_bench_launcher.py: a rank that never arrives
        print(f"bench.py: rank {rank} sleeping (SWMHD_BENCH_TEST_HANG_RANK)", file=sys.stderr, flush=True)
        time.sleep(3600)
    import numpy as np
    import torch
    dist = None
    if world > 1 or args.force_ring or args.rendezvous_only:
        import torch.distributed as dist
        kw = {}
        if args.backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
            kw = {"device_id": torch.device("cuda", torch.cuda.current_device())}
        if "RANK" not in os.environ:   # --force-ring started without a launcher
            kw.update(init_method=f"tcp://127.0.0.1:{29500 + os.getpid() % 2000}", rank=0, world_size=1)
        dist.init_process_group(args.backend, **kw)
    if args.rendezvous_only:
        dist.barrier()
        if rank == 0:
            json_out.write(json.dumps({"launcher": "ok", "n_gpus": world, "backend": args.backend}) + "\n"); json_out.flush()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))

    import swmhd_amd as S
    from swmhd_amd import _lib
    cfg, Nx, Ny_global, Ny_local, form, scaling, ydom = workload(args, world)
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    dec = S.SlabDecomposition(Ny_global, world, rank, force_ring=args.force_ring)
    g = dec.local_grid(S.RectilinearGrid, Nx, x=cfg["domain"]["x"], y=ydom, halo=dec.ring_halo())
    dt = args.dt if args.dt is not None else 0.2 * min(g.dx, g.dy) / 4.2      # CFL 0.2 on sqrt(g h) + U ~ 4.2
    m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, strict=args.strict, decomp=dec, native_ring=not args.torch_ring, dtype=dtype)
    n1, n2 = m.names[:2]
    m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})

    # The device needs ~30 ms of sustained load before its clocks settle (a step measured right after start-up is 6-10 % slower
    # than the same step a hundred steps later): spin the step 40 times before the W warm-up steps the contract counts.
    for _ in range(40 + args.warmup):
        m.time_step(dt)
    m.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    m.tendency_events = None if os.environ.get("SWMHD_BENCH_NO_LAUNCH_TIMING") else []   # (A/B knob: what the per-launch events cost)
    if m._ring is not None and not os.environ.get("SWMHD_BENCH_NO_LAUNCH_TIMING"):
        m.ring_time_launches(3 * args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.time_step(dt)
    m.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = t.item()
    finite = all(torch.isfinite(f.data).all().item() for f in m.fields)
    box = box_probe(torch, _lib) if (rank == 0 and world == 1) else None      # right after the timed region: clocks are settled
    companion = None
    if (world > 1 or args.force_ring) and args.scaling is None and args.n is None and not args.no_companion:
        # (--force-ring: one rank, both workloads coincide -- it only rehearses this code path on a one-GPU box)
        try:
            companion = companion_run(args, world, rank, dist, "weak" if scaling == "strong" else "strong")
        except Exception as exc:   # the headline line must survive a failure here (an error every rank hits alike)
            companion = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        cells_global = Nx * Ny_global
        value = cells_global * args.steps / wall / 1e6
        lor = "Jacobian" if form == "VectorInvariant" else "divergence"
        bpe = 8 if args.dtype == "f64" else 4
        line = {
            "metric": f"Mcell-steps/sec ({'fp64' if bpe == 8 else 'fp32'}) on "
                      + ("4096^2" if (Nx, Ny_global) == (4096, 4096) else f"{Nx}x{Ny_global}") + " periodic grid",
            "value": value, "unit": "Mcell-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: global grid {Nx}x{Ny_global} ({Nx}x{Ny_local} cells per GPU), periodic, "
                                   f"{form} formulation + {lor}-form Lorentz forcing, {CONFIGS[args.config]['text']}",
                       "step": "one RK3 time step = 3 fused tendency+substep launches (periodic images gathered on read: no halo launch between stages; "
                               "fast builds: the first stage stores no tendencies, the second forms G- from the two states)",
                       "kernels": "strict (oracle-order)" if args.strict else "fast",
                       "spin_up": "40 untimed steps before the warm-up steps (device clocks settle after ~30 ms of load)",
                       "decomposition": (f"y-slabs x{world} (ring halo exchange, backend {args.backend}, overlapped; "
                                         + ("native swmhd_ring driver" if m._ring is not None else "torch.distributed p2p") + ")")
                                        if dec.ring else "single GPU",
                       "dt": dt, "finite": finite, "kernel_source_hash": _lib.source_hash()},
        }
        if companion is not None:
            line["companion"] = companion
        default_workload = (args.config == 3 and world == 1 and not args.strict and not dec.ring and args.n is None
                            and args.formulation is None and bpe == 8)
        ks, ksrc = None, None
        launches = [(a.elapsed_time(b), r) for a, b, r in (m.tendency_events or [])]
        if m._ring is not None:
            launches = m.ring_launch_times()
        if launches:
            ms = [t for t, _ in launches]
            kern_ms = float(np.mean(ms))
            krows = float(np.mean([r for _, r in launches]))      # rank 0's launches (slab interior rows when N > 1)
            kcells = Nx * krows
            tend_bytes = TEND_BYTES_PER_CELL * bpe // 8
            achieved = tend_bytes * kcells / (kern_ms * 1e-3) / 1e9
            stage_bytes = (96, 128, 96) if args.strict else STAGE_BYTES_PER_CELL
            fused_bytes = sum(stage_bytes) / 3.0 * bpe / 8 * kcells
            kname = "k_tendency_vi_march" if form == "VectorInvariant" else "k_tendency_cons_march"
            roof = {"bound": "hbm", "kernel": kname + " (fused RHS of the 4 prognostic fields incl. Lorentz force + RK3 substep)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "frac_of_measured_copy_6290": achieved / HBM_COPY_GBS,
                    "frac_of_this_box_copy": (achieved / box["hbm_copy_GBps"]) if box else None,
                    "frac_of_this_box_oneshot_copy": (achieved / box["hbm_copy_oneshot_GBps"]) if (box and "hbm_copy_oneshot_GBps" in box) else None,
                    "formula": f"achieved = {tend_bytes} B/cell (SURVEY 8(d): 4 fields read + 4 tendencies written) x cells_per_launch / avg_launch_ms",
                    "algorithmic_bytes_per_launch": tend_bytes * kcells, "cells_per_launch": kcells,
                    "avg_launch_ms": kern_ms, "launches_timed": len(ms),
                    "timing": "HIP events around every stage launch inside the timed region, on the launch stream",
                    "fused_stage": {"bytes_per_cell_stage_1_2_3": [b * bpe // 8 for b in stage_bytes],
                                    "mean_bytes_per_launch": fused_bytes, "achieved": fused_bytes / (kern_ms * 1e-3) / 1e9,
                                    "frac": fused_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "frac_of_measured_copy_6290": fused_bytes / (kern_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                                    "frac_of_this_box_copy": (fused_bytes / (kern_ms * 1e-3) / 1e9 / box["hbm_copy_GBps"]) if box else None,
                                    "frac_of_this_box_oneshot_copy": (fused_bytes / (kern_ms * 1e-3) / 1e9 / box["hbm_copy_oneshot_GBps"]) if (box and "hbm_copy_oneshot_GBps" in box) else None,
                                    "what": "bytes the fused stage launches really move (new state written; no G store in stages 1 and 3: "
                                            "stage 2 takes G- from the two states it reads; G- read in stage 3); `achieved` above stays on the "
                                            "64-B tendency figure"},
                    "whole_step_bytes_per_cell": sum(stage_bytes) * bpe // 8,
                    "whole_step_GBps": sum(stage_bytes) * bpe / 8 * cells_global * args.steps / wall / 1e9}
            # HBM traffic from the PMC counters cannot be collected inside this process: the per-launch figure measured with
            # rocprofv3 --pmc on this same command (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950
            # note of the microarchitecture guide) lives under profiles/ and is echoed ONLY while the kernel sources are unchanged.
            tr, tsrc = committed("tendency_pmc_traffic.json")
            roof["traffic"] = tr.get("hbm_bytes_per_launch_corrected") if (tr and default_workload) else None
            roof["traffic_source"] = tsrc if default_workload else {"status": "not collected for this workload"}
            ks, ksrc = committed("fullstep_kernel_stats.json")
            if ks and default_workload:
                roof["rocprof_avg_launch_ms"] = ks.get("tendency_stage_mean_ms")
                roof["rocprof_source"] = ksrc
            # secondary bound (SURVEY 8(d) asks for it): the fp64 vector-ALU floor of the marching kernel =
            # VALU instructions per wave-row (PMC SQ_INSTS_VALU / wave-rows, committed) x wave-rows / SIMDs x ns per instruction
            geo = _lib.tendency_launch_geometry(Nx, int(round(krows)), 1 if form == "VectorInvariant" else 0, bpe,
                                                _lib.LEAVE_ROOM if dec.ring else 0)
            roof["launch_geometry"] = geo
            vp, vsrc = committed("tendency_pmc_valu.json")
            if geo["kind"] == 2 and not args.strict:
                waves = geo["threads"] // 64
                wave_rows = geo["nstrips"] * waves * krows
                valu = {"source": vsrc, "wave_rows_per_launch": wave_rows, "simds": geo["cus"] * 4,
                        "ns_per_wave_instruction_per_simd": (box or {}).get("fp64_ns_per_wave_instruction", VALU_NS_PER_WAVE_INST),
                        "ns_source": "measured in this run (box.fp64_ns_per_wave_instruction)" if (box or {}).get("fp64_ns_per_wave_instruction") else "tools/valu_probe.hip (2.05)",
                        "formula": "floor_ms = valu_insts_per_wave_row x wave_rows_per_launch / simds x ns_per_wave_instruction"}
                if vp and default_workload:
                    ipr = vp["valu_insts_per_wave_row"]
                    floor_ms = ipr * wave_rows / (geo["cus"] * 4) * valu["ns_per_wave_instruction_per_simd"] * 1e-6
                    valu.update({"valu_insts_per_wave_row": ipr, "floor_ms": floor_ms, "frac_of_floor": floor_ms / kern_ms})
                    rate = (box or {}).get("hbm_copy_GBps", 5100.0) * 1e9     # what a persistent kernel streams at on this box (torch's copy)
                    stream_ms = fused_bytes / rate * 1e3
                    roof["binding_bound"] = "fp64-valu" if floor_ms >= stream_ms else "hbm-streaming"
                    roof["binding_bound_note"] = (f"mean over the three stage launches: fp64 issue floor {floor_ms:.3f} ms vs {stream_ms:.3f} ms for the "
                                                  f"fused-stage bytes at this box's persistent-copy rate ({rate / 1e12:.2f} TB/s); stage 1 (64 B/cell) sits on "
                                                  "the first, stage 2 (128 B/cell) on the second; the 64-B HBM figure at 8 TB/s would need "
                                                  f"{tend_bytes * kcells / 8e12 * 1e3:.3f} ms")
                roof["valu"] = valu
            line["roofline"] = roof
        if box:
            line["box"] = box

        def timed_launches(fn, n_spin, K):
            """K back-to-back launches inside ONE event pair (settled clocks: n_spin launches first)."""
            for _ in range(n_spin):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(K):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / K

        if world == 1 and not dec.ring:
            cells = Nx * Ny_local
            ks, ksrc = committed("operators_kernel_stats.json")
            # the reference's own hot-path kernel (whole-field Lorentz force, 32 B/cell fp64: read A,h, write Fx,Fy) on the same fields
            op = S.lorentz_force_func if form == "VectorInvariant" else S.div_lorentz
            fld = {"A": m.solution["A"], "h": m.solution["h"]}
            out = (S.Field(g, dtype=dtype), S.Field(g, dtype=dtype))
            op_ms = timed_launches(lambda: op(g, fld, out=out, strict=args.strict), 300, 50)
            op_bytes = 4 * bpe * cells
            per = []                                   # the same launch timed one by one (events without the system-scope fence)
            for _ in range(50):
                ea, eb = _lib.TimingEvent(), _lib.TimingEvent()
                ea.record(); op(g, fld, out=out, strict=args.strict); eb.record()
                per.append((ea, eb))
            torch.cuda.synchronize()
            per = sorted(a.elapsed_time(b) for a, b in per)
            opk = "k_lorentz_jacobian_march" if form == "VectorInvariant" else "k_lorentz_divergence_march"
            line["lorentz_operator"] = {"kernel": opk, "bound": "hbm", "avg_launch_ms": op_ms, "achieved": op_bytes / (op_ms * 1e-3) / 1e9,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": op_bytes / (op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "frac_of_measured_copy_6290": op_bytes / (op_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                                        "frac_of_this_box_copy": (op_bytes / (op_ms * 1e-3) / 1e9 / box["hbm_copy_GBps"]) if box else None,
                                        "frac_of_this_box_oneshot_copy": (op_bytes / (op_ms * 1e-3) / 1e9 / box["hbm_copy_oneshot_GBps"]) if (box and "hbm_copy_oneshot_GBps" in box) else None,
                                        "per_launch_ms_min_median_max": [per[0], per[len(per) // 2], per[-1]],
                                        "algorithmic_bytes_per_launch": op_bytes, "timing": "50 back-to-back launches inside one HIP event pair",
                                        "rocprof_avg_launch_ms": (ks or {}).get(opk + "_mean_ms") if default_workload else None,
                                        "rocprof_source": ksrc if default_workload else None}
            # the launch SURVEY 8(d)'s 64-B figure describes literally: tendencies only (no fused substep), same kernel template
            m.tendency_events = None
            t_ms = timed_launches(m.calculate_tendencies, 100, 50)
            t_bytes = TEND_BYTES_PER_CELL * bpe // 8 * cells
            line["tendency_only_launch"] = {"what": "calculate_tendencies! alone (no fused substep): exactly the 64 B/cell of SURVEY 8(d)",
                                            "bound": "hbm", "avg_launch_ms": t_ms, "achieved": t_bytes / (t_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": t_bytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "frac_of_measured_copy_6290": t_bytes / (t_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                                            "frac_of_this_box_copy": (t_bytes / (t_ms * 1e-3) / 1e9 / box["hbm_copy_GBps"]) if box else None,
                                            "algorithmic_bytes_per_launch": t_bytes, "timing": "50 back-to-back launches inside one HIP event pair"}
        line["parity"] = parity_block(args)
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args, cfg, form, g.dx, g.dy, dt)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    m.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
