#!/usr/bin/env python3
"""Ring-of-ONE rehearsal of the multi-GPU step at strong-scaled slab sizes (one GPU: every send goes to self through RCCL, everything
but the xGMI hop is real): RK3 step time through the native ring driver vs the plain periodic single-GPU step of the same slab.
    python tools/ring_rehearsal.py [--out profiles/r03/ring_rehearsal.json]"""
import argparse, json, os, socket, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs


def timeit(fn, n, spin=60):
    for _ in range(spin): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--only", type=int, default=0); ap.add_argument("--config", type=int, default=3, choices=[3, 4]); a = ap.parse_args()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = configs.config3_bickley() if a.config == 3 else configs.config4_two_gaussians()
    NX, FULL, form = (4096, 4096, "VectorInvariant") if a.config == 3 else (8192, 8192, "Conservative")
    y0, y1 = cfg["domain"]["y"]; yc, Ly = 0.5 * (y0 + y1), y1 - y0
    res = {"device": torch.cuda.get_device_name(0), "workload": f"config {a.config} fields on {NX} x Ny slabs, fp64, {form}"}
    for Ny in ((a.only,) if a.only else ((4096, 2048, 1024, 512) if a.config == 3 else (4096, 2048, 1024))):
        ydom = (yc - Ly * Ny / FULL / 2, yc + Ly * Ny / FULL / 2)
        out, models = {}, {}
        import time, statistics
        for tag, ring in (("plain", False), ("ring_of_one", True)):
            dec = S.SlabDecomposition(Ny, 1, 0, force_ring=ring)
            g = dec.local_grid(S.RectilinearGrid, NX, x=cfg["domain"]["x"], y=ydom, halo=dec.ring_halo())
            m = S.ShallowWaterModel(g, formulation=form, decomp=dec)
            n1, n2 = m.names[:2]
            m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
            models[tag] = (m, 0.2 * min(g.dx, g.dy) / 4.2)
            out[tag + "_native_ring"] = m._ring is not None
        # the two models alternate, 5 rounds of 40 steps each (one C call per step); the MEDIAN of the rounds is reported: box-to-box and
        # run-to-run differences of a few % would otherwise drown the ring's overhead
        rounds = {"plain": [], "ring_of_one": []}
        for r in range(5):
            for tag in ("plain", "ring_of_one"):
                m, dt = models[tag]
                rounds[tag].append(timeit(lambda: m.time_steps(1, dt), 40, spin=60 if r == 0 else 10))
        for tag in rounds:
            m, dt = models[tag]
            out[tag + "_ms"] = statistics.median(rounds[tag])
            out[tag + "_ms_rounds"] = rounds[tag]
            # host side: wall time to ENQUEUE 40 steps (no synchronisation inside), one C call per step and one call for all 40
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): m.time_steps(1, dt)
            out[tag + "_enqueue_ms_per_step"] = (time.perf_counter() - t0) / 40 * 1e3
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.time_steps(40, dt)
            out[tag + "_enqueue_ms_per_step_one_call"] = (time.perf_counter() - t0) / 40 * 1e3
            torch.cuda.synchronize()
            out[tag + "_ms_one_call"] = timeit(lambda: m.time_steps(40, dt), 3, spin=2) / 40
        for tag in models:
            m, _ = models[tag]
            m.synchronize(); m.close()
        del models
        torch.cuda.empty_cache()
        out["ring_over_plain"] = out["ring_of_one_ms"] / out["plain_ms"]
        res[f"{NX}x{Ny}"] = out
        print(f"{NX}x{Ny}", json.dumps(out), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
