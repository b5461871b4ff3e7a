#!/usr/bin/env python3
"""Measure every BASELINE.json configuration that fits one GPU plus the strong-scaling slabs of the 4096^2 grid,
and write a JSON summary:  config 2 (1024^2 divergence), config 3 (4096^2 Jacobian, the headline), the per-GPU slabs of config 4
(8192 x 1024 divergence) and config 5 (16384 x 2048 Jacobian, fp64 and fp32), and 4096 x {2048, 1024, 512} slabs of config 3.
    python tools/run_configs.py [--out profiles/r03/configs.json] [--only name,name]"""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs, _lib


def timeit(fn, n, spin=40):
    for _ in range(spin): fn()        # (device clocks settle after ~30 ms of load)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name, Nx, Ny, cfg, form, dtype, ydom=None):
    y0, y1 = ydom or cfg["domain"]["y"]
    g = S.RectilinearGrid(size=(Nx, Ny), x=cfg["domain"]["x"], y=(y0, y1))
    m = S.ShallowWaterModel(g, formulation=form, dtype=dtype)
    n1, n2 = m.names[:2]
    m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
    dt = 0.2 * min(g.dx, g.dy) / 4.2
    op = S.lorentz_force_func if form == "VectorInvariant" else S.div_lorentz
    out = (S.Field(g, dtype=dtype), S.Field(g, dtype=dtype))
    fld = {"A": m.solution["A"], "h": m.solution["h"]}
    bpe = 8 if dtype == torch.float64 else 4
    t_op = timeit(lambda: op(g, fld, out=out), 50, 300)     # (300 launches first: the first ~40 ms of a process run 10-20 % slow)
    t_tend = timeit(m.calculate_tendencies, 20)
    t_step = timeit(lambda: m.time_steps(1, dt), 20)
    cells = Nx * Ny
    d = m.diagnostics()
    geo = _lib.tendency_launch_geometry(Nx, Ny, 1 if form == "VectorInvariant" else 0, bpe, 0)
    r = {"grid": f"{Nx}x{Ny}", "formulation": form, "dtype": "f64" if bpe == 8 else "f32",
         "lorentz_operator_us": t_op * 1e3, "lorentz_operator_GBps": 4 * bpe * cells / t_op / 1e6,
         "tendency_kernel_us": t_tend * 1e3, "tendency_GBps_on_8fields": 8 * bpe * cells / t_tend / 1e6,
         "rk3_step_ms": t_step, "Mcell_steps_per_s": cells / t_step / 1e3, "finite": bool(np.isfinite(d["total_energy"])),
         "launch_geometry": geo}
    print(name, json.dumps(r), flush=True)
    del m
    torch.cuda.empty_cache()
    return r


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--only", default=None)
    a = ap.parse_args()
    res = {"device": torch.cuda.get_device_name(0), "kernel_source_hash": _lib.source_hash(),
           "knobs": {k: v for k, v in os.environ.items() if k.startswith("SWMHD_")}}
    c3 = configs.config3_bickley()
    y0, y1 = c3["domain"]["y"]
    yc, Ly = 0.5 * (y0 + y1), y1 - y0
    slab = lambda rows: (yc - Ly * rows / 4096 / 2, yc + Ly * rows / 4096 / 2)
    jobs = [
        ("config2_1024sq_divergence_uniformBx", 1024, 1024, configs.config2_uniform_bx(), "Conservative", torch.float64, None),
        ("config3_4096sq_jacobian_bickley", 4096, 4096, c3, "VectorInvariant", torch.float64, None),
        ("config3_strong_slab_4096x2048", 4096, 2048, c3, "VectorInvariant", torch.float64, slab(2048)),
        ("config3_strong_slab_4096x1024", 4096, 1024, c3, "VectorInvariant", torch.float64, slab(1024)),
        ("config3_strong_slab_4096x512", 4096, 512, c3, "VectorInvariant", torch.float64, slab(512)),
        ("config3_1024sq_jacobian", 1024, 1024, c3, "VectorInvariant", torch.float64, None),
        ("config4_slab_8192x1024_divergence", 8192, 1024, configs.config4_slab(), "Conservative", torch.float64, None),
        ("config5_slab_16384x2048_jacobian_f64", 16384, 2048, configs.config5_bickley_slab(), "VectorInvariant", torch.float64, None),
        ("config5_slab_16384x2048_jacobian_f32", 16384, 2048, configs.config5_bickley_slab(), "VectorInvariant", torch.float32, None),
    ]
    only = set(a.only.split(",")) if a.only else None
    for name, Nx, Ny, cfg, form, dtype, ydom in jobs:
        if only and not any(o in name for o in only):
            continue
        res[name] = run(name, Nx, Ny, cfg, form, dtype, ydom)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
