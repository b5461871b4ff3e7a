#!/usr/bin/env python3
"""Measure every BASELINE.json configuration that fits one GPU and write a JSON summary:
config 2 (1024^2 divergence), config 3 (4096^2 Jacobian, the headline), the per-GPU slabs of config 4 (8192x1024 divergence)
and config 5 (16384x2048 Jacobian, fp64 and fp32).   python tools/run_configs.py [--out profiles/r01/configs.json]"""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs


def timeit(fn, n):
    for _ in range(40): fn()        # (device clocks settle after ~30 ms of load)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name, Nx, Ny, cfg, form, dtype, dt, y_extent_scale=1.0):
    y0, y1 = cfg["domain"]["y"]
    g = S.RectilinearGrid(size=(Nx, Ny), x=cfg["domain"]["x"], y=(y0, y0 + (y1 - y0) * y_extent_scale))
    m = S.ShallowWaterModel(g, formulation=form, dtype=dtype)
    n1, n2 = m.names[:2]
    m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
    op = S.lorentz_force_func if form == "VectorInvariant" else S.div_lorentz
    out = (S.Field(g, dtype=dtype), S.Field(g, dtype=dtype))
    fld = {"A": m.solution["A"], "h": m.solution["h"]}
    bpe = 8 if dtype == torch.float64 else 4
    t_op = timeit(lambda: op(g, fld, out=out), 20)
    t_tend = timeit(m.calculate_tendencies, 10)
    t_step = timeit(lambda: m.time_step(dt), 10)
    cells = Nx * Ny
    d = m.diagnostics()
    r = {"grid": f"{Nx}x{Ny}", "formulation": form, "dtype": "f64" if bpe == 8 else "f32",
         "lorentz_operator_us": t_op * 1e3, "lorentz_operator_GBps": 4 * bpe * cells / t_op / 1e6,
         "tendency_kernel_us": t_tend * 1e3, "tendency_GBps_on_8fields": 8 * bpe * cells / t_tend / 1e6,
         "rk3_step_ms": t_step, "Mcell_steps_per_s": cells / t_step / 1e3, "finite": bool(np.isfinite(d["total_energy"]))}
    print(name, json.dumps(r), flush=True)
    return r


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); a = ap.parse_args()
    res = {"device": torch.cuda.get_device_name(0)}
    res["config2_1024sq_divergence_uniformBx"] = run("config2", 1024, 1024, configs.config2_uniform_bx(), "Conservative", torch.float64, 2e-4)
    res["config3_4096sq_jacobian_bickley"] = run("config3", 4096, 4096, configs.config3_bickley(), "VectorInvariant", torch.float64, 1e-4)
    res["config4_slab_8192x1024_divergence"] = run("config4", 8192, 1024, configs.config4_two_gaussians(), "Conservative", torch.float64, 5e-5, 1 / 8)
    res["config5_slab_16384x2048_jacobian_f64"] = run("config5-f64", 16384, 2048, configs.config3_bickley(), "VectorInvariant", torch.float64, 2e-5, 1 / 8)
    res["config5_slab_16384x2048_jacobian_f32"] = run("config5-f32", 16384, 2048, configs.config3_bickley(), "VectorInvariant", torch.float32, 2e-5, 1 / 8)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
