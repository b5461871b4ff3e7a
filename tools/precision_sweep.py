#!/usr/bin/env python3
"""fp32 vs fp64 tolerance sweep (BASELINE config 5 / SURVEY.md 8(d)): both precisions start from identical fp64 initial
conditions; reports max|F32 - F64| / max|F64| after one RHS evaluation (operators and fused tendencies) and after 10 / 100
RK3 steps of the state.   python tools/precision_sweep.py [--size 2048] [--out profiles/r01/precision_sweep.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S  # noqa: E402
from swmhd_amd import configs  # noqa: E402


def rel(a32, a64):
    """max-norm difference relative to max|F64| (absolute difference when the fp64 field is identically zero)."""
    d, m = float((a32.double() - a64).abs().max()), float(a64.abs().max())
    return d / m if m > 0 else d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    N = args.size
    res = {"grid": f"{N}x{N}", "device": torch.cuda.get_device_name(0), "cases": {}}
    for form, cfg in (("VectorInvariant", configs.config3_bickley()), ("Conservative", configs.config4_two_gaussians())):
        g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
        dt = 0.2 * min(g.dx, g.dy) / 4.2
        ms = {}
        for name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            m = S.ShallowWaterModel(g, 9.81, 1.0, formulation=form, dtype=dtype)
            n1, n2 = m.names[:2]
            hfun = (lambda X, Y: cfg["h"](X, Y) + 0 * X)
            m.set(**{n1: (lambda X, Y: hfun(X, Y) * cfg["u"](X, Y)) if form == "Conservative" else cfg["u"],
                     n2: (lambda X, Y: hfun(X, Y) * cfg["v"](X, Y)) if form == "Conservative" else cfg["v"], "h": hfun, "A": cfg["A"]})
            ms[name] = m
        I = g.interior
        out = {}
        fields = lambda m: {"A": m.solution["A"], "h": m.solution["h"]}
        op = S.lorentz_force_func if form == "VectorInvariant" else S.div_lorentz
        F64, F32 = op(g, fields(ms["f64"])), op(g, fields(ms["f32"]))
        out["lorentz_operator_1_eval"] = [rel(a.data[I], b.data[I]) for a, b in zip(F32, F64)]
        for m in ms.values():
            m.calculate_tendencies()
        out["tendencies_1_eval"] = {n: rel(a.data[I], b.data[I]) for n, a, b in zip(ms["f64"].names, ms["f32"].Gn, ms["f64"].Gn)}
        done = 0
        for nsteps in (10, 100):
            for m in ms.values():
                for _ in range(nsteps - done):
                    m.time_step(dt)
                m.synchronize()
            done = nsteps
            out[f"state_after_{nsteps}_steps"] = {n: rel(ms["f32"].solution[n].data[I], ms["f64"].solution[n].data[I]) for n in ms["f64"].names}
        out["dt"] = dt
        out["energy_f64"] = ms["f64"].diagnostics()["total_energy"]
        out["energy_f32"] = ms["f32"].diagnostics()["total_energy"]
        res["cases"][form] = out
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        open(args.out, "w").write(txt)


if __name__ == "__main__":
    main()
