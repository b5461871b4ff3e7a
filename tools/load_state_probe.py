#!/usr/bin/env python3
"""Does what ran BEFORE decide how fast the streaming Lorentz operator runs?  bench.py times it after ~160 RK3 steps of fp64-heavy
load (125 us on one box), tools/run_configs.py right after start-up (109 us in the same call on the same box).  This probe times the
operator, the plain copy and the fp64 issue rate: cold, right after 300 heavy steps, and again after idling.
    python tools/load_state_probe.py"""
import sys, os, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs, _lib

N = 4096
cfg = configs.config3_bickley()
g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
m = S.ShallowWaterModel(g, formulation="VectorInvariant")
m.set(u=cfg["u"], v=cfg["v"], h=lambda X, Y: cfg["h"](X, Y) + 0 * X, A=cfg["A"])
out = (S.Field(g), S.Field(g))
fld = {"A": m.solution["A"], "h": m.solution["h"]}
src = torch.empty(1 << 27, dtype=torch.float64, device="cuda").normal_(); dst = torch.empty_like(src)
scratch = torch.zeros(8, dtype=torch.float64, device="cuda")


def ev(fn, K, spin=0):
    for _ in range(spin): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


def state(tag, spin):
    op = ev(lambda: S.lorentz_force_func(g, fld, out=out), 50, spin) * 1e3
    cp = 2 * src.numel() * 8 / (ev(lambda: dst.copy_(src), 20, 5) * 1e-3) / 1e12
    ns = ctypes.c_float(0)
    _lib.lib().swmhd_probe_fp64_issue(scratch.data_ptr(), ctypes.byref(ns), torch.cuda.current_stream().cuda_stream)
    op2 = ev(lambda: S.lorentz_force_func(g, fld, out=out), 50, 0) * 1e3
    print(f"{tag:44s} operator {op:6.1f} us ({32 * N * N / op / 1e6:.2f} TB/s)  copy {cp:.2f} TB/s  fp64 {ns.value:.3f} ns/inst ({4000 / ns.value:.0f} MHz)  operator again {op2:6.1f} us", flush=True)


state("cold (first launches of the process)", 0)
state("after 300 operator launches", 300)
t = ev(lambda: m.time_step(1e-4), 300) ; print(f"300 RK3 steps: {t:.3f} ms/step")
state("right after 300 heavy RK3 steps", 0)
t = ev(lambda: m.time_step(1e-4), 1500); print(f"1500 RK3 steps: {t:.3f} ms/step")
state("right after 1500 more RK3 steps (~2 s of load)", 0)
time.sleep(3.0)
state("after 3 s idle", 0)
state("after 300 operator launches", 300)
