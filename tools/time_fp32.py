"""fp32 vs fp64 speed of the fused RK3 step (same kernels, dtype template parameter): python tools/time_fp32.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for form, cfg in (("VectorInvariant", configs.config3_bickley()), ("Conservative", configs.config4_two_gaussians())):
    for dtype in (torch.float64, torch.float32):
        g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
        m = S.ShallowWaterModel(g, formulation=form, dtype=dtype)
        n1, n2 = m.names[:2]
        m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
        for _ in range(3): m.time_step(1e-4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.time_step(1e-4)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{form:16s} {str(dtype):14s} N={N}: step {ms:.3f} ms -> {N*N/ms/1e3:8.0f} Mcell-steps/s")
        del m
