"""Where does the row-marching kernel overtake the LDS-tiled kernel?  Full RK3 step (no per-launch events), both formulations."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
def step_ms(N, form, kern):
    cfg = configs.config3_bickley() if form == "VectorInvariant" else configs.config4_two_gaussians()
    g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
    m = S.ShallowWaterModel(g, formulation=form, kernel=kern)
    n1, n2 = m.names[:2]
    m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
    dt = 1e-4 * 4096 / N * 0.25
    n = max(60, int(40e-3 / 2e-4))
    for _ in range(n): m.time_step(dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): m.time_step(dt)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 100
for form in ("VectorInvariant", "Conservative"):
    for N in (384, 512, 640, 768, 896, 1024, 1280):
        t, mr = step_ms(N, form, "tile"), step_ms(N, form, "march")
        print(f"{form:16s} N={N:5d}: tile {t*1e3:7.1f} us/step   march {mr*1e3:7.1f} us/step   {'march' if mr < t else 'tile'}", flush=True)
