"""Scratch timing of the fused engine (not part of the contract): python scripts_time_model.py [N] [form]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
form = sys.argv[2] if len(sys.argv) > 2 else "VectorInvariant"
strict = len(sys.argv) > 3 and sys.argv[3] == "strict"
cfg = configs.config3_bickley() if form == "VectorInvariant" else configs.config4_two_gaussians()
g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
tile = len(sys.argv) > 3 and sys.argv[3] == "tile"
m = S.ShallowWaterModel(g, formulation=form, strict=strict, kernel=("tile" if tile else "march"))
if form == "VectorInvariant":
    m.set(u=cfg["u"], v=cfg["v"], h=lambda X, Y: cfg["h"](X, Y) + 0 * X, A=cfg["A"])
else:
    m.set(uh=cfg["u"], vh=cfg["v"], h=cfg["h"], A=cfg["A"])
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(m.calculate_tendencies)
print(f"{form} N={N} strict={strict}: tendency {t*1e3:.1f} us  -> {64*N*N/t/1e6:.0f} GB/s algorithmic, {N*N/t/1e6:.1f} Gcell/s")
dt = 1e-4
ts = timeit(lambda: m.time_step(dt), 5)
print(f"  RK3 step {ts:.3f} ms -> {N*N/ts/1e3:.0f} Mcell-steps/s")
print("  finite:", all(torch.isfinite(f.data).all().item() for f in m.fields))
