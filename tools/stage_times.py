"""Scratch: per-stage kernel times of the fused RK3 step (HIP events)."""
import sys, torch, statistics
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
FORM = __import__('os').environ.get("SWMHD_FORM", "VectorInvariant")      # SWMHD_FORM=Conservative: config 4's fields on the same grid
cfg = configs.config3_bickley() if FORM == "VectorInvariant" else configs.config4_two_gaussians()
g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
m = S.ShallowWaterModel(g, formulation=FORM, kernel=(sys.argv[2] if len(sys.argv) > 2 else "auto"), lorentz_forcing=(len(sys.argv) <= 3 or sys.argv[3] != "nolorentz"))
n1, n2 = m.names[:2]
m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
for _ in range(60): m.time_step(1e-4)   # (device clocks settle after ~30 ms of load)
m.tendency_events = []
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): m.time_step(1e-4)
e1.record(); torch.cuda.synchronize()
ms = [a.elapsed_time(b) * 1e3 for a, b, _ in m.tendency_events]
print("stage medians us:", [round(statistics.median(ms[s::3]), 1) for s in range(3)], " step ms:", round(e0.elapsed_time(e1) / 20, 3),
      " Mcell-steps/s:", round(N * N / (e0.elapsed_time(e1) / 20) / 1e3))
