"""How much do the per-launch HIP events of the roofline leg cost the step?  Same model, 3 ways of stepping."""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
N = 4096
cfg = configs.config3_bickley()
g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
m = S.ShallowWaterModel(g, formulation="VectorInvariant")
m.set(u=cfg["u"], v=cfg["v"], h=lambda X, Y: cfg["h"](X, Y) + 0 * X, A=cfg["A"])
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    m.tendency_events = []
    a = timed(lambda: m.time_step(1e-4))
    m.tendency_events = None
    b = timed(lambda: m.time_step(1e-4))
    c = timed(lambda: m.time_steps(1, 1e-4))
    d = timed(lambda: m.time_steps(20, 1e-4), 2) / 20
    print(f"step ms: python stages + events {a:.4f} | python stages, no events {b:.4f} | native driver per step {c:.4f} | native driver 20 steps/call {d:.4f}")
