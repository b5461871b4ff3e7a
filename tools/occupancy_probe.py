"""Experiment: time the fused VI stage kernel with 1, 2, 3 workgroups per CU (slab heights chosen so that the grid is
255 / 510 / 765 workgroups of 92 rows each at Nx=4096).  Run with SWMHD_T_LY=92."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
cfg = configs.config3_bickley()
for ny in (1380, 2760, 4140):
    g = S.RectilinearGrid(size=(4096, ny), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
    m = S.ShallowWaterModel(g, 9.81, 1.0, kernel="march")
    m.set(u=cfg["u"], v=cfg["v"], h=lambda X, Y: cfg["h"](X, Y) + 0 * X, A=cfg["A"])
    for _ in range(60): m.time_step(1e-4)   # (device clocks settle after ~30 ms of load)
    m.tendency_events = []
    for _ in range(10): m.time_step(1e-4)
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b, _ in m.tendency_events])
    print(f"Ny={ny}: blocks/CU={ny//1380}  stage {ms.mean()*1e3:.1f} us  -> {4096*ny/ms.mean()/1e6:.2f} Gcell/s/stage")
