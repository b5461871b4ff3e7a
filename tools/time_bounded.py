"""RK3 step time on a Bounded grid (LDS-tiled kernel + boundary-condition halo fill) beside the periodic step: python tools/time_bounded.py [N] [form]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swmhd_amd as S
from swmhd_amd import configs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
form = sys.argv[2] if len(sys.argv) > 2 else "VectorInvariant"
for topo in (("Periodic", "Periodic", "Flat"), ("Bounded", "Bounded", "Flat"), ("Periodic", "Bounded", "Flat")):
    g = S.RectilinearGrid(size=(N, N), x=(-5, 5), y=(-5, 5), topology=topo)
    m = S.ShallowWaterModel(g, formulation=form)
    n1, n2 = m.names[:2]
    m.set(**{n1: lambda X, Y: 0 * X, n2: lambda X, Y: 0 * X, "h": lambda X, Y: np.ones_like(X), "A": configs.two_gaussians(0.1)})
    dt = 1e-4
    for _ in range(20): m.time_step(dt)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
    for _ in range(n): m.time_step(dt)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
    print(f"{form[:4]} {N}^2 {topo[0][:4]}/{topo[1][:4]}: {el*1e3:8.3f} ms/step  {N*N/el/1e9:6.2f} Gcell-steps/s  finite={bool(torch.isfinite(m.solution['h'].data).all())}", flush=True)
