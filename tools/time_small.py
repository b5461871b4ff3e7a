"""Step rate on the reference's own grid sizes (64^2, 128^2: SWMHD_example.jl:11, energy_plots/*/128x128_*), eager vs HIP graph."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swmhd_amd as S
from swmhd_amd import configs
import os
for N in [int(x) for x in os.environ.get('SIZES', '64,128,512,1024').split(',')]:
    g = S.RectilinearGrid(size=(N, N), x=(-5, 5), y=(-5, 5))
    for mode in ("eager",) if os.environ.get("SIZES") else ("eager", "graph"):
        form = os.environ.get("FORM", "VectorInvariant")
        m = S.ShallowWaterModel(g, formulation=form)
        n1, n2 = m.names[:2]
        m.set(**{n1: lambda X, Y: 5 * Y * np.exp(-(X**2 + Y**2)) * 0.01, n2: lambda X, Y: -5 * X * np.exp(-(X**2 + Y**2)) * 0.01,
              "h": lambda X, Y: np.ones_like(X), "A": configs.two_gaussians(0.1)})
        dt = 0.01 * 64 / N
        m.time_step(dt)
        if mode == "graph": m.capture_graph(dt)
        m.time_steps(20, dt); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 1000
        m.time_steps(n, dt); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{form[:4]} N={N:5d} {mode:5s}: {el/n*1e6:8.1f} us/step  {N*N*n/el/1e6:9.1f} Mcell-steps/s   finite={bool(torch.isfinite(m.solution['h'].data).all())}")
