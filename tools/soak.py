"""Soak: 1500 RK3 steps of the 4096^2 configurations (both formulations), twice -- mass conservation, finiteness and run-to-run
bitwise determinism of the fast kernels over a long run (a rare LDS or hazard race would show here)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
res = []
for form, cfgf in (("VectorInvariant", configs.config3_bickley), ("Conservative", configs.config4_two_gaussians)):
    cfg = cfgf()
    outs = []
    for rep in range(2):
        g = S.RectilinearGrid(size=(4096, 4096), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
        m = S.ShallowWaterModel(g, formulation=form)
        n1, n2 = m.names[:2]
        m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
        dt = 0.2 * min(g.dx, g.dy) / 4.2
        d0 = m.diagnostics()
        mass0 = m.solution["h"].data[g.interior].sum().item()
        m.time_steps(1500, dt)
        m.synchronize()
        d1 = m.diagnostics()
        mass1 = m.solution["h"].data[g.interior].sum().item()
        outs.append([f.data.clone() for f in m.fields])
        print(form, rep, "mass drift", abs(mass1 - mass0) / mass0, "E", d0["total_energy"], d1["total_energy"], "finite", all(torch.isfinite(f).all().item() for f in outs[-1]), flush=True)
        del m
    print(form, "bitwise deterministic:", all(torch.equal(a, b) for a, b in zip(*outs)), flush=True)
    del outs; torch.cuda.empty_cache()
