"""Stand-alone divergence-form Lorentz operator on a Bounded grid beside the periodic one: python tools/time_bounded_op.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for topo in (("Periodic", "Periodic", "Flat"), ("Bounded", "Bounded", "Flat")):
    g = S.RectilinearGrid(size=(N, N), x=(-5, 5), y=(-5, 5), topology=topo)
    A, h = S.Field(g), S.Field(g)
    A.set(configs.two_gaussians(0.5)); h.data.fill_(1.0)
    out = (S.Field(g), S.Field(g))
    fn = lambda: S.div_lorentz(g, {"A": A, "h": h}, out=out)
    for _ in range(200): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"divergence operator {N}^2 {topo[0]}/{topo[1]}: {e0.elapsed_time(e1)/50*1e3:7.1f} us", flush=True)
