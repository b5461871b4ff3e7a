"""Scratch timing of the Lorentz operator kernels: python scripts_time_ops.py [N]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = configs.config3_bickley()
g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
A, h = S.Field(g), S.Field(g)
A.set(cfg["A"]); h.set(lambda X, Y: cfg["h"](X, Y) + 0 * X); A.fill_halo_regions(); h.fill_halo_regions()
out = (S.Field(g), S.Field(g))
def timeit(fn, n=30):
    for _ in range(300): fn()   # (device clocks settle after ~30 ms of load)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, fn in (("jacobian", S.lorentz_force_func), ("divergence", S.div_lorentz)):
    for tile in (False, True):
        t = timeit(lambda: fn(g, {"A": A, "h": h}, out=out, kernel=("tile" if tile else "march")))
        print(f"{name:10s} {'tile ' if tile else 'march'} N={N}: {t*1e3:7.1f} us  {32*N*N/t/1e6:7.0f} GB/s  ({32*N*N/t/1e6/80:.1f}% of 8 TB/s)")
