"""Timing of the Lorentz operator kernels: python tools/time_ops.py [Nx] [Ny] [f64|f32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs
Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ny = int(sys.argv[2]) if len(sys.argv) > 2 else Nx
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
bpc = 32 if dt == torch.float64 else 16
cfg = configs.config3_bickley()
g = S.RectilinearGrid(size=(Nx, Ny), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
A, h = S.Field(g, dtype=dt), S.Field(g, dtype=dt)
A.set(cfg["A"]); h.set(lambda X, Y: cfg["h"](X, Y) + 0 * X); A.fill_halo_regions(); h.fill_halo_regions()
out = (S.Field(g, dtype=dt), S.Field(g, dtype=dt))
def timeit(fn, n=50):
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
        print(f"{name:10s} {'tile ' if tile else 'march'} {Nx}x{Ny} {'f64' if bpc == 32 else 'f32'}: {t*1e3:7.1f} us  {bpc*Nx*Ny/t/1e6:7.0f} GB/s  ({bpc*Nx*Ny/t/1e6/80:.1f}% of 8 TB/s)", flush=True)
