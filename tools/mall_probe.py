#!/usr/bin/env python3
"""Does a fused stage run faster when its inputs were written a moment ago (Infinity Cache resident) than when they come from HBM?
Band experiment with the existing kernels: stage 1 over a band of rows writes (U1, G0); stage 2 over the same band reads them.
Compared with stage 2 over a band whose inputs were written long ago (a far-away band, evicted by the traffic in between).
    python tools/mall_probe.py [Nx] [Ny] [band]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs, _lib
Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ny = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
band = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cfg = configs.config3_bickley()
g = S.RectilinearGrid(size=(Nx, Ny), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
m = S.ShallowWaterModel(g, formulation="VectorInvariant")
m.set(u=cfg["u"], v=cfg["v"], h=lambda X, Y: cfg["h"](X, Y) + 0 * X, A=cfg["A"])
dt = 1e-5
for _ in range(30): m.time_step(dt)
torch.cuda.synchronize()
nb = Ny // band
def ev(): return _lib.TimingEvent()
def stage(st, rows): m._stage_fused(dt, st, rows, 0)
# (a) same band: s1(band k) then s2(band k) -- s2's inputs (state written by s1 into _alt, G0) are fresh
# (b) far band:  s1(band k) then s2(band k + nb/2) -- inputs written half a grid ago in the previous sweep
res = {}
for tag, shift in (("same_band", 0), ("far_band", nb // 2)):
    t1 = t2 = 0.0
    for rep in range(3):
        for k in range(nb):
            r1 = (k * band, (k + 1) * band)
            kk = (k + shift) % nb
            r2 = (kk * band + 3, (kk + 1) * band - 3)
            a, b, c = ev(), ev(), ev()
            a.record(); stage(0, r1); b.record()
            # stage 2 reads the buffers stage 1 wrote: swap roles by hand for this launch
            m._state, m._alt = m._alt, m._state; m.Gn, m.Gm = m.Gm, m.Gn
            stage(1, r2); c.record()
            m._state, m._alt = m._alt, m._state; m.Gn, m.Gm = m.Gm, m.Gn
            torch.cuda.synchronize()
            if rep > 0:
                t1 += a.elapsed_time(b); t2 += b.elapsed_time(c)
    n = 2 * nb
    res[tag] = (t1 / n * 1e3, t2 / n * 1e3)
    print(f"{tag:10s} band {band} rows x {Nx}: stage1 {t1/n*1e3:7.1f} us   stage2 {t2/n*1e3:7.1f} us   ({128*Nx*(band-6)/(t2/n*1e-3)/1e9:.0f} GB/s on 128 B/cell)", flush=True)
