#!/usr/bin/env python3
"""Go / no-go measurement for cross-stage overlap on mid-size grids, stream version: every RK3 stage is cut into B
y-bands, each its own launch on one of S streams; band b of stage k+1 waits (events) only for bands b-1, b, b+1 of stage k, so the
drain of one stage and the prologue bursts of the next can overlap instead of meeting at a full barrier.  Same kernels, same
arithmetic (the row ranges of swmhd_tendencies_rk3), results checked against the plain step.
    python tools/band_overlap_probe.py [Nx] [Ny] [formulation]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Ny = int(sys.argv[2]) if len(sys.argv) > 2 else Nx
form = sys.argv[3] if len(sys.argv) > 3 else "VectorInvariant"
cfg = configs.config3_bickley() if form == "VectorInvariant" else configs.config2_uniform_bx()
y0, y1 = cfg["domain"]["y"]; yc, Ly = 0.5 * (y0 + y1), (y1 - y0)
ydom = (yc - Ly * Ny / 4096 / 2, yc + Ly * Ny / 4096 / 2) if form == "VectorInvariant" else (y0, y1)
g = S.RectilinearGrid(size=(Nx, Ny), x=cfg["domain"]["x"], y=ydom)
dt = 0.2 * min(g.dx, g.dy) / 4.2


def model():
    m = S.ShallowWaterModel(g, formulation=form)
    n1, n2 = m.names[:2]
    m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
    return m


def banded_step(m, bands, streams, done):
    """one RK3 step; done[b] = event of band b's launch of the previous stage (None: nothing pending)"""
    B = len(bands)
    for stage in range(3):
        new = [None] * B
        for b, (j0, j1) in enumerate(bands):
            st = streams[b % len(streams)]
            for nb in ((b - 1) % B, b, (b + 1) % B):
                if done[nb] is not None:
                    st.wait_event(done[nb])
            with torch.cuda.stream(st):
                m._stage_fused(dt, stage, (j0, j1))
                ev = torch.cuda.Event(); ev.record(st)
            new[b] = ev
        done[:] = new
        m._state, m._alt = m._alt, m._state
        m.Gn, m.Gm = m.Gm, m.Gn
        m._halo_stale = True


def timeit(stepper, n=200, spin=100):
    for _ in range(spin): stepper()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): stepper()
    torch.cuda.synchronize(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = model()
t_plain = timeit(lambda: ref.time_step(dt))
ref.capture_graph(dt)
t_graph = timeit(lambda: ref.time_steps(2, dt), n=100) / 2
print(f"{form} {Nx}x{Ny}: plain step {t_plain * 1e3:.1f} us ({Nx * Ny / t_plain / 1e6:.2f} Gcell-steps/s), HIP-graph replay {t_graph * 1e3:.1f} us", flush=True)
for B, Sn in ((2, 2), (4, 2), (4, 4), (8, 4), (16, 4)):
    if Ny // B < 8:
        continue
    m = model()
    bands = [(b * Ny // B, (b + 1) * Ny // B) for b in range(B)]
    streams = [torch.cuda.Stream() for _ in range(Sn)]
    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream())
    done = [None] * B
    chk = model()
    for _ in range(4):
        banded_step(m, bands, streams, done); chk.time_step(dt)
    torch.cuda.synchronize()
    err = max(((a.data[g.interior] - b.data[g.interior]).abs().max() / b.data[g.interior].abs().max()).item() for a, b in zip(m._raw_fields, chk._raw_fields))
    # (Capturing the banded step as ONE multi-stream HIP graph, to take the host out of the picture, crashed inside the capture on this
    #  ROCm/PyTorch pair -- a host-side segmentation fault, 2026-10 -- so the eager numbers below carry the host's per-launch cost:
    #  about 20 us per band launch, which is the whole story at these sizes.)
    t = timeit(lambda: banded_step(m, bands, streams, done))
    print(f"  {B:2d} bands on {Sn} streams: {t * 1e3:7.1f} us per step ({100 * (t / t_plain - 1):+.0f} % vs plain)   max rel. diff to the plain step after 4 steps: {err:.1e}", flush=True)
