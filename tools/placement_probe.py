#!/usr/bin/env python3
"""Does the RELATIVE placement of the arrays in HBM matter?  bench.py's operator leg and tools/run_configs.py time the same Lorentz-operator
launch on the same box 15 % apart (125 vs 109 us); what differs is where the allocator put A, h, Fx, Fy.  Here the four arrays
(and, for the fused step, the sixteen) are carved out of ONE allocation at base + k x (field bytes rounded up to 2 MiB + stagger)
for a list of staggers, and each layout is timed in the same process.
    python tools/placement_probe.py [N] [--step]"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swmhd_amd as S
from swmhd_amd import configs

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
cfg = configs.config3_bickley()
g = S.RectilinearGrid(size=(N, N), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
n = g.parent_shape[0] * g.parent_shape[1]
MiB = 1 << 20
base_elems = ((n * 8 + 2 * MiB - 1) // (2 * MiB)) * (2 * MiB) // 8
STAGGERS = [0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, MiB, MiB + 65536 + 4096 + 256, 3 * MiB // 2 + 4096]


def timeit(fn, n_spin=150, K=40):
    for _ in range(n_spin): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


def carve(big, k, stagger):
    off = k * (base_elems + stagger // 8)
    return big[off:off + n].view(g.parent_shape)


X, Y = g.nodes(("Center", "Center"))
A0 = torch.from_numpy(cfg["A"](X, Y) + 0 * X).cuda()
h0 = torch.from_numpy(cfg["h"](X, Y) + 0 * X).cuda()
big = torch.zeros(4 * (base_elems + max(STAGGERS) // 8) + 16, dtype=torch.float64, device="cuda")
print(f"grid {N}^2, field {n * 8} B, slot {base_elems * 8} B (2 MiB multiple); base address % 2 MiB = {big.data_ptr() % (2 * MiB)}")
for st in STAGGERS:
    f = [S.Field(g, data=carve(big, k, st)) for k in range(4)]
    f[0].data.copy_(A0); f[1].data.copy_(h0)
    fld = {"A": f[0], "h": f[1]}
    t = [timeit(lambda: S.lorentz_force_func(g, fld, out=(f[2], f[3]))) for _ in range(3)]
    print(f"stagger {st:>9d} B : Lorentz operator {statistics.median(t):7.1f} us  ({32 * N * N / statistics.median(t) / 1e6:.2f} TB/s)  rounds {[round(x, 1) for x in t]}", flush=True)
# separately allocated tensors, as the model does
f = [S.Field(g) for _ in range(4)]
f[0].data.copy_(A0); f[1].data.copy_(h0)
fld = {"A": f[0], "h": f[1]}
t = timeit(lambda: S.lorentz_force_func(g, fld, out=(f[2], f[3])))
print("torch-allocated fields  : %7.1f us   pointer deltas (MiB): %s" % (t, [round((f[k + 1].ptr - f[k].ptr) / MiB, 3) for k in range(3)]))
