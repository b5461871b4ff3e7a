import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import swmhd_amd as S
from oracle import oracle as O
from test_bounded_oracle import state, fill_all, LOC, G, F, P, B
from test_bounded_gpu import grid_for, FORM
form, lor, topo = 1, 1, (B, B)
Nx, Ny, dt = 48, 40, 2e-3
g = grid_for(S, Nx, Ny, topo, 0.1, 0.1)
q = fill_all(O, state(Nx, Ny, 9, form), Nx, Ny, topo, dx=g.dx, dy=g.dy)
m = S.ShallowWaterModel(g, G, F, formulation=FORM[form], lorentz_forcing=True, strict=True)
for f, a in zip(m._raw_fields, q):
    f.data.copy_(torch.from_numpy(a))
I = g.interior
# stage 1 by hand with the oracle
Gw = O.tendencies(*q, Nx, Ny, 3, 3, g.dx, g.dy, form, lor, G, F, nthreads=4, topo=topo)
m._stage_fused(dt, 0); torch.cuda.synchronize()
for n, w, gf in zip(m.names, Gw, m.Gn):
    print("G", n, np.abs(w[I] - gf.numpy()[I]).max())
new = [a.copy() for a in q]
for a, gg in zip(new, Gw):
    a[I] += dt * (8.0 / 15.0) * gg[I]
for n, w, f in zip(m.names, new, [m._alt[k] for k in m.names]):
    print("U1 interior", n, np.abs(w[I] - f.numpy()[I]).max())
m._state, m._alt = m._alt, m._state
m.Gn, m.Gm = m.Gm, m.Gn
m._fill_x(); torch.cuda.synchronize()
for a, loc in zip(new, LOC):
    O.fill_halo(a, Nx, Ny, 3, 3, topo=topo, face=loc, dx=g.dx, dy=g.dy)
for n, w, f in zip(m.names, new, m._raw_fields):
    d = np.abs(w - f.numpy())
    print("U1 after fill", n, d.max(), np.argwhere(d > 0)[:5].tolist())
# stage 2
Gw2 = O.tendencies(*new, Nx, Ny, 3, 3, g.dx, g.dy, form, lor, G, F, nthreads=4, topo=topo)
m._stage_fused(dt, 1); torch.cuda.synchronize()
for n, w, gf in zip(m.names, Gw2, m.Gn):
    d = np.abs(w[I] - gf.numpy()[I])
    print("G2", n, d.max(), np.argwhere(d > 0)[:5].tolist())
