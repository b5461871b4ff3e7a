"""Generates tests/golden/*.npz|json from the CPU oracle.

Provenance: RESTATEMENT-GENERATED.  The reference (Julia + un-vendored Oceananigans.jl) cannot be executed in
this image, and it stores no numeric outputs; these fixtures are outputs of oracle/ (the C restatement pinned
by the reference's analytic tests) on the reference's own test inputs.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402
import helpers as Hh  # noqa: E402


def main():
    # (1) BASELINE config 1 / test_formulations.jl input: 128x128, A = exp(-r^2), h = 1
    N, H = 128, 3
    A, h, d, ex, ey = Hh.gaussian_case(N, H)
    Jx, Jy = O.lorentz_jacobian(A, h, N, N, H, H, d, d)
    Dx, Dy = O.lorentz_divergence(A, h, N, N, H, H, d, d)
    np.savez_compressed(os.path.join(HERE, "gaussian_128.npz"), A=A, h=h, d=d, Jx=Jx, Jy=Jy, Dx=Dx, Dy=Dy)

    # (2) divergence_sw_mhd.jl:33 two-Gaussian A (amplitude 0.5), non-uniform h, 64x64 periodic
    N, H, L = 64, 3, 10.0
    xc, xf, d = Hh.coords(N, L, H)
    X, Y = np.meshgrid(xc, xc)
    A = Hh.fill_halo_periodic(Hh.two_gaussians(X, Y), N, N, H, H)
    h = Hh.fill_halo_periodic(1 + 0.1 * np.sin(2 * np.pi * X / L) * np.cos(2 * np.pi * Y / L), N, N, H, H)
    Jx, Jy = O.lorentz_jacobian(A, h, N, N, H, H, d, d)
    Dx, Dy = O.lorentz_divergence(A, h, N, N, H, H, d, d)
    np.savez_compressed(os.path.join(HERE, "two_gaussians_64.npz"), A=A, h=h, d=d, Jx=Jx, Jy=Jy, Dx=Dx, Dy=Dy)

    # (3) convergence table: max-norm error vs the analytic force of test_formulations.jl:14-15
    table = {}
    for N in (64, 128, 256, 512):
        A, h, d, ex, ey = Hh.gaussian_case(N, 3)
        Jx, Jy = O.lorentz_jacobian(A, h, N, N, 3, 3, d, d)
        Dx, Dy = O.lorentz_divergence(A, h, N, N, 3, 3, d, d)
        I = (slice(3, 3 + N), slice(3, 3 + N))
        table[str(N)] = {"jacobian_x": float(np.abs(Jx - ex)[I].max()), "jacobian_y": float(np.abs(Jy - ey)[I].max()),
                         "divergence_x": float(np.abs(Dx - ex)[I].max()), "divergence_y": float(np.abs(Dy - ey)[I].max())}
    with open(os.path.join(HERE, "convergence_table.json"), "w") as f:
        json.dump(table, f, indent=1)

    # (4) model path (A9 + forcing), 48x40 periodic, both formulations as the reference configures them
    #     (SWMHD_example.jl:21-33: VectorInvariant + Jacobian force; divergence_sw_mhd.jl:19-31: Conservative + divergence
    #     force): tendencies of the initial state and the state after two RK3 steps.  PARITY UNPINNED for A9 (the oracle's
    #     base RHS restates Oceananigans from its published scheme); the fixture guards the restatement against drift.
    import test_model_oracle as M
    Nx, Ny, H, dt = 48, 40, 3, 2e-3
    dx, dy = M.Lx / Nx, M.Ly / Ny
    xc, xf = (np.arange(-H, Nx + H) + 0.5) * dx, np.arange(-H, Nx + H) * dx
    yc, yf = (np.arange(-H, Ny + H) + 0.5) * dy, np.arange(-H, Ny + H) * dy
    cc, fc, cf = np.meshgrid(xc, yc), np.meshgrid(xf, yc), np.meshgrid(xc, yf)
    out = dict(Nx=Nx, Ny=Ny, H=H, dx=dx, dy=dy, dt=dt)
    for form, lor, tag in ((1, 1, "vi"), (0, 2, "cons")):
        h, A = M.hf(*cc), M.Af(*cc)
        q1, q2 = (M.uf(*fc), M.vf(*cf)) if form == 1 else (M.hf(*fc) * M.uf(*fc), M.hf(*cf) * M.vf(*cf))
        q = [O.fill_halo_periodic(np.ascontiguousarray(a), Nx, Ny, H, H) for a in (q1, q2, h, A)]
        G = O.tendencies(*q, Nx, Ny, H, H, dx, dy, form, lor)
        s = [a.copy() for a in q]
        for _ in range(2):
            O.time_step(*s, Nx, Ny, H, H, dx, dy, dt, form, lor)
        out.update({f"{tag}_q": np.stack(q), f"{tag}_G": np.stack(G), f"{tag}_after2": np.stack(s)})
    np.savez_compressed(os.path.join(HERE, "model_48x40.npz"), **out)


if __name__ == "__main__":
    main()
