"""GPU tests of the diagnostics kernel and of the soft physical pins the reference publishes as plots
(energy_plots/{jacobian,divergence}_formulation/128x128_*.png, definitions SWMHD_example.jl:67-77)."""
import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu
G = 9.81


def np_diagnostics(q1, q2, h, A, g, form, href=1.0):
    """numpy evaluation of the formulas documented in swmhd_amd/csrc/diagnostics.hip (test-side restatement): the reference's
    energy expressions with Oceananigans' operand-location rule -- KE: SWMHD_example.jl:74 (form 1) / divergence_sw_mhd.jl:71 (form 0,
    literally (1/2)(1/h)(uh^2 + vh^2)); ME: :75 / :72 with B_x = -dA/dy / h, B_y = dA/dx / h (:69-70 / :67-68)."""
    H, Nx, Ny = g.Hx, g.Nx, g.Ny
    S = lambda a, di, dj: a[H + dj:H + dj + Ny, H + di:H + di + Nx]
    hc = S(h, 0, 0)
    W = lambda di: S(q1, di, 0) ** 2 + 0.5 * (0.5 * (S(q2, di - 1, 0) ** 2 + S(q2, di, 0) ** 2) + 0.5 * (S(q2, di - 1, 1) ** 2 + S(q2, di, 1) ** 2))
    wbar = 0.5 * (W(0) + W(1))
    ke = 0.5 * (1.0 / hc) * wbar if form == 0 else 0.5 * hc * wbar
    BX = lambda di, dj: -((S(A, di, dj) - S(A, di, dj - 1)) / g.dy) / (0.5 * (S(h, di, dj - 1) + S(h, di, dj)))
    BY = lambda di, dj: ((S(A, di, dj) - S(A, di - 1, dj)) / g.dx) / (0.5 * (S(h, di - 1, dj) + S(h, di, dj)))
    Z = lambda dj: BX(0, dj) ** 2 + 0.5 * (0.5 * (BY(0, dj - 1) ** 2 + BY(1, dj - 1) ** 2) + 0.5 * (BY(0, dj) ** 2 + BY(1, dj) ** 2))
    me = 0.5 * hc * (0.5 * (Z(0) + Z(1)))
    pe = 0.5 * G * (hc - href) ** 2
    c = g.dx * g.dy
    uw, vs = S(q1, 0, 0), S(q2, 0, 0)
    if form == 0:      # u = uh / h at uh's faces (divergence_sw_mhd.jl:45-47)
        uw, vs = uw / (0.5 * (S(h, -1, 0) + hc)), vs / (0.5 * (S(h, 0, -1) + hc))
    return dict(kinetic_energy=ke.sum() * c, magnetic_energy=me.sum() * c, potential_energy=pe.sum() * c,
                max_abs_u=np.abs(uw).max(), max_abs_v=np.abs(vs).max(), max_abs_A=np.abs(S(A, 0, 0)).max(), min_h=hc.min())


@pytest.mark.parametrize("form", ["VectorInvariant", "Conservative"])
def test_diagnostics_match_numpy(swmhd, form):
    Nx, Ny = 200, 75
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, 3.0), y=(0, 2.0))
    m = swmhd.ShallowWaterModel(g, G, 1.0, formulation=form)
    rng = np.random.default_rng(4)
    raw = [rng.standard_normal(g.parent_shape), rng.standard_normal(g.parent_shape), 1 + rng.random(g.parent_shape), rng.standard_normal(g.parent_shape)]
    for f, a in zip(m.fields, raw):
        f.data.copy_(torch.from_numpy(a))
    m.update_state()
    d = m.diagnostics()
    want = np_diagnostics(*[f.numpy() for f in m.fields], g, 1 if form == "VectorInvariant" else 0)
    for k, w in want.items():
        assert abs(d[k] - w) <= 1e-12 * max(abs(w), 1.0), (k, d[k], w)
    assert abs(d["total_energy"] - (want["kinetic_energy"] + want["magnetic_energy"] + want["potential_energy"])) < 1e-9


def two_gaussians(amp):
    return lambda X, Y: amp * np.exp(-((X - 0.5) ** 2 + Y ** 2)) - amp * np.exp(-((X + 0.5) ** 2 + Y ** 2))


@pytest.mark.parametrize("amp,me_plot,me_exact", [(0.1, 0.0218, 0.02189), (0.5, 0.545, 0.5472)])
def test_initial_magnetic_energy_of_the_reference_cases(swmhd, amp, me_plot, me_exact):
    """128x128 two_Gaussians_{low,high}_B (SWMHD_example.jl:37 / divergence_sw_mhd.jl:33): the committed plots start at
    ME ~ 0.0218 / 0.545; analytically 1/2 int |grad A|^2 = 0.02189 / 0.5472."""
    for form in ("VectorInvariant", "Conservative"):
        g = swmhd.RectilinearGrid(size=(128, 128), x=(-5, 5), y=(-5, 5))
        m = swmhd.ShallowWaterModel(g, G, 1.0, formulation=form)
        m.set(h=lambda X, Y: np.ones_like(X), A=two_gaussians(amp))
        d = m.diagnostics()
        assert abs(d["magnetic_energy"] - me_exact) < 3e-3 * me_exact     # 2nd-order discretisation error at 128^2 (~2e-3)
        assert abs(d["magnetic_energy"] - me_plot) < 6e-3 * me_plot       # value read off the plot
        assert d["kinetic_energy"] == 0.0 and d["potential_energy"] == 0.0 and d["min_h"] == 1.0
        assert abs(d["max_abs_A"] - np.abs(two_gaussians(amp)(*g.nodes(("Center", "Center")))).max()) < 1e-15


def test_initial_energies_low_B_low_U(swmhd):
    """128x128 low_B_low_U: |B| = 0.05 uniform (A = -0.05 y, divergence_sw_mhd.jl:34), u = y e^{-r^2}, v = -x e^{-r^2}
    (:36-37): plots start at KE ~ 0.393 (= pi/8) and ME = 0.125."""
    g = swmhd.RectilinearGrid(size=(128, 128), x=(-5, 5), y=(-5, 5))
    m = swmhd.ShallowWaterModel(g, G, 1.0, formulation="VectorInvariant")
    f = m.solution
    f["u"].set(lambda X, Y: Y * np.exp(-(X ** 2 + Y ** 2))); f["v"].set(lambda X, Y: -X * np.exp(-(X ** 2 + Y ** 2)))
    f["h"].set(lambda X, Y: np.ones_like(X)); f["A"].set(lambda X, Y: -0.05 * Y)     # halos: linear extension (not wrapped)
    d = m.diagnostics()
    assert abs(d["kinetic_energy"] - np.pi / 8) < 2e-3 and abs(d["kinetic_energy"] - 0.393) < 2e-3
    assert abs(d["magnetic_energy"] - 0.125) < 1e-12


# (The dynamic pins -- KE / ME / PE / |E - E0| time series of all twelve plotted runs -- are tests/test_reference_plots.py.)


def test_measurement_probes_return_sane_numbers(swmhd):
    """swmhd_probe_fp64_issue / swmhd_probe_copy (bench.py's `box` block): an MI355X issues one fp64 wave-instruction per SIMD every
    4 cycles of a 1.4-2.4 GHz clock and copies at 3-8 TB/s; bad arguments are refused."""
    import ctypes
    L = swmhd._lib.lib()
    scratch = torch.zeros(8, dtype=torch.float64, device="cuda")
    ns = ctypes.c_float(0)
    assert L.swmhd_probe_fp64_issue(scratch.data_ptr(), ctypes.byref(ns), None) == 0
    assert 1.6 < ns.value < 3.0, ns.value
    src = torch.ones(1 << 25, dtype=torch.float64, device="cuda")       # 256 MiB
    dst = torch.zeros_like(src)
    gb = ctypes.c_float(0)
    assert L.swmhd_probe_copy(dst.data_ptr(), src.data_ptr(), src.numel() * 8, 5, ctypes.byref(gb), None) == 0
    assert 3000 < gb.value < 9000 and torch.equal(dst, src), gb.value
    assert L.swmhd_probe_copy(dst.data_ptr(), src.data_ptr(), 100, 5, ctypes.byref(gb), None) == 1          # not a multiple of 16 / too small
    assert L.swmhd_probe_fp64_issue(None, ctypes.byref(ns), None) == 1
