"""CPU tests (no GPU): pin the oracle against the reference's own known answers.

The reference holds no numeric fixtures; its tests are analytic:
  test_formulations.jl:12-15   A = exp(-(x^2+y^2)), h = 1  ->  F = (-4x, -4y) exp(-2 r^2) at (fcc), (cfc)
  test_formulations.jl:205-210 both discretisations converge at 2nd order in max-norm, N = 64..512
"""
import json
import os

import numpy as np
import pytest

import helpers as Hh

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# max-norm errors produced by an INDEPENDENT numpy restatement written during the survey
SURVEY_TABLE = {
    64: (6.554741068160697e-2, 8.998312934496977e-2, 2.4828e-2),
    128: (1.7106036417579684e-2, 2.297462105153114e-2, 6.1484e-3),
    256: (4.323230380995691e-3, 5.778557424003683e-3, 1.5028e-3),
    512: (1.0837563102470416e-3, 1.4478868317953086e-3, 3.7314e-4),
}


def _errs(oracle, N):
    A, h, d, ex, ey = Hh.gaussian_case(N, 3)
    Jx, Jy = oracle.lorentz_jacobian(A, h, N, N, 3, 3, d, d)
    Dx, Dy = oracle.lorentz_divergence(A, h, N, N, 3, 3, d, d)
    I = (slice(3, 3 + N), slice(3, 3 + N))
    return (np.abs(Jx - ex)[I].max(), np.abs(Jy - ey)[I].max(), np.abs(Dx - ex)[I].max(), np.abs(Dy - ey)[I].max(),
            np.abs(Jx - Dx)[I].max())


def test_analytic_lorentz_force_second_order(oracle):
    """test_formulations.jl:188-189,205-210: error vs the analytic force shrinks at 2nd order for both forms."""
    Ns = [64, 128, 256, 512]
    E = np.array([_errs(oracle, N)[:4] for N in Ns])
    for col in range(4):
        slope = -np.polyfit(np.log10(Ns), np.log10(E[:, col]), 1)[0]
        assert 1.9 < slope < 2.1, slope
    assert E[-1].max() < 1.5e-3


def test_matches_survey_independent_restatement(oracle):
    """Same numbers as the survey's scratch numpy restatement (different code, same semantics) to 1e-13 relative."""
    for N, (ej, ed, ejd) in SURVEY_TABLE.items():
        jx, jy, dx, dy, jd = _errs(oracle, N)
        assert abs(jx - ej) <= 1e-13 * ej and abs(jy - ej) <= 1e-13 * ej
        assert abs(dx - ed) <= 1e-13 * ed and abs(dy - ed) <= 1e-11 * ed
        assert abs(jd - ejd) <= 1e-4 * ejd   # table quotes 5 digits


def test_committed_convergence_table(oracle):
    with open(os.path.join(GOLD, "convergence_table.json")) as f:
        table = json.load(f)
    for N in (64, 128, 256):
        jx, jy, dx, dy, _ = _errs(oracle, N)
        t = table[str(N)]
        assert (jx, jy, dx, dy) == (t["jacobian_x"], t["jacobian_y"], t["divergence_x"], t["divergence_y"])


@pytest.mark.parametrize("name", ["gaussian_128", "two_gaussians_64"])
def test_golden_fixtures_reproduce(oracle, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    A, h, d = z["A"], z["h"], float(z["d"])
    N = A.shape[0] - 6
    Jx, Jy = oracle.lorentz_jacobian(A, h, N, N, 3, 3, d, d)
    Dx, Dy = oracle.lorentz_divergence(A, h, N, N, 3, 3, d, d)
    for got, want in ((Jx, z["Jx"]), (Jy, z["Jy"]), (Dx, z["Dx"]), (Dy, z["Dy"])):
        assert np.array_equal(got, want)


def test_B_field_of_gaussian(oracle):
    """MHD_visualize.jl:8-20 style check: centre Bx = -dA/dy / h, By = dA/dx / h, 2nd order in max-norm."""
    errs = []
    for N in (32, 64):
        A, h, d, _, _ = Hh.gaussian_case(N, 3)
        h = h * 2.0
        xc, _, _ = Hh.coords(N, 10.0, 3)
        ebx = eby = 0.0
        for j in range(1, N + 1):
            for i in range(1, N + 1):
                x, y = xc[i - 1 + 3], xc[j - 1 + 3]
                a = np.exp(-(x * x + y * y))
                ebx = max(ebx, abs(oracle.probe("jac_Bx", A, h, i, j, N, N, 3, 3, d, d) - (2 * y * a) / 2.0))
                eby = max(eby, abs(oracle.probe("jac_By", A, h, i, j, N, N, 3, 3, d, d) - (-2 * x * a) / 2.0))
        errs.append((ebx, eby))
    assert errs[1][0] < errs[0][0] / 3.5 and errs[1][1] < errs[0][1] / 3.5
    assert abs(errs[0][0] - errs[0][1]) < 1e-12   # x/y symmetry of the Gaussian


def _footprint(oracle, fn, field, seed=7):
    """Which input offsets (di,dj) can change output (i,j)?  Perturbation test of SURVEY.md 8(a)."""
    N, H = 16, 4
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N + 2 * H, N + 2 * H))
    h = 1 + rng.random((N + 2 * H, N + 2 * H))
    i = j = 8
    d = 0.37
    base = oracle.probe(fn, A, h, i, j, N, N, H, H, d, d)
    hits = set()
    for dj in range(-4, 5):
        for di in range(-4, 5):
            A2, h2 = A.copy(), h.copy()
            (A2 if field == "A" else h2)[j - 1 + H + dj, i - 1 + H + di] += 0.5
            if oracle.probe(fn, A2, h2, i, j, N, N, H, H, d, d) != base:
                hits.add((di, dj))
    return hits


def test_footprints(oracle):
    fa = _footprint(oracle, "lorentz_force_func_x", "A")
    assert fa == {(di, dj) for di in (-1, 0) for dj in range(-2, 3)}                       # 10 pts
    assert _footprint(oracle, "lorentz_force_func_x", "h") == {(di, dj) for di in (-1, 0) for dj in (-1, 0, 1)}
    fa = _footprint(oracle, "lorentz_force_func_y", "A")
    assert fa == {(di, dj) for di in range(-2, 3) for dj in (-1, 0)}
    assert _footprint(oracle, "lorentz_force_func_y", "h") == {(di, dj) for di in (-1, 0, 1) for dj in (-1, 0)}
    # divergence form: upwinding makes the exact set data dependent -> union over seeds, compare bounding boxes
    # with SURVEY.md 8(a) row A8 (Fx: A i-3..i+2 x j-3..j+3, h i-3..i+2 x j-2..j+2; Fy: transposed)
    def bbox(fn, fld):
        u = set()
        for seed in range(8):
            u |= _footprint(oracle, fn, fld, seed)
        xs, ys = [a for a, _ in u], [b for _, b in u]
        return (min(xs), max(xs), min(ys), max(ys)), len(u)
    assert bbox("div_lorentz_x", "A")[0] == (-3, 2, -3, 3)
    assert bbox("div_lorentz_x", "h") == ((-3, 2, -2, 2), 14)
    assert bbox("div_lorentz_y", "A")[0] == (-3, 3, -3, 2)
    assert bbox("div_lorentz_y", "h") == ((-2, 2, -3, 2), 14)


def test_periodic_shift_equivariance(oracle):
    """Shifting periodic inputs by (sx, sy) cells shifts the outputs identically (bitwise)."""
    Nx, Ny, H = 24, 20, 3
    A, h = Hh.random_case(Nx, Ny, H, H, 3)
    d = (0.4, 0.3)
    J = oracle.lorentz_jacobian(A, h, Nx, Ny, H, H, *d)
    D = oracle.lorentz_divergence(A, h, Nx, Ny, H, H, *d)
    sx, sy = 5, 7

    def shift(a):
        core = np.roll(Hh.interior(a, Nx, Ny, H, H), (sy, sx), axis=(0, 1))
        out = np.zeros_like(a)
        out[H:H + Ny, H:H + Nx] = core
        return Hh.fill_halo_periodic(out, Nx, Ny, H, H)

    J2 = oracle.lorentz_jacobian(shift(A), shift(h), Nx, Ny, H, H, *d)
    D2 = oracle.lorentz_divergence(shift(A), shift(h), Nx, Ny, H, H, *d)
    for a, b in zip(J + D, J2 + D2):
        assert np.array_equal(np.roll(Hh.interior(a, Nx, Ny, H, H), (sy, sx), axis=(0, 1)), Hh.interior(b, Nx, Ny, H, H))


def test_quadratic_scaling_exact(oracle):
    """The Lorentz force is quadratic in A: scaling A by 2 scales F by exactly 4 (power-of-two => bitwise)."""
    Nx, Ny, H = 17, 9, 3
    A, h = Hh.random_case(Nx, Ny, H, H, 11)
    for fn in (oracle.lorentz_jacobian, oracle.lorentz_divergence):
        F1 = fn(A, h, Nx, Ny, H, H, 0.2, 0.25)
        F2 = fn(2 * A, h, Nx, Ny, H, H, 0.2, 0.25)
        assert np.array_equal(4 * F1[0], F2[0]) and np.array_equal(4 * F1[1], F2[1])


def test_bounded_branches_differ_only_near_walls(oracle):
    """sw_mhd_divergence_functions.jl:42-53,66-77,90-101,114-125: the Bounded branches only touch fluxes within
    two cells of a wall; far from walls Bounded == Periodic."""
    Nx, Ny, H = 20, 18, 3
    A, h = Hh.random_case(Nx, Ny, H, H, 5)
    P = oracle.lorentz_divergence(A, h, Nx, Ny, H, H, 0.5, 0.5)
    B = oracle.lorentz_divergence(A, h, Nx, Ny, H, H, 0.5, 0.5, topo=(oracle.BOUNDED, oracle.BOUNDED))
    for p, b in zip(P, B):
        pi, bi = Hh.interior(p, Nx, Ny, H, H), Hh.interior(b, Nx, Ny, H, H)
        assert np.array_equal(pi[3:-3, 3:-3], bi[3:-3, 3:-3])
        assert not np.array_equal(pi, bi)
    Bx = oracle.lorentz_divergence(A, h, Nx, Ny, H, H, 0.5, 0.5, topo=(oracle.BOUNDED, oracle.PERIODIC))
    assert np.array_equal(Hh.interior(Bx[0], Nx, Ny, H, H)[:, 3:-3], Hh.interior(P[0], Nx, Ny, H, H)[:, 3:-3])


def test_f32_oracle_tracks_f64(oracle):
    Nx, Ny, H = 32, 32, 3
    A, h = Hh.random_case(Nx, Ny, H, H, 2)
    for fn in (oracle.lorentz_jacobian, oracle.lorentz_divergence):
        F64 = fn(A, h, Nx, Ny, H, H, 0.3, 0.3)
        F32 = fn(A.astype(np.float32), h.astype(np.float32), Nx, Ny, H, H, 0.3, 0.3)
        for a, b in zip(F64, F32):
            assert b.dtype == np.float32
            assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max()   # SURVEY.md 8(c) fp32 tolerance


def test_halo_requirements(oracle):
    A, h = Hh.random_case(8, 8, 2, 2, 1)
    oracle.lorentz_jacobian(A, h, 8, 8, 2, 2, 1.0, 1.0)
    with pytest.raises(ValueError):
        oracle.lorentz_divergence(A, h, 8, 8, 2, 2, 1.0, 1.0)


def test_discrete_jacobian_against_the_analytic_gaussian(oracle):
    """test_jacobian.jl:16-24,45-63: the discrete Jacobian stencils applied to (A, dA/dx) at (xf, yc) and (A, dA/dy) at (xc, yf) against
    the closed forms for A = exp(-x^2 - y^2) on [-5,5]^2, N = 50 .. 400; the script fits the order of the max-norm error (2)."""
    errs = []
    for N in (50, 100, 200, 400):
        xc, xf, d = Hh.coords(N, 10.0, 3)
        X, Y = np.meshgrid(xc, xc)
        A = np.ascontiguousarray(np.exp(-X ** 2 - Y ** 2))          # A(i,j,k,grid,x,y) evaluated from the extended coordinate vectors
        h = np.ones_like(A)
        ex = ey = 0.0
        idx = range(1, N + 1, max(1, N // 50))                     # the probe is per point: sample the grid, always incl. the centre band
        for j in idx:
            for i in idx:
                x, y = xf[i - 1 + 3], xc[j - 1 + 3]
                e = np.exp(-x * x - y * y)
                exact = (-2 * x * e) * (4 * x * y * e) - ((4 * x * x - 2) * e) * (-2 * y * e)       # dA_x dA_xy - dA_xx dA_y  (:57-58)
                ex = max(ex, abs(oracle.probe("test_jacobian_x", A, h, i, j, N, N, 3, 3, d, d) - exact))
                x, y = xc[i - 1 + 3], xf[j - 1 + 3]
                e = np.exp(-x * x - y * y)
                exact = (-2 * x * e) * ((4 * y * y - 2) * e) - (4 * x * y * e) * (-2 * y * e)       # dA_x dA_yy - dA_yx dA_y  (:59-60)
                ey = max(ey, abs(oracle.probe("test_jacobian_y", A, h, i, j, N, N, 3, 3, d, d) - exact))
        errs.append((ex, ey))
    E = np.array(errs)
    order = np.log2(E[:-1] / E[1:])
    assert np.all(order > 1.8) and np.all(order < 2.3), (E, order)
    assert np.all(E[-1] < 2e-3), E
