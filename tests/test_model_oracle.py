"""CPU tests of the base-RHS oracle (oracle/sw_rhs.inc).  PARITY UNPINNED against Oceananigans itself (absent, un-vendored,
un-pinned); what can be checked is checked: every tendency converges at 2nd order to the continuous shallow-water MHD
equations (spectral evaluation of the PDE right-hand side on the staggered grids), conservation, steady states and RK3."""
import numpy as np
import pytest

import helpers as Hh

Lx, Ly, G, F = 2 * np.pi, 2 * np.pi, 9.81, 1.0
kx, ky = 2 * np.pi / Lx, 2 * np.pi / Ly


def hf(x, y): return 1.0 + 0.2 * np.sin(kx * x) * np.cos(ky * y) + 0.1 * np.cos(2 * kx * x + 0.3)
def uf(x, y): return 0.5 * np.cos(kx * x + 0.4) * np.sin(2 * ky * y) + 0.3
def vf(x, y): return -0.4 * np.sin(2 * kx * x) * np.cos(ky * y + 0.2) - 0.2
def Af(x, y): return 0.3 * np.sin(kx * x + 0.1) * np.sin(ky * y - 0.5) + 0.1 * np.cos(2 * ky * y)


def _spec(Fld, L, axis):
    n = Fld.shape[axis]
    k = 2 * np.pi * np.fft.fftfreq(n, d=L / n)
    shp = [1, 1]; shp[axis] = n
    return np.real(np.fft.ifft(1j * k.reshape(shp) * np.fft.fft(Fld, axis=axis), axis=axis))


def pde_rhs(X, Y, form, lorentz):
    """Continuous RHS at the points (X, Y) (a uniform periodic grid), derivatives spectral."""
    dX = lambda q: _spec(q, Lx, 1)
    dY = lambda q: _spec(q, Ly, 0)
    h, u, v, A = hf(X, Y), uf(X, Y), vf(X, Y), Af(X, Y)
    Bx, By = -dY(A) / h, dX(A) / h
    Fx = Fy = 0 * h
    if lorentz == 1:    # J(A, B)/h           sw_mhd_jacobian_functions.jl:10-26
        Fx = (dX(A) * dY(Bx) - dY(A) * dX(Bx)) / h
        Fy = (dX(A) * dY(By) - dY(A) * dX(By)) / h
    elif lorentz == 2:  # div(h B B)          sw_mhd_divergence_functions.jl:162-170
        Fx = dX(h * Bx * Bx) + dY(h * By * Bx)
        Fy = dX(h * Bx * By) + dY(h * By * By)
    if form == 1:
        zeta, K = dX(v) - dY(u), 0.5 * (u * u + v * v)
        return (zeta * v - dX(K) - G * dX(h) + F * v + Fx, -zeta * u - dY(K) - G * dY(h) - F * u + Fy,
                -dX(h * u) - dY(h * v), -(u * dX(A) + v * dY(A)))
    U, V = h * u, h * v
    return (-dX(U * U / h) - dY(V * U / h) - dX(0.5 * G * h * h) + F * V + Fx,
            -dX(U * V / h) - dY(V * V / h) - dY(0.5 * G * h * h) - F * U + Fy,
            -dX(U) - dY(V), -(u * dX(A) + v * dY(A)))


def staggered_fields(N, form, H=3):
    dx, dy = Lx / N, Ly / N
    xc, xf = (np.arange(-H, N + H) + 0.5) * dx, np.arange(-H, N + H) * dx
    yc, yf = (np.arange(-H, N + H) + 0.5) * dy, np.arange(-H, N + H) * dy
    cc, fc, cf = np.meshgrid(xc, yc), np.meshgrid(xf, yc), np.meshgrid(xc, yf)
    h, A = hf(*cc), Af(*cc)
    if form == 1:
        q1, q2 = uf(*fc), vf(*cf)
    else:
        q1, q2 = hf(*fc) * uf(*fc), hf(*cf) * vf(*cf)
    return [np.ascontiguousarray(a) for a in (q1, q2, h, A)], (cc, fc, cf), dx, dy


@pytest.mark.parametrize("form,lorentz", [(1, 1), (0, 2), (1, 0), (0, 0)])
def test_tendencies_converge_to_the_pde(oracle, form, lorentz):
    errs = []
    for N in (32, 64, 128):
        q, (cc, fc, cf), dx, dy = staggered_fields(N, form)
        Gs = oracle.tendencies(*q, N, N, 3, 3, dx, dy, form, lorentz, G, F, nthreads=8)
        I = (slice(3, 3 + N), slice(3, 3 + N))
        r1 = pde_rhs(fc[0][I], fc[1][I], form, lorentz)[0]
        r2 = pde_rhs(cf[0][I], cf[1][I], form, lorentz)[1]
        rc = pde_rhs(cc[0][I], cc[1][I], form, lorentz)
        errs.append([np.abs(Gs[0][I] - r1).max() / np.abs(r1).max(), np.abs(Gs[1][I] - r2).max() / np.abs(r2).max(),
                     np.abs(Gs[2][I] - rc[2]).max() / np.abs(rc[2]).max(), np.abs(Gs[3][I] - rc[3]).max() / np.abs(rc[3]).max()])
    E = np.array(errs)
    order = np.log2(E[1] / E[2])
    assert np.all(order > 1.9), (E, order)
    assert np.all(E[2] < 6e-4), E


@pytest.mark.parametrize("form", [0, 1])
def test_mass_is_conserved_and_rest_state_is_steady(oracle, form):
    N = 48
    q, _, dx, dy = staggered_fields(N, form)
    I = (slice(3, 3 + N), slice(3, 3 + N))
    Gs = oracle.tendencies(*q, N, N, 3, 3, dx, dy, form, 0, G, F)
    assert abs(Gs[2][I].sum()) < 1e-11 * np.abs(Gs[2][I]).sum()      # flux form: sum of dh/dt telescopes to zero
    rest = [np.zeros_like(q[0]), np.zeros_like(q[0]), np.ones_like(q[0]), np.full_like(q[0], 0.7)]
    for lor in (0, 2 - form):
        Gs = oracle.tendencies(*rest, N, N, 3, 3, dx, dy, form, lor, G, F)
        assert all(np.array_equal(g_[I], np.zeros((N, N))) for g_ in Gs)


def test_rejects_mismatched_forcing(oracle):
    q, _, dx, dy = staggered_fields(16, 0)
    with pytest.raises(ValueError):
        oracle.tendencies(*q, 16, 16, 3, 3, dx, dy, 0, 1)   # Jacobian forcing acts on (u, v), not (uh, vh)


@pytest.mark.parametrize("form,lorentz,lo", [(1, 1, 7.0), (1, 0, 7.0), (0, 0, 3.5), (0, 2, 3.5)])
def test_rk3_step_is_third_order(oracle, form, lorentz, lo):
    """Halving dt must cut the one-step-group error by ~8 (RK3) -- checks gamma/zeta coefficients and G- bookkeeping.
    (Upwinding on the sign of the interpolated transport makes the conservative-form RHS only Lipschitz where uh
    changes sign, so the
    observed order is erratic there; it is held to > 3.5.  The vector-invariant runs show a clean factor 8.)"""
    N = 24
    q0, _, dx, dy = staggered_fields(N, form)
    q0 = [oracle.fill_halo_periodic(a, N, N, 3, 3) for a in q0]

    def run(nsteps, dt):
        q = [a.copy() for a in q0]
        for _ in range(nsteps):
            oracle.time_step(*q, N, N, 3, 3, dx, dy, dt, form, lorentz, G, F, nthreads=4)
        return q

    T = 0.02
    ref = run(16, T / 16)
    e1 = max(np.abs(a - b).max() for a, b in zip(run(1, T), ref))
    e2 = max(np.abs(a - b).max() for a, b in zip(run(2, T / 2), ref))
    assert lo < e1 / e2 < 10.5, (e1, e2)
    h = ref[2]
    assert np.array_equal(h[:3, :], h[N:N + 3, :]) and np.array_equal(h[:, :3], h[:, N:N + 3])   # halos periodic on exit
    assert abs(h[3:3 + N, 3:3 + N].sum() - q0[2][3:3 + N, 3:3 + N].sum()) < 1e-11 * N * N         # mass conserved


@pytest.mark.parametrize("form,lorentz", [(1, 1), (0, 0)])
def test_reflection_symmetry(oracle, form, lorentz):
    """Mirror the flow in x (x -> -x, u -> -u, Coriolis f -> -f): h- and A-tendencies mirror, the u-tendency mirrors with a sign flip.
    * With the textbook (mirrored) right-biased smoothness indicators (oracle switch rbeta_mirror = 1) this holds up to rounding: no term
      of the restatement has a directional bias.
    * With the indicators of the library version the reference ran (the default since round 3: right_biased_beta_0/2 are the left-biased
      end-point forms swapped, not reflected) the scheme is NOT mirror symmetric -- a property of that library
      version, visible in the reference's own energy plots -- by an amount that is small on smooth fields.
    (The reference's divergence-form forcing is left out: it upwinds the Maxwell stress on the sign of hB, and B is a pseudo-vector, so
    that scheme is not mirror symmetric by construction: 4e-4 relative here.  A property of sw_mhd_divergence_functions.jl:38-132.)"""
    N, H = 40, 3
    q, _, dx, dy = staggered_fields(N, form)
    q = [oracle.fill_halo_periodic(a, N, N, H, H) for a in q]
    I = (slice(H, H + N), slice(H, H + N))
    q1, q2, h, A = [a[I] for a in q]
    # centre fields: column i -> N-1-i ; x-face field (face i between i-1 and i): i -> (N - i) mod N, with a sign flip for u
    mc = lambda a: a[:, ::-1]
    mf = lambda a: np.roll(a[:, ::-1], 1, axis=1)

    def pad(a):
        out = np.zeros((N + 2 * H, N + 2 * H)); out[I] = a
        return oracle.fill_halo_periodic(out, N, N, H, H)

    qm = [pad(-mf(q1)), pad(mc(q2)), pad(mc(h)), pad(mc(A))]

    def asymmetry():
        G0 = oracle.tendencies(*q, N, N, H, H, dx, dy, form, lorentz, G, F)
        Gm = oracle.tendencies(*qm, N, N, H, H, dx, dy, form, lorentz, G, -F)
        want = [-mf(G0[0][I]), mc(G0[1][I]), mc(G0[2][I]), mc(G0[3][I])]
        return max(np.abs(w - g_[I]).max() / np.abs(w).max() for w, g_ in zip(want, Gm))

    try:
        oracle.set_variant(rbeta_mirror=1)
        assert asymmetry() <= 1e-11
    finally:
        oracle.set_variant(reset=1)
    a = asymmetry()
    assert 1e-9 < a < 2e-2, a


@pytest.mark.parametrize("tag,form,lor", [("vi", 1, 1), ("cons", 0, 2)])
def test_committed_model_fixture_reproduces(oracle, tag, form, lor):
    """tests/golden/model_48x40.npz (restatement-generated; tests/golden/make_golden.py): guards the oracle against drift."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_48x40.npz"))
    Nx, Ny, H, dx, dy, dt = int(z["Nx"]), int(z["Ny"]), int(z["H"]), float(z["dx"]), float(z["dy"]), float(z["dt"])
    q = [np.ascontiguousarray(a) for a in z[f"{tag}_q"]]
    Gt = oracle.tendencies(*q, Nx, Ny, H, H, dx, dy, form, lor, nthreads=4)
    for w, g in zip(z[f"{tag}_G"], Gt):
        assert np.array_equal(w, g)
    for _ in range(2):
        oracle.time_step(*q, Nx, Ny, H, H, dx, dy, dt, form, lor, nthreads=4)
    for w, s in zip(z[f"{tag}_after2"], q):
        assert np.array_equal(w, s)
