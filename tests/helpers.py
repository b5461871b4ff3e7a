"""Shared input builders for the tests (numpy, host side)."""
import numpy as np


def coords(N, L, H, lo=None):
    """Cell-centre and face coordinates including H halo nodes, extended linearly (Oceananigans grid.xᶜᵃᵃ/xᶠᵃᵃ)."""
    lo = -L / 2 if lo is None else lo
    d = L / N
    k = np.arange(-H, N + H)
    return lo + (k + 0.5) * d, lo + k * d, d


def fill_halo_periodic(a, Nx, Ny, Hx, Hy):
    """numpy reference of the periodic halo fill (x first, then y over the padded width)."""
    a = a.copy()
    a[:, :Hx] = a[:, Nx:Nx + Hx]
    a[:, Nx + Hx:] = a[:, Hx:2 * Hx]
    a[:Hy, :] = a[Ny:Ny + Hy, :]
    a[Ny + Hy:, :] = a[Hy:2 * Hy, :]
    return a


def gaussian_case(N, H=3, L=10.0):
    """test_formulations.jl:12-18,156-159: A = exp(-(x^2+y^2)) evaluated from the (linearly extended) coordinate
    vectors, h = 1, on [-L/2, L/2]^2.  Returns A, h, d and the analytic Lorentz force at (fcc), (cfc)."""
    xc, xf, d = coords(N, L, H)
    X, Y = np.meshgrid(xc, xc)
    A = np.exp(-(X ** 2 + Y ** 2))
    h = np.ones_like(A)
    Xf, Yc = np.meshgrid(xf, xc)
    Xc, Yf = np.meshgrid(xc, xf)
    ex = -4 * Xf * np.exp(-2 * (Xf ** 2 + Yc ** 2))   # test_formulations.jl:14
    ey = -4 * Yf * np.exp(-2 * (Xc ** 2 + Yf ** 2))   # test_formulations.jl:15
    return A, h, d, ex, ey


def random_case(Nx, Ny, Hx, Hy, seed, dtype=np.float64, periodic=True):
    """Smooth-ish random A and a positive h, halos periodic-filled (what Oceananigans guarantees before a forcing call)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((Ny + 2 * Hy, Nx + 2 * Hx))
    h = 1.0 + 0.3 * rng.random((Ny + 2 * Hy, Nx + 2 * Hx))
    if periodic:
        A, h = fill_halo_periodic(A, Nx, Ny, Hx, Hy), fill_halo_periodic(h, Nx, Ny, Hx, Hy)
    return np.ascontiguousarray(A.astype(dtype)), np.ascontiguousarray(h.astype(dtype))


def two_gaussians(X, Y, amp=0.5):
    """divergence_formulation/divergence_sw_mhd.jl:33"""
    return amp * np.exp(-((X - 0.5) ** 2 + Y ** 2)) - amp * np.exp(-((X + 0.5) ** 2 + Y ** 2))


def interior(a, Nx, Ny, Hx, Hy):
    return a[Hy:Hy + Ny, Hx:Hx + Nx]
