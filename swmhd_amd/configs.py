"""The BASELINE.json configurations as concrete initial conditions.

All functions take numpy coordinate arrays (X, Y) and return numpy arrays; parameters cite the reference script they
come from.  Synthetic, deterministic, no files."""
import numpy as np

G = 9.81   # gravitational_acceleration, SWMHD_example.jl:27
F = 1.0    # FPlane(f=1), SWMHD_example.jl:28


def two_gaussians(amp):
    """divergence_sw_mhd.jl:33 (amp = 0.5) / SWMHD_example.jl:37 (amp = 0.1)"""
    return lambda X, Y: amp * np.exp(-((X - 0.5) ** 2 + Y ** 2)) - amp * np.exp(-((X + 0.5) ** 2 + Y ** 2))


def config3_bickley(Lx=2 * np.pi, Ly=20.0):
    """4096^2 Jacobian formulation, Bickley-jet-style h/u (test_example.jl:63-70: Lx = 2pi, Ly = 20, U = 1,
    dη = f U / g), A = y-periodic current sheets 0.05|y| (SWMHD_example.jl:36 scaled) + a 0.1 two-Gaussian."""
    U = 1.0
    deta = F * U / G
    tg = two_gaussians(0.1)
    return dict(
        domain=dict(x=(-Lx / 2, Lx / 2), y=(-Ly / 2, Ly / 2)),
        formulation="VectorInvariant",
        h=lambda X, Y: 1.0 - deta * np.tanh(Y),
        u=lambda X, Y: U / np.cosh(Y) ** 2 + 1e-4 * np.exp(-Y ** 2) * np.sin(3 * X),
        v=lambda X, Y: 1e-4 * np.exp(-Y ** 2) * np.cos(2 * X),
        A=lambda X, Y: 0.05 * np.abs(Y) + tg(X, Y),
    )


def config2_uniform_bx(L=10.0):
    """1024^2 divergence formulation, 'uniform B_x': A = 0.5|y| (SWMHD_example.jl:36) + 1e-3 exp(-r^2), h = 1."""
    return dict(
        domain=dict(x=(-L / 2, L / 2), y=(-L / 2, L / 2)),
        formulation="Conservative",
        h=lambda X, Y: np.ones_like(X),
        u=lambda X, Y: np.zeros_like(X),
        v=lambda X, Y: np.zeros_like(X),
        A=lambda X, Y: 0.5 * np.abs(Y) + 1e-3 * np.exp(-(X ** 2 + Y ** 2)),
    )


def config4_two_gaussians(L=10.0):
    """8192^2 divergence formulation: divergence_sw_mhd.jl:33,35,38 (A two Gaussians amp 0.5, h = 1, uh = vh = 0)."""
    return dict(
        domain=dict(x=(-L / 2, L / 2), y=(-L / 2, L / 2)),
        formulation="Conservative",
        h=lambda X, Y: np.ones_like(X),
        u=lambda X, Y: np.zeros_like(X),
        v=lambda X, Y: np.zeros_like(X),
        A=two_gaussians(0.5),
    )


def config5_bickley_slab(Lx=2 * np.pi, Ly=20.0, nslabs=8):
    """16384^2 Jacobian formulation weak-scaled over 8 GPUs: the 16384 x 2048 slab of config 3's Bickley jet that contains the jet
    axis (y in [-Ly/16, Ly/16]), run as its own y-periodic domain on one GPU (BASELINE config 5; fp32-vs-fp64 sweep)."""
    c = config3_bickley(Lx, Ly)
    c["domain"] = dict(x=(-Lx / 2, Lx / 2), y=(-Ly / (2 * nslabs), Ly / (2 * nslabs)))
    return c


def config4_slab(L=10.0, nslabs=8):
    """8192^2 divergence formulation over 8 GPUs: the 8192 x 1024 slab that holds the two Gaussians (y in [-L/16, L/16]) as its
    own y-periodic domain (BASELINE config 4 at per-GPU size)."""
    c = config4_two_gaussians(L)
    c["domain"] = dict(x=(-L / 2, L / 2), y=(-L / (2 * nslabs), L / (2 * nslabs)))
    return c


# name -> (config builder, Nx, Ny, formulation) of every single-GPU workload BASELINE.json names
WORKLOADS = {
    "config2": (config2_uniform_bx, 1024, 1024, "Conservative"),
    "config3": (config3_bickley, 4096, 4096, "VectorInvariant"),
    "config4_slab": (config4_slab, 8192, 1024, "Conservative"),
    "config5_slab": (config5_bickley_slab, 16384, 2048, "VectorInvariant"),
}


def build_model(S, name, dtype=None, strict=False, kernel="auto", **kw):
    """ShallowWaterModel + grid for one of WORKLOADS, initial condition set and halos filled."""
    import torch
    mk, Nx, Ny, form = WORKLOADS[name]
    cfg = mk()
    g = S.RectilinearGrid(size=(Nx, Ny), x=cfg["domain"]["x"], y=cfg["domain"]["y"])
    m = S.ShallowWaterModel(g, G, F, formulation=form, dtype=dtype or torch.float64, strict=strict, kernel=kernel, **kw)
    n1, n2 = m.names[:2]
    m.set(**{n1: cfg["u"], n2: cfg["v"], "h": lambda X, Y: cfg["h"](X, Y) + 0 * X, "A": cfg["A"]})
    return m, g, cfg
