"""A9 pinning kit, consumer side.  integration/make_oceananigans_fixtures.jl (a text deliverable: no Julia here) builds the
reference's two ShallowWaterModels on the committed inputs of tests/golden/model_48x40.npz and dumps Oceananigans' own tendencies
and the state after two RK3 steps as tests/golden/oceananigans_*.npy.  While those files do not exist the base right-hand side
 stays PARITY UNPINNED and these tests say so (xfail with that reason); the moment a maintainer commits
them, the oracle (here) and the HIP engine (-m gpu) are compared with the reference-generated numbers instead of with the
restatement's own.

Stated tolerance against a real Oceananigans run: 1e-12 max-norm relative (the library may order a few additions differently
from the restatement; a different WENO variant -- see DESIGN.md section 3's list of version-dependent choices -- would miss it
by ~1e-3 and be caught)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G, F = 9.81, 1.0
TOL = 1e-12
CASES = [("vi", 1, 1), ("cons", 0, 2)]


def reference(tag):
    g, s = (os.path.join(GOLDEN, f"oceananigans_{tag}_{k}.npy") for k in ("G", "after2"))
    if not (os.path.exists(g) and os.path.exists(s)):
        pytest.xfail("parity unpinned: tests/golden/oceananigans_*.npy absent (run integration/make_oceananigans_fixtures.jl where "
                     "Julia + the reference's Oceananigans exist)")
    return np.load(g, allow_pickle=False), np.load(s, allow_pickle=False)


def inputs(tag):
    z = np.load(os.path.join(GOLDEN, "model_48x40.npz"))
    return z, [np.ascontiguousarray(a) for a in z[f"{tag}_q"]]


def close(a, b, I):
    return np.abs(a[I] - b[I]).max() <= TOL * max(np.abs(b[I]).max(), 1e-300)


def test_kit_is_complete():
    """The generator script and its inputs are committed and agree on names and shapes (checked without running Julia)."""
    src = open(os.path.join(os.path.dirname(GOLDEN), "..", "integration", "make_oceananigans_fixtures.jl")).read()
    for needle in ("model_48x40.npz", "oceananigans_$(tag)_G.npy", "oceananigans_$(tag)_after2.npy", "VelocityStencil", "ConservativeFormulation",
                   "lorentz_force_func_x", "div_lorentz_x", "RungeKutta3", "time_step!"):
        assert needle in src, needle
    z = np.load(os.path.join(GOLDEN, "model_48x40.npz"))
    Nx, Ny, H = int(z["Nx"]), int(z["Ny"]), int(z["H"])
    for tag in ("vi", "cons"):
        assert z[f"{tag}_q"].shape == (4, Ny + 2 * H, Nx + 2 * H)


@pytest.mark.parametrize("tag,form,lor", CASES)
def test_oracle_against_reference_generated_fixture(oracle, tag, form, lor):
    Gref, Sref = reference(tag)
    z, q = inputs(tag)
    Nx, Ny, H, dx, dy, dt = int(z["Nx"]), int(z["Ny"]), int(z["H"]), float(z["dx"]), float(z["dy"]), float(z["dt"])
    I = (slice(H, H + Ny), slice(H, H + Nx))
    got = oracle.tendencies(*q, Nx, Ny, H, H, dx, dy, form, lor, G, F)
    for w, g_ in zip(Gref, got):
        assert close(g_, w, I), "oracle tendencies differ from the Oceananigans run: A9 restatement is wrong somewhere"
    s = [a.copy() for a in q]
    for _ in range(2):
        oracle.time_step(*s, Nx, Ny, H, H, dx, dy, dt, form, lor, G, F)
    for w, g_ in zip(Sref, s):
        assert close(g_, w, I)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,form,lor", CASES)
def test_hip_engine_against_reference_generated_fixture(swmhd, tag, form, lor):
    import torch
    Gref, Sref = reference(tag)
    z, q = inputs(tag)
    Nx, Ny, H, dx, dy, dt = int(z["Nx"]), int(z["Ny"]), int(z["H"]), float(z["dx"]), float(z["dy"]), float(z["dt"])
    g = swmhd.RectilinearGrid(size=(Nx, Ny), x=(0, dx * Nx), y=(0, dy * Ny), halo=(H, H))
    I = g.interior
    for strict in (True, False):
        m = swmhd.ShallowWaterModel(g, G, F, formulation="VectorInvariant" if form == 1 else "Conservative", strict=strict)
        for f, a in zip(m._raw_fields, q):
            f.data.copy_(torch.from_numpy(a))
        m.calculate_tendencies(); torch.cuda.synchronize()
        for w, gf in zip(Gref, m.Gn):
            assert close(gf.numpy(), w, I)
        m.time_step(dt); m.time_step(dt); m.synchronize()
        for w, f in zip(Sref, m.fields):
            assert close(f.numpy(), w, I)
