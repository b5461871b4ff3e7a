"""ctypes/numpy front end of the CPU oracle (liboracle.so).

ORACLE = test infrastructure.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package (swmhd_amd) never does and fails loudly without its HIP library.

Arrays are halo-padded parents of shape (Ny+2Hy, Nx+2Hx), C-contiguous (x fastest) -- byte-for-byte the
layout of the column-major (Nx+2Hx, Ny+2Hy, 1) OffsetArray parents Oceananigans hands to the reference's
forcing functions (SURVEY.md section 8 conventions).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PERIODIC, BOUNDED = 0, 1
PROBES = {name: k for k, name in enumerate(
    ["jac_Bx", "jac_By", "jacobian_x", "jacobian_y", "lorentz_force_func_x", "lorentz_force_func_y",
     "div_Bx", "div_By", "div_hBx", "div_hBy", "lorentz_flux_hBx_bx", "lorentz_flux_hBy_bx",
     "lorentz_flux_hBx_by", "lorentz_flux_hBy_by", "div_lorentz_x", "div_lorentz_y", "test_jacobian_x", "test_jacobian_y"])}


def build(force=False):
    """Compile liboracle.so with oracle/Makefile (gcc, -ffp-contract=off)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        for sfx, ct in (("f64", C.c_double), ("f32", C.c_float)):
            p = C.c_void_p
            f = getattr(_LIB, f"oracle_lorentz_jacobian_{sfx}")
            f.argtypes = [p, p, p, p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, ct, ct, C.c_int]
            f.restype = C.c_int
            f = getattr(_LIB, f"oracle_lorentz_divergence_{sfx}")
            f.argtypes = [p, p, p, p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, ct, ct, C.c_int, C.c_int, C.c_int]
            f.restype = C.c_int
            f = getattr(_LIB, f"oracle_tendencies_{sfx}")
            f.argtypes = [p] * 8 + [C.c_int] * 4 + [C.c_long, ct, ct, ct, ct, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
            f.restype = C.c_int
            f = getattr(_LIB, f"oracle_fill_halo_{sfx}")
            f.argtypes = [p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, p, ct, ct]
            f.restype = None
            f = getattr(_LIB, f"oracle_fill_halo_periodic_{sfx}")
            f.argtypes = [p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long]
            f.restype = None
            f = getattr(_LIB, f"oracle_time_step_{sfx}")
            f.argtypes = [p] * 6 + [C.c_long] + [C.c_int] * 4 + [C.c_long, ct, ct, ct, ct, C.c_int, C.c_int, ct, C.c_int, C.c_int, p, C.c_int]
            f.restype = C.c_int
            _LIB.oracle_set_variant.argtypes = [C.c_char_p, C.c_double]
            _LIB.oracle_set_variant.restype = C.c_int
            f = getattr(_LIB, f"oracle_probe_{sfx}")
            f.argtypes = [C.c_int, p, p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, ct, ct,
                          C.c_int, C.c_int]
            f.restype = ct
    return _LIB


def set_variant(**kw):
    """Switch one of the version-dependent choices of oracle/sw_rhs.inc (see oracle_variant_t in swmhd_oracle.c); `reset=1`
    restores the defaults (= what the product kernels implement).  Sweep / discriminator use only."""
    for k, v in kw.items():
        if lib().oracle_set_variant(k.encode(), float(v)):
            raise KeyError(f"unknown oracle variant {k!r}")


def _sfx(a):
    if a.dtype == np.float64:
        return "f64"
    if a.dtype == np.float32:
        return "f32"
    raise TypeError(f"oracle supports float64/float32, got {a.dtype}")


def _check(*arrs):
    a0 = arrs[0]
    for a in arrs:
        assert a.flags["C_CONTIGUOUS"] and a.dtype == a0.dtype and a.shape == a0.shape, "layout mismatch"


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def lorentz_jacobian(A, h, Nx, Ny, Hx, Hy, dx, dy, nthreads=1):
    """(Fx@fcc, Fy@cfc) = lorentz_force_func_x/y over i=1:Nx, j=1:Ny (sw_mhd_jacobian_functions.jl:20-26)."""
    _check(A, h)
    assert A.shape == (Ny + 2 * Hy, Nx + 2 * Hx)
    Fx, Fy = np.zeros_like(A), np.zeros_like(A)
    rc = getattr(lib(), f"oracle_lorentz_jacobian_{_sfx(A)}")(
        _ptr(A), _ptr(h), _ptr(Fx), _ptr(Fy), Nx, Ny, Hx, Hy, A.shape[1], dx, dy, nthreads)
    if rc:
        raise ValueError(f"oracle_lorentz_jacobian rc={rc} (needs Hx,Hy >= 2)")
    return Fx, Fy


def lorentz_divergence(A, h, Nx, Ny, Hx, Hy, dx, dy, topo=(PERIODIC, PERIODIC), nthreads=1):
    """(Fx@fcc, Fy@cfc) = div_lorentz_x/y over i=1:Nx, j=1:Ny (sw_mhd_divergence_functions.jl:162-170)."""
    _check(A, h)
    assert A.shape == (Ny + 2 * Hy, Nx + 2 * Hx)
    Fx, Fy = np.zeros_like(A), np.zeros_like(A)
    rc = getattr(lib(), f"oracle_lorentz_divergence_{_sfx(A)}")(
        _ptr(A), _ptr(h), _ptr(Fx), _ptr(Fy), Nx, Ny, Hx, Hy, A.shape[1], dx, dy, topo[0], topo[1], nthreads)
    if rc:
        raise ValueError(f"oracle_lorentz_divergence rc={rc} (needs Hx,Hy >= 3)")
    return Fx, Fy


def probe(name, A, h, i, j, Nx, Ny, Hx, Hy, dx, dy, topo=(PERIODIC, PERIODIC)):
    """Evaluate one of the reference's point functions at Julia indices (i, j)."""
    _check(A, h)
    return getattr(lib(), f"oracle_probe_{_sfx(A)}")(
        PROBES[name], _ptr(A), _ptr(h), i, j, Nx, Ny, Hx, Hy, A.shape[1], dx, dy, topo[0], topo[1])


# ------------------------------------------------------------------------------------------------------------
# Base shallow-water RHS + RK3 (oracle/sw_rhs.inc) -- PARITY UNPINNED restatement of Oceananigans' ShallowWaterModel
# ------------------------------------------------------------------------------------------------------------
CONSERVATIVE, VECTOR_INVARIANT = 0, 1
LORENTZ_NONE, LORENTZ_JACOBIAN, LORENTZ_DIVERGENCE = 0, 1, 2


def tendencies(q1, q2, h, A, Nx, Ny, Hx, Hy, dx, dy, formulation, lorentz, g=9.81, f=1.0, nthreads=1, topo=(PERIODIC, PERIODIC)):
    """(G_q1, G_q2, G_h, G_A) for q = (uh, vh) [formulation 0] or (u, v) [formulation 1]; halos must be filled (fill_halo)."""
    _check(q1, q2, h, A)
    assert A.shape == (Ny + 2 * Hy, Nx + 2 * Hx)
    G = [np.zeros_like(A) for _ in range(4)]
    rc = getattr(lib(), f"oracle_tendencies_{_sfx(A)}")(
        _ptr(q1), _ptr(q2), _ptr(h), _ptr(A), *[_ptr(x) for x in G], Nx, Ny, Hx, Hy, A.shape[1], dx, dy, g, f,
        formulation, lorentz, topo[0], topo[1], nthreads)
    if rc:
        raise ValueError(f"oracle_tendencies rc={rc}")
    return tuple(G)


def fill_halo_periodic(a, Nx, Ny, Hx, Hy):
    _check(a)
    getattr(lib(), f"oracle_fill_halo_periodic_{_sfx(a)}")(_ptr(a), Nx, Ny, Hx, Hy, a.shape[1])
    return a


def _grad(a, grad):
    """4 GradientBoundaryCondition values (west, east, south, north; None = default BC) as an array of a's dtype, NaN = default."""
    g = np.full(4, np.nan, dtype=a.dtype)
    for k, v in enumerate(grad or ()):
        if v is not None:
            g[k] = v
    return g


def fill_halo(a, Nx, Ny, Hx, Hy, topo=(PERIODIC, PERIODIC), face=(False, False), grad=None, dx=1.0, dy=1.0):
    """fill_halo_regions! of one field in place: periodic wrap, or in Bounded directions the default BCs (no-flux mirror for
    centre-located fields, impenetrable walls for the normal velocity) / GradientBoundaryCondition values grad = (w, e, s, n)."""
    _check(a)
    gr = _grad(a, grad)
    getattr(lib(), f"oracle_fill_halo_{_sfx(a)}")(_ptr(a), Nx, Ny, Hx, Hy, a.shape[1], topo[0], topo[1], int(face[0]), int(face[1]),
                                                   _ptr(gr), dx, dy)
    return a


def time_step(q1, q2, h, A, Nx, Ny, Hx, Hy, dx, dy, dt, formulation, lorentz, g=9.81, f=1.0, nthreads=1, work=None,
              topo=(PERIODIC, PERIODIC), gradA=None):
    """One RK3 step in place (RungeKutta3: γ = 8/15, 5/12, 3/4; ζ = -17/60, -5/12).  Halos filled on entry and exit."""
    _check(q1, q2, h, A)
    n = A.size
    if work is None:
        work = (np.zeros(4 * n, A.dtype), np.zeros(4 * n, A.dtype))
    rc = getattr(lib(), f"oracle_time_step_{_sfx(A)}")(
        _ptr(q1), _ptr(q2), _ptr(h), _ptr(A), _ptr(work[0]), _ptr(work[1]), n, Nx, Ny, Hx, Hy, A.shape[1], dx, dy, g, f,
        formulation, lorentz, dt, topo[0], topo[1], _ptr(_grad(A, gradA)), nthreads)
    if rc:
        raise ValueError(f"oracle_time_step rc={rc}")
    return work
