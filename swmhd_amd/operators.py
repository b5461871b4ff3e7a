"""Whole-field forms of the reference's discrete forcing functions, dispatched to the HIP kernels in
libswmhd.so through the C-ABI (include/swmhd.h).

Reference signatures being mirrored (per-cell callbacks handed to Oceananigans' `Forcing(..., discrete_form=true)`):
    lorentz_force_func_x(i, j, k, grid, clock, fields)   jacobian_formulation/sw_mhd_jacobian_functions.jl:20-22
    lorentz_force_func_y(i, j, k, grid, clock, fields)   ... :24-26
    div_lorentz_x(i, j, k, grid, clock, fields)          divergence_formulation/sw_mhd_divergence_functions.jl:162-165
    div_lorentz_y(i, j, k, grid, clock, fields)          ... :167-170
`fields` is anything with `.A` and `.h` Field attributes (or a dict).  The value for every (i, j) of the
interior is returned in a pair of Fields located at (Face, Center) and (Center, Face).
"""
import torch

from . import _lib
from .fields import Field, _SFX, _stream_ptr
from .grid import Center, Face


def _get(fields, name):
    return fields[name] if isinstance(fields, dict) else getattr(fields, name)


def _call(form, grid, fields, out, strict, rows, stream, kernel="auto"):
    A, h = _get(fields, "A"), _get(fields, "h")
    if not (A.data.is_cuda and h.data.is_cuda):
        raise _lib.SwmhdError("swmhd_amd operators run on the GPU only (no CPU fallback); got a host tensor")
    assert A.data.dtype == h.data.dtype and A.stride_y == h.stride_y
    g = grid
    if out is None:
        out = (Field(g, (Face, Center), A.data.dtype, A.data.device), Field(g, (Center, Face), A.data.dtype, A.data.device))
    Fx, Fy = out
    assert Fx.stride_y == A.stride_y and Fy.stride_y == A.stride_y
    sfx = _SFX[A.data.dtype]
    flags = (_lib.STRICT if strict else _lib.FAST) | _lib.KERNEL_FLAGS[kernel]
    j0, j1 = (0, g.Ny) if rows is None else rows
    L = _lib.lib()
    sp = _stream_ptr(stream)
    if form == "jacobian":
        rc = getattr(L, f"swmhd_lorentz_jacobian_rows_{sfx}")(
            A.ptr, h.ptr, Fx.ptr, Fy.ptr, g.Nx, g.Ny, g.Hx, g.Hy, A.stride_y, g.dx, g.dy, j0, j1, flags, sp)
    else:
        tx, ty = g.topo_codes()
        rc = getattr(L, f"swmhd_lorentz_divergence_rows_{sfx}")(
            A.ptr, h.ptr, Fx.ptr, Fy.ptr, g.Nx, g.Ny, g.Hx, g.Hy, A.stride_y, g.dx, g.dy, tx, ty, j0, j1, flags, sp)
    _lib.check(rc, f"swmhd_lorentz_{form}")
    return Fx, Fy


def lorentz_force_func(grid, fields, out=None, strict=False, rows=None, stream=None, kernel="auto"):
    """(lorentz_force_func_x, lorentz_force_func_y) for all interior cells -- Jacobian formulation."""
    return _call("jacobian", grid, fields, out, strict, rows, stream, kernel)


def div_lorentz(grid, fields, out=None, strict=False, rows=None, stream=None, kernel="auto"):
    """(div_lorentz_x, div_lorentz_y) for all interior cells -- divergence (Maxwell-stress) formulation."""
    return _call("divergence", grid, fields, out, strict, rows, stream, kernel)
