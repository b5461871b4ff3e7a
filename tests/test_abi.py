"""CPU tests (no GPU): libswmhd.so builds, loads and exports exactly what include/swmhd.h declares; the host-side
mirror of the reference interface behaves; the product never falls back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "swmhd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(swmhd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(swmhd):
    L = swmhd._lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/swmhd.h but not exported by libswmhd.so"
    assert sorted(swmhd._lib.EXPORTS) == declared, "swmhd_amd/_lib.py EXPORTS out of sync with include/swmhd.h"


def test_version_and_strerror(swmhd):
    L = swmhd._lib.lib()
    assert L.swmhd_version() == 300
    assert b"RCCL" in L.swmhd_strerror(4)
    assert b"success" in L.swmhd_strerror(0)
    assert b"halo" in L.swmhd_strerror(2)


def test_argument_validation_without_gpu(swmhd):
    """Validation happens before any HIP call, so error paths are testable on a CPU-only box."""
    L = swmhd._lib.lib()
    buf = (ctypes.c_double * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    f = L.swmhd_lorentz_jacobian_f64
    assert f(None, p, p, p, 4, 4, 2, 2, 8, 1.0, 1.0, 0, None) == 1          # null pointer
    assert f(p, p, p, p, 4, 4, 2, 2, 7, 1.0, 1.0, 0, None) == 1             # stride < Nx+2Hx
    assert f(p, p, p, p, 4, 4, 1, 2, 8, 1.0, 1.0, 0, None) == 2             # halo too small (needs 2)
    assert L.swmhd_lorentz_divergence_f64(p, p, p, p, 4, 4, 2, 2, 8, 1.0, 1.0, 0, None) == 2   # needs 3
    assert f(p, p, p, p, 4, 4, 2, 2, 8, -1.0, 1.0, 0, None) == 1            # dx <= 0
    assert f(p, p, p, p, 4, 4, 2, 2, 8, 1.0, 1.0, 99, None) == 1            # unknown flags
    g = L.swmhd_lorentz_divergence_rows_f64
    assert g(p, p, p, p, 4, 4, 3, 3, 10, 1.0, 1.0, 0, 0, 3, 2, 0, None) == 1  # j_begin > j_end
    assert g(p, p, p, p, 4, 4, 3, 3, 10, 1.0, 1.0, 7, 0, 0, 4, 0, None) == 1  # unknown topology code
    assert g(p, p, p, p, 4, 4, 3, 3, 10, 1.0, 1.0, 0, 0, 2, 2, 0, None) == 0  # empty row range is a no-op
    # tendency entry points: topology flags
    t = L.swmhd_tendencies_f64
    B = swmhd._lib
    args = (p,) * 8 + (4, 4, 3, 3, 10, 1.0, 1.0, 9.81, 1.0, 1, 1, 0, 4)
    assert t(*args, B.BOUNDED_X | B.WRAP_X, None) == 1                         # a direction is Bounded or wrapped, not both
    assert t(*args, B.BOUNDED_Y | B.MARCH_KERNEL, None) == 3                   # walls: LDS-tiled kernel only (SWMHD_ENOTSUP)
    assert t(*args[:-2], 2, 2, B.BOUNDED_Y, None) == 0                          # empty row range
    arr = (ctypes.c_void_p * 4)(p, p, p, p)
    assert L.swmhd_step_rk3_f64(arr, arr, arr, arr, 4, 4, 3, 3, 10, 1.0, 1.0, 9.81, 1.0, 1, 1, 0.1, 1, B.BOUNDED_X, None, None) == 3
    h = L.swmhd_fill_halo_f64
    assert h(arr, 4, 4, 4, 3, 3, 10, 0, 5, 1, 2, None, 1.0, 1.0, None) == 1    # unknown topology code
    assert h(arr, 5, 4, 4, 3, 3, 10, 0, 1, 1, 2, None, 1.0, 1.0, None) == 1    # more than 4 fields
    assert h(arr, 4, 2, 4, 3, 3, 10, 1, 1, 1, 2, None, 1.0, 1.0, None) == 2    # Nx < Hx


def test_geometry_query(swmhd):
    """swmhd_tendency_launch_geometry: bench.py derives the VALU floor from it.  No GPU needed (the CU count falls back to 256)."""
    g = swmhd._lib.tendency_launch_geometry(4096, 4096, 1, 8, 0)
    assert g["kind"] == 2 and g["threads"] == 256 and g["nstrips"] == 17 and g["nstrips"] * 250 >= 4096
    assert g["nseg"] * g["rows_per_segment"] >= 4096 and (g["nseg"] - 1) * g["rows_per_segment"] < 4096
    small = swmhd._lib.tendency_launch_geometry(128, 128, 1, 8, 0)
    assert small["kind"] == 1
    n1024 = swmhd._lib.tendency_launch_geometry(1024, 1024, 1, 8, 0)
    assert n1024["threads"] == 128 and n1024["nstrips"] == 9        # 9 x 128 lanes instead of 5 x 256


def test_grid_mirror(swmhd):
    g = swmhd.RectilinearGrid(size=(64, 32), x=(-5, 5), y=(-5, 5))
    assert g.parent_shape == (38, 70) and g.dx == 10 / 64 and g.dy == 10 / 32
    # Oceananigans: xᶜ[1] = x_west + dx/2, xᶠ[1] = x_west  (Julia index 1 == parent index H)
    assert g.xc[g.Hx] == -5 + g.dx / 2 and g.xf[g.Hx] == -5.0
    assert np.allclose(np.diff(g.yc), g.dy)
    slab = swmhd.RectilinearGrid(size=(64, 8), x=(-5, 5), y=(-5, 5), j_offset=8, Ny_global=32)
    assert slab.dy == g.dy and np.array_equal(slab.yc, g.yc[8:8 + 8 + 6])


def test_no_cpu_fallback(swmhd):
    """Host tensors are refused: the product path is the HIP library or nothing."""
    import torch
    g = swmhd.RectilinearGrid(size=(8, 8), x=(0, 1), y=(0, 1))
    A = swmhd.Field(g, device="cpu")
    h = swmhd.Field(g, device="cpu")
    with pytest.raises(swmhd._lib.SwmhdError):
        swmhd.lorentz_force_func(g, {"A": A, "h": h})
    with pytest.raises(swmhd._lib.SwmhdError):
        swmhd.div_lorentz(g, {"A": A, "h": h})


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "swmhd_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".inc", ".cpp")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, fn
