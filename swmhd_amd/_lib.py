"""ctypes binding of libswmhd.so (include/swmhd.h).  The product path: fails loudly if the HIP library is
missing -- there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libswmhd.so")
_LIB = None

STRICT = 1
FAST = 0
TILE_KERNEL = 2
MARCH_KERNEL = 4
WRAP_X, WRAP_Y = 16, 32
LEAVE_ROOM = 64
GM_IS_PREV_STATE = 1024
BOUNDED_X, BOUNDED_Y = 256, 512
KERNEL_FLAGS = {None: 0, "auto": 0, "tile": 2, "march": 4}
PERIODIC, BOUNDED = 0, 1
HALO_X, HALO_Y = 1, 2
DIAG_NOUT, DIAG_WORKSPACE = 7, 1024 * 7
CONSERVATIVE, VECTOR_INVARIANT = 0, 1
LORENTZ_NONE, LORENTZ_JACOBIAN, LORENTZ_DIVERGENCE = 0, 1, 2


def ptr_array(ptrs):
    """Host array of device pointers (void*[n]) for the *_multi / rk3 entry points."""
    return (C.c_void_p * len(ptrs))(*ptrs)


class SwmhdError(RuntimeError):
    pass


def _declare(lib):
    p, i, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.swmhd_version.restype = i
    lib.swmhd_strerror.restype = C.c_char_p
    lib.swmhd_strerror.argtypes = [i]
    lib.swmhd_event_create.argtypes = [C.POINTER(p)]
    lib.swmhd_event_record.argtypes = [p, p]
    lib.swmhd_event_elapsed_ms.argtypes = [p, p, C.POINTER(C.c_float)]
    lib.swmhd_event_destroy.argtypes = [p]
    for n in ("create", "record", "elapsed_ms", "destroy"):
        getattr(lib, "swmhd_event_" + n).restype = i
    lib.swmhd_probe_fp64_issue.argtypes = [p, C.POINTER(C.c_float), p]
    lib.swmhd_probe_fp64_issue.restype = i
    lib.swmhd_probe_copy.argtypes = [p, p, C.c_size_t, i, C.POINTER(C.c_float), p]
    lib.swmhd_probe_copy.restype = i
    for sfx, ft in (("f64", C.c_double), ("f32", C.c_float)):
        for form in ("jacobian", "divergence"):
            f = getattr(lib, f"swmhd_lorentz_{form}_{sfx}")
            f.argtypes = [p, p, p, p, i, i, i, i, i64, ft, ft, i, p]
            f.restype = i
        f = getattr(lib, f"swmhd_lorentz_jacobian_rows_{sfx}")
        f.argtypes = [p, p, p, p, i, i, i, i, i64, ft, ft, i, i, i, p]
        f.restype = i
        f = getattr(lib, f"swmhd_lorentz_divergence_rows_{sfx}")
        f.argtypes = [p, p, p, p, i, i, i, i, i64, ft, ft, i, i, i, i, i, p]
        f.restype = i
        f = getattr(lib, f"swmhd_fill_halo_periodic_{sfx}")
        f.argtypes = [p, i, i, i, i, i64, i, p]
        f.restype = i
        f = getattr(lib, f"swmhd_fill_halo_{sfx}")
        f.argtypes = [C.POINTER(p), i, i, i, i, i, i64, i, i, i, i, p, ft, ft, p]
        f.restype = i
        f = getattr(lib, f"swmhd_fill_halo_periodic_multi_{sfx}")
        f.argtypes = [C.POINTER(p), i, i, i, i, i, i64, i, p]
        f.restype = i
        f = getattr(lib, f"swmhd_tendencies_{sfx}")
        f.argtypes = [p] * 8 + [i, i, i, i, i64, ft, ft, ft, ft, i, i, i, i, i, p]
        f.restype = i
        f = getattr(lib, f"swmhd_tendencies_rk3_{sfx}")
        f.argtypes = [C.POINTER(p)] * 4 + [i, i, i, i, i64, ft, ft, ft, ft, i, i, ft, ft, ft, i, i, i, i, p]
        f.restype = i
        f = getattr(lib, f"swmhd_step_rk3_{sfx}")
        f.argtypes = [C.POINTER(p)] * 4 + [i, i, i, i, i64, ft, ft, ft, ft, i, i, ft, i, i, C.POINTER(i), p]
        f.restype = i
        f = getattr(lib, f"swmhd_diagnostics_{sfx}")
        f.argtypes = [p, p, p, p, i, i, i, i, i64, ft, ft, ft, ft, i, i, i, p, p, p]
        f.restype = i
        f = getattr(lib, f"swmhd_rk3_substep_{sfx}")
        f.argtypes = [C.POINTER(p), C.POINTER(p), C.POINTER(p), i, i, i, i, i64, ft, ft, ft, i, i, i, p]
        f.restype = i
        f = getattr(lib, f"swmhd_ring_exchange_y_{sfx}")
        f.argtypes = [p, C.POINTER(p), i, i, i, i, i, i64, p]
        f.restype = i
        f = getattr(lib, f"swmhd_ring_step_rk3_{sfx}")
        f.argtypes = [p] + [C.POINTER(p)] * 4 + [i, i, i, i, i64, ft, ft, ft, ft, i, i, ft, i, i, C.POINTER(i), p]
        f.restype = i
    lib.swmhd_tendency_launch_geometry.argtypes = [i, i, i, i, i, C.POINTER(i)]
    lib.swmhd_tendency_launch_geometry.restype = i
    lib.swmhd_ring_available.argtypes = [C.c_char_p]
    lib.swmhd_ring_available.restype = i
    lib.swmhd_ring_unique_id.argtypes = [C.c_char_p, p]
    lib.swmhd_ring_unique_id.restype = i
    lib.swmhd_ring_create.argtypes = [C.POINTER(p), C.c_char_p, i, i, p]
    lib.swmhd_ring_create.restype = i
    lib.swmhd_ring_create_loopback.argtypes = [C.POINTER(p), i, C.c_double]
    lib.swmhd_ring_create_loopback.restype = i
    lib.swmhd_ring_destroy.argtypes = [p]
    lib.swmhd_ring_destroy.restype = i
    lib.swmhd_ring_last_error.argtypes = [p]
    lib.swmhd_ring_last_error.restype = C.c_char_p
    lib.swmhd_ring_comm_stream.argtypes = [p]
    lib.swmhd_ring_comm_stream.restype = p
    lib.swmhd_ring_join.argtypes = [p, p]
    lib.swmhd_ring_join.restype = i
    lib.swmhd_ring_time_launches.argtypes = [p, i]
    lib.swmhd_ring_time_launches.restype = i
    lib.swmhd_ring_launch_times.argtypes = [p, C.POINTER(C.c_float), C.POINTER(i), i]
    lib.swmhd_ring_launch_times.restype = i


# every symbol include/swmhd.h declares (tests/test_abi.py checks the .so exports each of them)
EXPORTS = ["swmhd_version", "swmhd_strerror", "swmhd_tendency_launch_geometry", "swmhd_probe_fp64_issue", "swmhd_probe_copy"] + ["swmhd_event_" + n for n in ("create", "record", "elapsed_ms", "destroy")] + [
    f"swmhd_{name}_{sfx}" for sfx in ("f64", "f32") for name in (
        "lorentz_jacobian", "lorentz_jacobian_rows", "lorentz_divergence", "lorentz_divergence_rows",
        "fill_halo", "fill_halo_periodic", "fill_halo_periodic_multi", "tendencies", "tendencies_rk3", "rk3_substep", "step_rk3", "diagnostics",
        "ring_exchange_y", "ring_step_rk3")] + [
    "swmhd_ring_" + name for name in ("available", "unique_id", "create", "create_loopback", "destroy", "last_error", "comm_stream", "join", "time_launches",
                                      "launch_times")]
RING_ID_BYTES = 128


class TimingEvent:
    """A HIP timing event without the system-scope fence (swmhd_event_*), with the two methods of torch.cuda.Event that the
    benchmarks use.  `record` takes the raw stream handle (default: torch's current stream)."""

    def __init__(self):
        self._e = C.c_void_p()
        check(lib().swmhd_event_create(C.byref(self._e)), "swmhd_event_create")

    def record(self, stream=None):
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        check(lib().swmhd_event_record(self._e, C.c_void_p(stream)), "swmhd_event_record")

    def elapsed_time(self, stop):
        ms = C.c_float()
        check(lib().swmhd_event_elapsed_ms(self._e, stop._e, C.byref(ms)), "swmhd_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self._e:
                lib().swmhd_event_destroy(self._e)
        except Exception:
            pass


def lib():
    """Load libswmhd.so (built in-tree by __graft_entry__.build() / swmhd_amd/csrc/Makefile)."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("SWMHD_LIBRARY", LIB_PATH)     # A/B builds of the same library (tools/): never a different backend
        if not os.path.exists(path):
            raise SwmhdError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  swmhd_amd has no CPU fallback.")
        _LIB = C.CDLL(path)
        _declare(_LIB)
    return _LIB


def tendency_launch_geometry(Nx, rows, formulation, elem_size=8, flags=0):
    """dict(kind, threads, nstrips, nseg, rows_per_segment, wg_per_cu, halo_lanes, cus) of the fast tendency launch (swmhd.h)."""
    out = (C.c_int * 8)()
    check(lib().swmhd_tendency_launch_geometry(Nx, rows, formulation, elem_size, flags, out), "swmhd_tendency_launch_geometry")
    keys = ("kind", "threads", "nstrips", "nseg", "rows_per_segment", "wg_per_cu", "halo_lanes", "cus")
    return dict(zip(keys, list(out)))


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources libswmhd.so is built from: profiles/ files record it, so that bench.py
    can tell whether a committed counter value still describes the kernels that are running."""
    import glob, hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.inc")) + glob.glob(os.path.join(src, "*.hpp"))
                    + [os.path.join(src, "Makefile")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def check(rc, what=""):
    if rc != 0:
        raise SwmhdError(f"{what}: rc={rc}: {lib().swmhd_strerror(rc).decode()}")
