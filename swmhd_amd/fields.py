"""Halo-padded device fields.  torch is used for device memory and streams only."""
import numpy as np
import torch

from . import _lib
from .grid import Center, Face

_SFX = {torch.float64: "f64", torch.float32: "f32"}
LOCS = {"u": (Face, Center), "uh": (Face, Center), "v": (Center, Face), "vh": (Center, Face),
        "h": (Center, Center), "A": (Center, Center)}


def _stream_ptr(stream=None):
    if stream is None:
        stream = torch.cuda.current_stream()
    return stream.cuda_stream


class Field:
    """A field on the staggered C-grid stored as its halo-padded parent, shape (Ny+2Hy, Nx+2Hx)."""

    def __init__(self, grid, loc=(Center, Center), dtype=torch.float64, device="cuda", data=None):
        self.grid, self.loc = grid, loc
        if data is None:
            data = torch.zeros(grid.parent_shape, dtype=dtype, device=device)
        # rows may be pitched (stride_y > Nx+2Hx, e.g. a column slice of a wider allocation); x must be unit-stride
        assert tuple(data.shape) == grid.parent_shape and data.stride(1) == 1 and data.stride(0) >= data.shape[1]
        self.data = data

    # --- reference-style helpers -------------------------------------------------------------
    def set(self, value):
        """set!(field, f(x, y)) or an array/number: evaluate at this field's nodes (halos included, coordinates
        extended linearly); call fill_halo_regions() afterwards for periodic wrap."""
        g = self.grid
        if callable(value):
            X, Y = g.nodes(self.loc)
            arr = np.asarray(value(X, Y), dtype=np.float64) + np.zeros(g.parent_shape)
        else:
            arr = np.asarray(value, dtype=np.float64)
            if arr.shape == (g.Ny, g.Nx):
                full = np.zeros(g.parent_shape)
                full[g.interior] = arr
                arr = full
            else:
                arr = arr + np.zeros(g.parent_shape)
        self.data.copy_(torch.from_numpy(arr).to(self.data.dtype))
        return self

    def fill_halo_regions(self, stream=None, boundary_conditions=None):
        """fill_halo_regions!(field) on the device: periodic copy in Periodic directions; in Bounded ones the default boundary condition
        of this field's location (no-flux mirror / impenetrable wall) or the given FieldBoundaryConditions (swmhd_fill_halo_*)."""
        import ctypes
        g = self.grid
        sfx = _SFX[self.data.dtype]
        tx, ty = g.topo_codes()
        ct = ctypes.c_double if sfx == "f64" else ctypes.c_float
        grads = boundary_conditions.gradients() if boundary_conditions is not None else [float("nan")] * 4
        f = getattr(_lib.lib(), f"swmhd_fill_halo_{sfx}")
        rc = f(_lib.ptr_array([self.data.data_ptr()]), 1, g.Nx, g.Ny, g.Hx, g.Hy, self.stride_y, tx, ty,
               1 if self.loc[0] == Face else 0, 1 if self.loc[1] == Face else 0, (ct * 4)(*grads), g.dx, g.dy, _stream_ptr(stream))
        _lib.check(rc, "swmhd_fill_halo")
        return self

    @property
    def stride_y(self):
        return self.data.stride(0)

    @property
    def ptr(self):
        return self.data.data_ptr()

    # --- dump / restore incl. halos (the reference writes fields `with_halos = true`,
    #     divergence_formulation/divergence_sw_mhd.jl:78-82).  Raw .npy of the parent, shape (Ny+2Hy, Nx+2Hx): transposing it
    #     gives Julia's (Nx+2Hx, Ny+2Hy) parent, so a real Oceananigans run can be diffed against it wherever Julia exists.
    def save(self, path):
        np.save(path, self.numpy())

    def load(self, path):
        arr = np.load(path, allow_pickle=False)
        if tuple(arr.shape) != self.grid.parent_shape:
            raise ValueError(f"{path}: shape {arr.shape} != parent shape {self.grid.parent_shape}")
        self.data.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(self.data.dtype))
        return self

    def interior(self):
        return self.data[self.grid.interior]

    def numpy(self):
        return self.data.detach().cpu().numpy()
