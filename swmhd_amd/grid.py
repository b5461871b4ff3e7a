"""Host-side mirror of the bits of Oceananigans' RectilinearGrid the reference uses
(jacobian_formulation/SWMHD_example.jl:14-16, divergence_formulation/divergence_sw_mhd.jl:12-14):
uniform 2-D grid, topology (Periodic|Bounded, Periodic|Bounded, Flat), halo-padded fields."""
from dataclasses import dataclass, field

import numpy as np

Periodic, Bounded, Flat = "Periodic", "Bounded", "Flat"
Center, Face = "Center", "Face"


@dataclass
class RectilinearGrid:
    size: tuple            # (Nx, Ny)
    x: tuple               # (x_west, x_east)
    y: tuple               # (y_south, y_north)
    topology: tuple = (Periodic, Periodic, Flat)
    halo: tuple = (3, 3)   # WENO5 needs 3 (Oceananigans' default for this model)
    # y-slab decomposition: this rank owns global rows [j_offset, j_offset + Ny) of Ny_global
    j_offset: int = 0
    Ny_global: int = None
    xc: np.ndarray = field(init=False, repr=False)
    xf: np.ndarray = field(init=False, repr=False)
    yc: np.ndarray = field(init=False, repr=False)
    yf: np.ndarray = field(init=False, repr=False)

    def __post_init__(self):
        self.Nx, self.Ny = int(self.size[0]), int(self.size[1])
        self.Hx, self.Hy = int(self.halo[0]), int(self.halo[1])
        if self.Ny_global is None:
            self.Ny_global = self.Ny
        self.Lx = float(self.x[1] - self.x[0])
        self.Ly = float(self.y[1] - self.y[0])
        self.dx = self.Lx / self.Nx
        self.dy = self.Ly / self.Ny_global
        # node coordinates INCLUDING halos, extended linearly (as grid.xᶜᵃᵃ etc. are in Oceananigans):
        # index k of these arrays is Julia index k - H + 1, i.e. parent position k.
        ii = np.arange(-self.Hx, self.Nx + self.Hx)
        jj = np.arange(-self.Hy, self.Ny + self.Hy) + self.j_offset
        self.xf = self.x[0] + ii * self.dx
        self.xc = self.x[0] + (ii + 0.5) * self.dx
        self.yf = self.y[0] + jj * self.dy
        self.yc = self.y[0] + (jj + 0.5) * self.dy

    @property
    def parent_shape(self):
        """(rows, cols) = (Ny+2Hy, Nx+2Hx); C-contiguous == Julia's column-major (Nx+2Hx, Ny+2Hy, 1) parent."""
        return (self.Ny + 2 * self.Hy, self.Nx + 2 * self.Hx)

    @property
    def interior(self):
        return (slice(self.Hy, self.Hy + self.Ny), slice(self.Hx, self.Hx + self.Nx))

    def nodes(self, loc):
        """2-D coordinate arrays (X, Y) of parent shape for a field at loc = (Center|Face, Center|Face)."""
        x = self.xc if loc[0] == Center else self.xf
        y = self.yc if loc[1] == Center else self.yf
        return np.meshgrid(x, y)

    def topo_codes(self):
        code = {Periodic: 0, Bounded: 1}
        return code[self.topology[0]], code[self.topology[1]]


# --- boundary conditions (the names the reference's commented lines use: SWMHD_example.jl:18-19, divergence_sw_mhd.jl:17) ---------
@dataclass
class GradientBoundaryCondition:
    """GradientBoundaryCondition(g): the first halo point is linearly extrapolated with slope g (Oceananigans fills only that one)."""
    gradient: float


@dataclass
class FieldBoundaryConditions:
    """FieldBoundaryConditions(north = GradientBoundaryCondition(-0.05), south = ...): sides left at None keep the default of the
    field's location (no-flux for centre-located fields, impenetrable walls for the normal velocity).  Bounded directions only."""
    west: GradientBoundaryCondition = None
    east: GradientBoundaryCondition = None
    south: GradientBoundaryCondition = None
    north: GradientBoundaryCondition = None

    def gradients(self):
        """(west, east, south, north) as floats, NaN = default boundary condition."""
        return [float("nan") if b is None else float(b.gradient) for b in (self.west, self.east, self.south, self.north)]
