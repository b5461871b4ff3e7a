"""swmhd_amd -- MI355X-native shallow-water MHD tendency engine (hot path of writingindy/SWMHD).

The product is libswmhd.so (hand-written HIP for gfx950 behind the C-ABI in include/swmhd.h); this package is
the thin host side: grid/field containers and whole-field forms of the reference's forcing functions.
"""
from . import _lib
from .grid import RectilinearGrid, Periodic, Bounded, Flat, Center, Face, GradientBoundaryCondition, FieldBoundaryConditions
from .fields import Field
from .operators import lorentz_force_func, div_lorentz
from .model import ShallowWaterModel, loopback_rings, VectorInvariantFormulation, ConservativeFormulation
from .distributed import SlabDecomposition, exchange_y_halos

__all__ = ["RectilinearGrid", "Periodic", "Bounded", "Flat", "Center", "Face", "Field", "GradientBoundaryCondition", "FieldBoundaryConditions",
           "lorentz_force_func", "div_lorentz", "ShallowWaterModel", "VectorInvariantFormulation",
           "ConservativeFormulation", "SlabDecomposition", "exchange_y_halos", "_lib"]
