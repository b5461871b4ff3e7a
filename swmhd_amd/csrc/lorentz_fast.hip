// Fast build of the Lorentz operator kernels: reciprocal multiplies, FMA contraction on.
#include "common.hpp"
#define SWMHD_STRICT 0
#define LAUNCH_SFX fast
#include "lorentz_tile_kernels.inc"
