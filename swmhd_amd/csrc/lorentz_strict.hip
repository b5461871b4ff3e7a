// Strict build of the Lorentz operator kernels: reference operation order, IEEE divides, and this
// translation unit is compiled with -ffp-contract=off (see Makefile) => bit-identical to the CPU oracle.
#include "common.hpp"
#define SWMHD_STRICT 1
#define LAUNCH_SFX strict
#include "lorentz_tile_kernels.inc"
