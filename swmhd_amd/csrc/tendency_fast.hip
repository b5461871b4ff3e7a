// Fast build of the fused tendency / RK3 kernels.
#include "common.hpp"
#define SWMHD_STRICT 0
#define LAUNCH_SFX fast
#include "tendency_tile_kernels.inc"
