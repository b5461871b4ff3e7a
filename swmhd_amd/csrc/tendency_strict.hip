// Strict build of the fused tendency / RK3 kernels (-ffp-contract=off, oracle expression order).
#include "common.hpp"
#define SWMHD_STRICT 1
#define LAUNCH_SFX strict
#include "tendency_tile_kernels.inc"
