/*
 * ORACLE -- test infrastructure, not product code.
 *
 * CPU restatement of the reference's hot path (writingindy/SWMHD), built as liboracle.so by
 * oracle/Makefile with -ffp-contract=off (Julia never fuses a*b+c).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (swmhd_amd/, libswmhd.so) never does.
 *
 * Pinning status:
 *   - Lorentz operators (lorentz_ops.inc): pinned against the reference's own
 *     analytic known-answer tests (test_formulations.jl:12-15,173-189,205-210: Lorentz force of a Gaussian,
 *     2nd-order convergence).  The reference stores no numeric outputs, so last-bit parity with a Julia run
 *     cannot be pinned here (no Julia toolchain in this image; nothing was denied).
 *   - Base shallow-water RHS + RK3 (sw_rhs.inc; row A9): lives in Oceananigans.jl, which the reference
 *     neither vendors nor pins.  PARITY UNPINNED -- restated from the library's published scheme.
 */
#include <math.h>
#include <stddef.h>

#define REAL double
#define SFX _f64
#include "lorentz_ops.inc"
#include "sw_rhs.inc"
#undef REAL
#undef SFX

#define REAL float
#define SFX _f32
#include "lorentz_ops.inc"
#include "sw_rhs.inc"
#undef REAL
#undef SFX
