/*
 * tactile_gym_hip_test.h - C ABI of libtactile_gym_hip_test.so: device self-tests of pieces of the product's kernels.
 *
 * TEST INFRASTRUCTURE, not part of the boundary a maintainer binds (that is tactile_gym_hip.h / libtactile_gym_hip.so): built by
 * csrc/build.sh next to the product library from the same device headers, loaded only by tests/ (tactile_gym_amd._capi.test_lib()).
 * Every function returns 0 on success, -1 bad argument, -2 no HIP device / allocation failed, -3 launch failed.
 */
#ifndef TACTILE_GYM_HIP_TEST_H
#define TACTILE_GYM_HIP_TEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Self-test of the wave-mapped GJK / EPA (tg_config.narrowphase; csrc/tg_narrowphase.hpp) on n_cases placements of a convex hull against the
 * box of half extents half[3]: hulls [n_cases][n_hull][3] in the box frame (n_hull <= 1152); out [n_cases][11] = found (1 / 0), signed core
 * distance (< 0: overlap depth), unit normal from the box to the hull, witness point on the hull, witness point on the box. */
int tg_selftest_narrowphase(int32_t n_cases, int32_t n_hull, const double* hulls, const double* half, double* out);
/* The same for two hulls (csrc/tg_spin.hip's pair; oracle/narrowphase.c: mb_gjk_epa_hull_hull): hulls [n_cases][n_hull][3] = body A's hull in body
 * B's frame, hull_b [n_b][3] = body B's in its own frame (n_b <= 256); out as above. */
int tg_selftest_narrowphase_hulls(int32_t n_cases, int32_t n_hull, const double* hulls, int32_t n_b, const double* hull_b, double* out);

/* Self-test of the raster's depth division (tactile_sensor.py:239-294 reads an IEEE depth buffer): n pseudo-random operand pairs
 * with exponents 2^-40 .. 2^24 divided by the kernels' refinement and by the correctly rounded `/`; *mismatches = quotients whose
 * bits differ (must be 0). */
int tg_selftest_division(int64_t n, uint64_t seed, int64_t* mismatches);
/* The same refinement where t_s_camera divides the clipped penetration by max_penetration = 0.05 (tactile_sensor.py:284-289): EVERY float in
 * {0} u [1e-4, 0.05] divided by 0.05f both ways; *mismatches must be 0. */
int tg_selftest_penetration_division(int64_t* mismatches);
/* Self-test of the raster's edge-function block test (csrc/tg_raster.hip: edges_exclude_rect - a record is skipped for a block of pixels that
 * its triangle provably cannot cover): n pseudo-random triangles (image-sized, slivers, huge, on pixel centres, heightfield-sized) x
 * rectangles as the kernels pass them, every pixel centre put through the pixel loops' own edge expressions.  out[0] = rectangles
 * excluded although they hold a coverable pixel (must be 0), out[1] = rectangles excluded, out[2] = rectangles without a coverable pixel.
 * The converse rule of round 5 (edges_cover_rect: a block wholly inside the triangle needs no coverage test per pixel): out[3] = rectangles
 * called covered in which some pixel fails the pixel loops' coverage predicate (must be 0), out[4] = rectangles called covered, out[5] =
 * rectangles whose every pixel passes.  out: int64 [6]. */
int tg_selftest_edge_exclusion(int64_t n, uint64_t seed, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif
