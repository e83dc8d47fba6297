// tg_noise.h — interface of the surface-generation translation unit (tg_noise.hip, compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {
// heights[env][rows*cols] (double, heightfield_data[x][y] flattened x*cols + y) and zoff[env] for envs whose mask byte is
// non-zero (mask == nullptr: all).  seeds[env] is the OpenSimplex seed drawn at reset.
void launch_gen_surface(int n_envs, const uint8_t* mask, const int64_t* seeds, int rows, int cols, double interp, double range, int center_z,
                        double* heights, float* zoff, hipStream_t stream);
}  // namespace tg
