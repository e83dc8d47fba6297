// tg_noise.h — interface of the surface-generation translation unit (tg_noise.hip, compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {
// heights[env][rows*cols] (double, heightfield_data[x][y] flattened x*cols + y) and zoff[env] for envs whose mask byte is
// non-zero (mask == nullptr: all).  seeds[env] is the OpenSimplex seed drawn at reset.
// mode: 2-D simplex; 1-D simplex (varies along y only: movement modes yz / yzRx, gen_heigtfield_simplex_1d); flat (noise_mode "none");
// random 2x2 blocks (noise_mode "random", gen_heigtfield_noisey: seeds[] = the env's SplitMix64 state before the block draws).
enum { TG_SURF_SIMPLEX_2D = 0, TG_SURF_SIMPLEX_1D = 1, TG_SURF_FLAT = 2, TG_SURF_RANDOM = 3, TG_SURF_SIMPLEX_1D_VERT = 4 };   // _VERT: varies along x only
// skip_mask (reset bank): an env whose byte is set took its precomputed surface - nothing is generated for it, the byte is cleared - whatever
// `mask` says.  slot (env states: heights / zoff are [3][n_envs][...], round 6): slot[env] = the third this env's surface goes to; nullptr: [n_envs][...].
void launch_gen_surface(int n_envs, const uint8_t* mask, const int64_t* seeds, int rows, int cols, double interp, double range, int center_z,
                        int mode, double* heights, float* zoff, hipStream_t stream, uint8_t* skip_mask = nullptr, const uint8_t* slot = nullptr);
// object_push goal trajectory (update_trajectory_simplex): traj [3][16][n_envs] work-frame x, y, yaw; refreshes the goal half of
// the extended_feature rows (feature may be nullptr).
void launch_gen_traj(int n_envs, const uint8_t* mask, const int64_t* seeds, int n_points, double spacing, double max_perturb, double init_offset,
                     int goal0, double* traj, float* feature, hipStream_t stream);
}  // namespace tg
