// tg_broadphase.h - launch interface of the broadphase guard (tg_broadphase.hip); see include/tactile_gym_hip.h for what it checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tactile_gym_hip.h"

namespace tg {
struct State;
struct BpScene {                     // device copy of tg_broadphase (hull_verts -> device memory)
    tg_bp_box box[TG_BP_SLOTS];
    double margin, hull_margin, sphere_half, ball_radius, stim_pos[3];
    int32_t table_slot, has_ball;
    const double* hull;
};
// out: int32 [3][n] pairs | hits | mask; totals: {env-checks, pairs, hits}.  Returns 0, or -1 for an unsupported (dtype, topology).
int launch_broadphase(int physics_dtype, int topology, int n, hipStream_t stream, const void* d_robot, const BpScene* d_scene, const State& st,
                      int32_t* out, unsigned long long* totals);
}  // namespace tg
