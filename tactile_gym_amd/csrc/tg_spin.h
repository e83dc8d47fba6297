// tg_spin.h - launch interface of object_balance's spinning_plate step (tg_spin.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tg {

struct State;

// One env step of object_balance with object_mode "spinning_plate" on `stream`: one wavefront per env (f64, UR5 chain).  n_dish: hull vertices of
// the dish (staged in LDS; the spool's ride on the lanes).  Returns 0, or -1 if the combination is not instantiated.
int launch_step_spin(int physics_dtype, int topology, int control_mode, int num_envs, int n_dish, hipStream_t stream, const void* d_robot,
                     const void* d_const, const State& st, const float* d_actions);

}  // namespace tg
