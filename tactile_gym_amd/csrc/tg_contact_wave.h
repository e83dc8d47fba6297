// tg_contact_wave.h - launch interface of the wave-per-env contact solver (tg_contact_wave.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tg {

struct State;

// Enqueue one env step of object_push (env_kind TG_ENV_OBJECT_PUSH) or object_roll (TG_ENV_OBJECT_ROLL) on `stream` with the
// wave-per-env mapping: one 64-lane wavefront per env, one solver row per lane.  d_robot / d_const: DevRobot<T> / EnvConst<T> of the
// context (T by physics_dtype); n_tip_verts: hull vertices of the tip core (they are staged in LDS).  Returns 0, or -1 if the combination
// is not instantiated (the caller then takes the lane-per-env kernels).
int launch_step_contact_wave(int env_kind, int physics_dtype, int topology, int control_mode, int cone_friction, int num_envs, int n_tip_verts,
                             hipStream_t stream,
                             const void* d_robot, const void* d_const, const State& st, const float* d_actions, int narrowphase = 0);
// narrowphase (tg_config.narrowphase, object_push, f64): 0 closed forms; otherwise the tip - cube pair goes through GJK / EPA and the persistent
// manifold (tg_narrowphase.hpp): the kernel's four-tip-slot variant.

// object_balance (arm + pole + point-to-point constraint), TCP_velocity_control, f64, UR5: one wavefront per env (the env's own licence for the
// analytic fixed point, full ticks on the wave mapping).  inline_reset: finished envs are reset by their own wavefront at the end of the step
// (the template-only reset: the caller has made sure the template is valid) - no k_reset_body launch behind this one.  -1: not instantiated.
int launch_step_body_wave(int physics_dtype, int topology, int control_mode, int num_envs, hipStream_t stream, const void* d_robot, const void* d_const,
                          const State& st, const float* d_actions, int inline_reset);
// edge_follow / surface_follow (contact-free arm, TCP_velocity_control, f64; UR5 and MG400): one wavefront per env, every tick a full tick
// (lane-parallel dynamics, the motor pass as a linear map).  -1: not instantiated.
int launch_step_arm_wave(int physics_dtype, int topology, int control_mode, int num_envs, hipStream_t stream, const void* d_robot, const void* d_const,
                         const State& st, const float* d_actions);
// env.reset() for the envs flagged in d_mask (nullptr: all) with the same mapping: one wavefront per resetting env, the others exit at once.
int launch_reset_contact_wave(int env_kind, int physics_dtype, int topology, int cone_friction, int num_envs, int n_tip_verts, hipStream_t stream,
                              const void* d_robot, const void* d_const, const State& st, const uint8_t* d_mask, int narrowphase = 0);

}  // namespace tg
