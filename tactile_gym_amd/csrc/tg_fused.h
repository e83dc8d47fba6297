// tg_fused.h - launch interface of the one-launch env step (tg_fused.hip: k_step_render).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tg_raster.h"

namespace tg {

struct State;

// edge_follow (a shared stimulus mesh of <= 32 triangles on the block raster), TCP_velocity_control, f64: step + auto-reset + tactile image(s) of
// every env in ONE launch - the wavefront that steps an env draws it.  d_bank: the context's BankDev (reset bank) or null.
// Returns 0, or -1 when the stimulus / image size has no block raster (the caller then takes k_step -> k_reset -> launch_render).
int launch_step_render(int topology, int num_envs, hipStream_t stream, const void* d_robot, const void* d_const, const State& st, const float* d_actions,
                       int auto_reset, const void* d_bank, const RasterParams& P, const Stimulus& S, const float* nodef_dep, const uint8_t* gray_u8,
                       const uint8_t* border, uint8_t* out, uint8_t* term_out);
int fused_envs_per_wave(int num_envs);   // E: envs stepped (one per lane) and drawn (one after the other) by one wavefront

}  // namespace tg
