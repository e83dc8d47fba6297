// tg_fused.hip - one launch per env step (round 5): the wavefront that steps an env also resets it when its episode ended and draws its
// tactile image(s).
//
// BaseTactileEnv.step is one chain per env - apply_action -> 24 sim ticks -> reward / done -> get_observation
// (reference rl_envs/base_tactile_env.py:166-185; the VecEnv's auto-reset behind it, sb3_helpers/rl_utils.py:17-30) - and until round 4
// that chain was three dependent launches (k_step -> k_reset -> k_render_blocks) of which the first and the last are latency chains on a
// nearly empty chip: 16 wavefronts for 16.7 us, then 4096 wavefronts whose slowest takes 17 us, plus the dispatch floors and graph gaps
// between them (44 us per step at 1024 envs).  The lane-mapped step's duration IS one lane's serial latency, whatever the number of lanes
// in use, so here every env group gets a wavefront of its own:
//
//   grid = ceil(n / E) workgroups of ONE wavefront (E = ceil(n / 1024): at 1024 envs one env per wavefront, one wavefront per SIMD -
//   the step code holds ~490 of the SIMD lane's 512 registers, so residency is exactly one wavefront per SIMD);
//   lanes 0 .. E-1 run step_env (tg_kernels.hpp: the code of k_step, unchanged) for envs blockIdx.x * E + lane, a finished env's lane
//   runs the reset (k_reset's body; the reset bank's swap-in when the context has one);
//   then the whole wavefront draws the E envs' images one after the other with the single-wavefront block raster
//   (tg_raster_dev.hpp: render_blocks_wave - k_render_blocks' arithmetic; a finished env gets its terminal image first).
//
// The camera transforms travel from the step lanes to the raster through registers (v_readlane), not through HBM.  Physics keeps the default
// FMA contraction (like tg_api.hip), the raster is contraction-free by the pragma in tg_raster_dev.hpp, which is included last.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tg_fused.h"
#include "tg_kernels.hpp"
#include "tg_raster_dev.hpp"   // LAST: switches FMA contraction off for everything below

namespace tg {

template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_step_render(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                    const float* __restrict__ actions, int E, int auto_reset, const BankDev* __restrict__ bd,
                                                    RasterParams P, Stimulus S, const float* __restrict__ nodef_dep,
                                                    const uint8_t* __restrict__ gray_u8, const uint8_t* __restrict__ border,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ term_out, int rec_cap) {
    KtScope kt_scope_(st.kt);
    extern __shared__ TriRec recs[];
    __shared__ int count;
    const int lane = threadIdx.x;
    const int n = cp->num_envs;
    const int env0 = blockIdx.x * E;
    const int env = env0 + lane;
    int dn = 0;
    float xs[12], xt[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { xs[k] = 0.0f; xt[k] = 0.0f; }
    if (lane < E && env < n) {
        step_env<T, TOPO>(*mp, *cp, st, env, actions);
        dn = st.done[env];                                    // (this lane's own store)
        if (auto_reset && dn) reset_or_swap<T, TOPO>(mp, cp, st, env, true, 0, bd);
        else dn = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) xs[k] = st.stim_xform[k * n + env];
        if (dn) {
#pragma unroll
            for (int k = 0; k < 12; ++k) xt[k] = st.term_xform[k * n + env];
        }
    }
    draw_counter_advance(st);
    const int n_regions = (P.W / 128) * (P.H / 128);
    const size_t img_bytes = (size_t)P.W * P.H;
    for (int e = 0; e < E; ++e) {
        const int ee = env0 + e;
        if (ee >= n) break;
        float M[12];
        if (__builtin_amdgcn_readlane(dn, e)) {               // the finished episode's last observation, then the new episode's first
#pragma unroll
            for (int k = 0; k < 12; ++k) M[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xt[k]), e));
            render_blocks_wave<kBlockW>(P, S, M, n_regions, nodef_dep, gray_u8, border, term_out + (size_t)ee * img_bytes, nullptr, rec_cap, recs, &count);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) M[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xs[k]), e));
        render_blocks_wave<kBlockW>(P, S, M, n_regions, nodef_dep, gray_u8, border, out + (size_t)ee * img_bytes,
                                    P.drawn != nullptr ? P.drawn + (size_t)ee * n_regions : nullptr, rec_cap, recs, &count);
    }
}

int fused_envs_per_wave(int num_envs) {
    int E = (num_envs + 1023) / 1024;
    return E < 1 ? 1 : (E > 64 ? 64 : E);
}

int launch_step_render(int topology, int num_envs, hipStream_t stream, const void* d_robot, const void* d_const, const State& st, const float* d_actions,
                       int auto_reset, const void* d_bank, const RasterParams& P, const Stimulus& S, const float* nodef_dep, const uint8_t* gray_u8,
                       const uint8_t* border, uint8_t* out, uint8_t* term_out) {
    if (S.kind != 0 || S.n_tris > 32 || P.blockmax == nullptr || P.tmpl == nullptr || P.W % 128 != 0 || P.H % 128 != 0) return -1;
    const int rec_cap = 2 * S.n_tris < 2 ? 2 : 2 * S.n_tris;
    const size_t lds = (size_t)rec_cap * sizeof(TriRec);
    const int E = fused_envs_per_wave(num_envs);
    const dim3 grid((num_envs + E - 1) / E), block(64);
    if (topology == 0)
        hipLaunchKernelGGL((k_step_render<double, 0>), grid, block, lds, stream, (const DevRobot<double>*)d_robot, (const EnvConst<double>*)d_const, st,
                           d_actions, E, auto_reset, (const BankDev*)d_bank, P, S, nodef_dep, gray_u8, border, out, term_out, rec_cap);
    else
        hipLaunchKernelGGL((k_step_render<double, 1>), grid, block, lds, stream, (const DevRobot<double>*)d_robot, (const EnvConst<double>*)d_const, st,
                           d_actions, E, auto_reset, (const BankDev*)d_bank, P, S, nodef_dep, gray_u8, border, out, term_out, rec_cap);
    return 0;
}

}  // namespace tg
