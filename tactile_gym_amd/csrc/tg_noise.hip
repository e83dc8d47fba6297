// tg_noise.hip — per-episode surface generation for surface_follow (gfx950).
//
// Replaces BaseSurfaceEnv.update_surface -> gen_heigtfield_simplex_2d (reference
// tactile_gym/rl_envs/exploration/surface_follow/base_surface_env.py:319-337, :434-452): a 64x64 heightfield
// height[x][y] = OpenSimplex(seed).noise2(x*0.05, y*0.05) * 0.025 per env and episode.  The noise itself lives in the
// third-party `opensimplex` package (requirements.txt:4, unpinned); its published algorithm is restated here and,
// independently, in oracle/minibullet.c.  COMPILED WITH -ffp-contract=off: all arithmetic is IEEE double without
// fused multiply-adds, so the heights are bit-identical to the CPU oracle's.
//
// One 256-thread workgroup per resetting env: lane 0 builds the 256-entry permutation in LDS (a serial 64-bit LCG
// shuffle, 256 steps), then every lane evaluates rows*cols/256 samples; min/max for Bullet's vertical centring of the
// heightfield shape are reduced through LDS.  Envs whose mask byte is 0 exit immediately.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tg_noise.h"

namespace tg {

__device__ __constant__ int8_t kGrad2[16] = {5, 2, 2, 5, -5, 2, -2, 5, 5, -2, 2, -5, -5, -2, -2, -5};
constexpr double kStretch2D = -0.211324865405187;
constexpr double kSquish2D = 0.366025403784439;
constexpr double kNorm2D = 47.0;

__device__ __forceinline__ double extrapolate2(const int16_t* perm, long long xsb, long long ysb, double dx, double dy) {
    const int index = perm[(perm[xsb & 0xFF] + ysb) & 0xFF] & 0x0E;
    return (double)kGrad2[index] * dx + (double)kGrad2[index + 1] * dy;
}

__device__ double opensimplex_noise2(const int16_t* perm, double x, double y) {
    const double stretch = (x + y) * kStretch2D;
    const double xs = x + stretch, ys = y + stretch;
    const double fxs = floor(xs), fys = floor(ys);
    long long xsb = (long long)fxs, ysb = (long long)fys;
    const double squish = (fxs + fys) * kSquish2D;
    const double xb = fxs + squish, yb = fys + squish;
    const double xins = xs - fxs, yins = ys - fys;
    const double in_sum = xins + yins;
    double dx0 = x - xb, dy0 = y - yb;
    double value = 0.0, dx_ext, dy_ext;
    long long xsv_ext, ysv_ext;
    const double dx1 = dx0 - 1.0 - kSquish2D, dy1 = dy0 - 0.0 - kSquish2D;
    double attn1 = 2.0 - dx1 * dx1 - dy1 * dy1;
    if (attn1 > 0.0) { attn1 *= attn1; value += attn1 * attn1 * extrapolate2(perm, xsb + 1, ysb + 0, dx1, dy1); }
    const double dx2 = dx0 - 0.0 - kSquish2D, dy2 = dy0 - 1.0 - kSquish2D;
    double attn2 = 2.0 - dx2 * dx2 - dy2 * dy2;
    if (attn2 > 0.0) { attn2 *= attn2; value += attn2 * attn2 * extrapolate2(perm, xsb + 0, ysb + 1, dx2, dy2); }
    if (in_sum <= 1.0) {
        const double zins = 1.0 - in_sum;
        if (zins > xins || zins > yins) {
            if (xins > yins) { xsv_ext = xsb + 1; ysv_ext = ysb - 1; dx_ext = dx0 - 1.0; dy_ext = dy0 + 1.0; }
            else { xsv_ext = xsb - 1; ysv_ext = ysb + 1; dx_ext = dx0 + 1.0; dy_ext = dy0 - 1.0; }
        } else {
            xsv_ext = xsb + 1; ysv_ext = ysb + 1;
            dx_ext = dx0 - 1.0 - 2.0 * kSquish2D; dy_ext = dy0 - 1.0 - 2.0 * kSquish2D;
        }
    } else {
        const double zins = 2.0 - in_sum;
        if (zins < xins || zins < yins) {
            if (xins > yins) { xsv_ext = xsb + 2; ysv_ext = ysb + 0; dx_ext = dx0 - 2.0 - 2.0 * kSquish2D; dy_ext = dy0 + 0.0 - 2.0 * kSquish2D; }
            else { xsv_ext = xsb + 0; ysv_ext = ysb + 2; dx_ext = dx0 + 0.0 - 2.0 * kSquish2D; dy_ext = dy0 - 2.0 - 2.0 * kSquish2D; }
        } else { dx_ext = dx0; dy_ext = dy0; xsv_ext = xsb; ysv_ext = ysb; }
        xsb += 1; ysb += 1;
        dx0 = dx0 - 1.0 - 2.0 * kSquish2D; dy0 = dy0 - 1.0 - 2.0 * kSquish2D;
    }
    double attn0 = 2.0 - dx0 * dx0 - dy0 * dy0;
    if (attn0 > 0.0) { attn0 *= attn0; value += attn0 * attn0 * extrapolate2(perm, xsb, ysb, dx0, dy0); }
    double attn_ext = 2.0 - dx_ext * dx_ext - dy_ext * dy_ext;
    if (attn_ext > 0.0) { attn_ext *= attn_ext; value += attn_ext * attn_ext * extrapolate2(perm, xsv_ext, ysv_ext, dx_ext, dy_ext); }
    return value / kNorm2D;
}

// Draw number d (0-based) of the SplitMix64 stream whose state is `state` (tg_api.hip: rng_uniform; oracle/ref_env.py: Rng.uniform).
__device__ inline double splitmix_uniform(unsigned long long state, unsigned long long d, double lo, double hi) {
#pragma clang fp contract(off)
    unsigned long long z = state + (d + 1ULL) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    const double span = (hi - lo) * u;
    return lo + span;
}

// grid: n_envs blocks of 256 threads
__global__ __launch_bounds__(256) void k_gen_surface(int n_envs, const uint8_t* __restrict__ mask, const int64_t* __restrict__ seeds, int rows,
                                                     int cols, double interp, double range, int center_z, int mode,
                                                     double* __restrict__ heights, float* __restrict__ zoff, uint8_t* __restrict__ skip_mask,
                                                     const uint8_t* __restrict__ slot) {
    __shared__ int16_t perm[256];
    __shared__ int16_t source[256];
    __shared__ float red_min[256], red_max[256];
    const int env = blockIdx.x, tid = threadIdx.x;
    if (env >= n_envs) return;
    if (skip_mask != nullptr && skip_mask[env] != 0) {   // reset bank: this env took its precomputed entry - its surface is in the slot it switched to
        if (tid == 0) skip_mask[env] = 0;
        return;
    }
    if (mask != nullptr && mask[env] == 0) return;
    source[tid] = (int16_t)tid;
    __syncthreads();
    if (tid == 0 && (mode <= TG_SURF_SIMPLEX_1D || mode == TG_SURF_SIMPLEX_1D_VERT)) {   // OpenSimplex.__init__: three warm-up LCG steps, then a Fisher-Yates style draw without replacement
        unsigned long long s = (unsigned long long)seeds[env];
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        for (int i = 255; i >= 0; --i) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            const long long v = (long long)(s + 31ULL);
            long long r = v % (long long)(i + 1);
            if (r < 0) r += (i + 1);
            perm[i] = source[r];
            source[r] = source[i];
        }
    }
    __syncthreads();
    // slot (surface_follow env states, round 6): heights / zoff are [3][n_envs][...] and slot[env] names the third this surface goes to (the env's live
    // slot for a reset on the spot, the next one for the bank's refill); nullptr: one surface per env, [n_envs][...]
    const size_t idx = slot != nullptr ? (size_t)(slot[env] & 3) * n_envs + env : (size_t)env;
    double* out = heights + idx * rows * cols;
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int k = tid; k < rows * cols; k += 256) {
        const int x = k / cols, y = k % cols;       // heightfield_data[x, y], base_surface_env.py:327-335
        double h;
        if (mode == TG_SURF_FLAT) {                 // noise_mode "none" (:452-453)
            h = 0.0;
        } else if (mode == TG_SURF_RANDOM) {
            // noise_mode "random": gen_heigtfield_noisey (:302-317), one np_random.uniform(0, 0.2 range) per 2x2 block, drawn with the
            // column block j outer and the row block i inner.  seeds[env] is the env's SplitMix64 state before the first of these
            // draws (k_reset advanced its own copy past them), so draw number d is a pure function of (state, d).
            const int i = x >> 1, j = y >> 1;
            const unsigned long long d = (unsigned long long)j * (unsigned long long)(rows / 2) + (unsigned long long)i;
            h = (2 * i + 1 < rows && 2 * j + 1 < cols) ? splitmix_uniform((unsigned long long)seeds[env], d, 0.0, range * 0.2) : 0.0;
        } else if (mode == TG_SURF_SIMPLEX_1D_VERT) {   // gen_heigtfield_simplex_1d_vertical (:358-381): noise2(x c, 1 c), constant along y
            h = opensimplex_noise2(perm, (double)x * interp, 1.0 * interp) * range;
        } else {
            // 2-D: noise2(x c, y c) (gen_heigtfield_simplex_2d, :319-337); 1-D: noise2(1 c, y c), constant along x (_1d, :339-357)
            h = opensimplex_noise2(perm, (double)(mode == TG_SURF_SIMPLEX_1D ? 1 : x) * interp, (double)y * interp) * range;
        }
        out[k] = h;
        const float hf = (float)h;                  // Bullet receives the samples as float (PHY_FLOAT)
        lo = fminf(lo, hf); hi = fmaxf(hi, hf);
    }
    red_min[tid] = lo; red_max[tid] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red_min[tid] = fminf(red_min[tid], red_min[tid + s]); red_max[tid] = fmaxf(red_max[tid], red_max[tid + s]); }
        __syncthreads();
    }
    if (tid == 0 && zoff != nullptr) zoff[idx] = center_z ? 0.5f * (red_min[0] + red_max[0]) : 0.0f;
}

// object_push make_goal -> update_trajectory_simplex (object_push_env.py:248-281): y_i = noise2(i * 0.1, 1) * max_perturb - y_0,
// x_i = init_offset + i * spacing, yaw = np.gradient(y, spacing).  One 64-thread workgroup per resetting env.
// traj layout: [3][TG_MAX_TRAJ_POINTS = 16][n_envs]; feature: float [n_envs][12], goal slots 6..11 refreshed for goal 0.
__global__ __launch_bounds__(64) void k_gen_traj(int n_envs, const uint8_t* __restrict__ mask, const int64_t* __restrict__ seeds, int n_points,
                                                 double spacing, double max_perturb, double init_offset, int goal0,
                                                 double* __restrict__ traj, float* __restrict__ feature) {
    __shared__ int16_t perm[256];
    __shared__ int16_t source[256];
    __shared__ double ys[16];
    const int env = blockIdx.x, tid = threadIdx.x;
    if (env >= n_envs || (mask != nullptr && mask[env] == 0)) return;
    for (int k = tid; k < 256; k += 64) source[k] = (int16_t)k;
    __syncthreads();
    if (tid == 0) {
        unsigned long long s = (unsigned long long)seeds[env];
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        for (int i = 255; i >= 0; --i) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            const long long v = (long long)(s + 31ULL);
            long long r = v % (long long)(i + 1);
            if (r < 0) r += (i + 1);
            perm[i] = source[r];
            source[r] = source[i];
        }
    }
    __syncthreads();
    if (tid < n_points) ys[tid] = opensimplex_noise2(perm, (double)tid * 0.1, 1.0) * max_perturb;
    __syncthreads();
    if (tid < n_points) {
        const double off = -ys[0];
        const double y = off + ys[tid];
        const double x = init_offset + ((double)tid * spacing);
        double g;   // np.gradient of the offset trajectory
        if (tid == 0) g = ((off + ys[1]) - (off + ys[0])) / spacing;
        else if (tid == n_points - 1) g = ((off + ys[tid]) - (off + ys[tid - 1])) / spacing;
        else g = ((off + ys[tid + 1]) - (off + ys[tid - 1])) / (2.0 * spacing);
        traj[((size_t)0 * 16 + tid) * n_envs + env] = x;
        traj[((size_t)1 * 16 + tid) * n_envs + env] = y;
        traj[((size_t)2 * 16 + tid) * n_envs + env] = g;
        if (tid == goal0 && feature != nullptr) {   // goal slots of the extended_feature observation: the goal current after reset
            float* f = feature + (size_t)env * 12;
            f[6] = (float)x; f[7] = (float)y; f[8] = 0.0f; f[9] = 0.0f; f[10] = 0.0f; f[11] = (float)g;
        }
    }
}

void launch_gen_traj(int n_envs, const uint8_t* mask, const int64_t* seeds, int n_points, double spacing, double max_perturb, double init_offset,
                     int goal0, double* traj, float* feature, hipStream_t stream) {
    hipLaunchKernelGGL(k_gen_traj, dim3(n_envs), dim3(64), 0, stream, n_envs, mask, seeds, n_points, spacing, max_perturb, init_offset, goal0, traj,
                       feature);
}

void launch_gen_surface(int n_envs, const uint8_t* mask, const int64_t* seeds, int rows, int cols, double interp, double range, int center_z,
                        int mode, double* heights, float* zoff, hipStream_t stream, uint8_t* skip_mask, const uint8_t* slot) {
    hipLaunchKernelGGL(k_gen_surface, dim3(n_envs), dim3(256), 0, stream, n_envs, mask, seeds, rows, cols, interp, range, center_z, mode, heights,
                       zoff, skip_mask, slot);
}

}  // namespace tg
