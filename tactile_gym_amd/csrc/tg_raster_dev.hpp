// tg_raster_dev.hpp - device helpers of the tactile raster shared by tg_raster.hip (the stand-alone render kernels) and tg_fused.hip (the
// env step and its render in one launch): the projected-triangle record, set-up, the conservative culls and the single-wavefront form of
// the block raster.  EVERY float expression in this header follows the raster specification of DESIGN.md section 5 operation by operation:
// no FMA contraction - the pragma below holds from here to the end of the including translation unit (include this header LAST).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tg_raster.h"
#include "tg_kt.hpp"

#pragma clang fp contract(off)

namespace tg {

#ifndef TG_EDGE_REACH
#define TG_EDGE_REACH 1
#endif
constexpr bool kEdgeReach = TG_EDGE_REACH != 0;   // A/B build switch for the edge-function block test (edges_exclude_rect)

struct TriRec {      // projected triangle, window coordinates + depth
    float x0, y0, d0, x1, y1, d1, x2, y2, d2;
    float ymin, ymax, xmin, xmax, dmin;
};

// Depth culling margin.  The interpolated depth ((e0 d0 + e1 d1) + e2 d2) / s has same-signed weights, so it lies within
// ~6 roundings (< 4e-7 for d <= 1.01) of the convex hull [min d_k, max d_k]; a triangle whose min d_k exceeds a z value
// by more than kDepthSlack can therefore never pass `d < z` there.  Skipping it does not change the image.
constexpr float kDepthSlack = 2e-6f;

// a / b rounded like the IEEE division the specification (and the oracle's C `/`) prescribes, for operands whose exponents are far from the
// ends of the range (pixel-space edge functions: 1e-12 .. 1e6): the refinement the compiler's own expansion performs - reciprocal, one
// Newton step on it, the product and two residual corrections - without the operand pre-scaling, the denormal-mode switches and the
// special-case fix-up that only matter beyond 2^+-96.  The result is discarded by the caller when b == 0.
__device__ __forceinline__ float div_mid_range(float a, float b) {
    float y = __builtin_amdgcn_rcpf(b);
    y = __builtin_fmaf(__builtin_fmaf(-b, y, 1.0f), y, y);
    float q = a * y;
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
    return q;
}

__device__ __forceinline__ void project_vertex(float cx, float cy, float cw, const RasterParams& P, float& sx, float& sy, float& d) {
    float iw = 1.0f / cw;
    sx = P.hw + P.kx * (cx * iw);
    sy = P.hh - P.ky * (cy * iw);
    d = P.C0 + P.C1 * iw;
}

// Back-face test in eye space (eye at the origin, x right, y up, -z forward; v_k = (cx, cy, -cw)).  For a stimulus made of closed,
// consistently outward-wound surfaces (Stimulus::closed_outward, verified on the host) that lies entirely beyond the near plane, a ray
// from the eye enters every solid through a front face before it leaves it through a back face, so a back face can never win the depth
// test: dropping it at set-up leaves the image as it is.  (The pixel predicates agree: the edge function of a shared edge is exactly
// antisymmetric in its two vertices, so a pixel centre is on the solid's side of a silhouette edge for the front face and the back face
// alike.)  Faces within 1e-4 rad of edge-on are kept.  `cull` is wave-uniform per env: closed_outward and no vertex with w < near.
__device__ __forceinline__ bool back_facing(const float* cx, const float* cy, const float* cw) {
    const float ax = cx[1] - cx[0], ay = cy[1] - cy[0], az = -(cw[1] - cw[0]);
    const float bx = cx[2] - cx[0], by = cy[2] - cy[0], bz = -(cw[2] - cw[0]);
    const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float nv = (nx * cx[0] + ny * cy[0]) + nz * (-cw[0]);                       // n . v0: > 0 = the outward normal points away from the eye
    const float nn = (nx * nx + ny * ny) + nz * nz, vv = (cx[0] * cx[0] + cy[0] * cy[0]) + cw[0] * cw[0];
    return nv > 0.0f && nv * nv > 1e-8f * (nn * vv);
}
// All three vertices of triangle t at or beyond the near plane?  (the per-env vote that licenses the back-face cull)
__device__ __forceinline__ bool tri_beyond_near(const float* __restrict__ soup, int t, const float* M, float near_) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* v = soup + 9 * t + 3 * k;
        ok = ok && (-(((M[6] * v[0] + M[7] * v[1]) + M[8] * v[2]) + M[11]) >= near_);
    }
    return ok;
}

// Returns false when the record buffer is full (nothing written): the caller restarts from this triangle in the next round.
// rbands (banded lane mapping only): per record, bit b = the bounding box reaches pixel centres of the tile's b-th 32-pixel column band.
__device__ __forceinline__ bool emit(TriRec* recs, int* count, int cap, const float* vx, const float* vy, const float* vw, int a, int b, int c,
                                     const RasterParams& P, float tx0, float ty0, float tx1, float ty1, unsigned* rbands = nullptr) {
    TriRec r;
    project_vertex(vx[a], vy[a], vw[a], P, r.x0, r.y0, r.d0);
    project_vertex(vx[b], vy[b], vw[b], P, r.x1, r.y1, r.d1);
    project_vertex(vx[c], vy[c], vw[c], P, r.x2, r.y2, r.d2);
    r.xmin = fminf(r.x0, fminf(r.x1, r.x2)); r.xmax = fmaxf(r.x0, fmaxf(r.x1, r.x2));
    r.ymin = fminf(r.y0, fminf(r.y1, r.y2)); r.ymax = fmaxf(r.y0, fmaxf(r.y1, r.y2));
    r.dmin = fminf(r.d0, fminf(r.d1, r.d2)) - kDepthSlack;
    // conservative culls: outside this workgroup's tile, or entirely behind the undeformed skin/body depth image
    if (r.xmax < tx0 || r.xmin > tx1 || r.ymax < ty0 || r.ymin > ty1) return true;
    if (r.dmin >= P.zcull) return true;
    const int slot = atomicAdd(count, 1);
    if (slot >= cap) return false;
    recs[slot] = r;
    if (rbands != nullptr) {
        unsigned bands = 0u;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)   // band bb holds pixel centres tx0 + 32 bb + 0.5 .. + 31.5; coverage needs xmin <= fx <= xmax
            bands |= (r.xmax >= tx0 + 32.0f * (float)bb + 0.5f && r.xmin <= tx0 + 32.0f * (float)bb + 31.5f) ? (1u << bb) : 0u;
        // bits 4..: the row groups (8 rows each: one pass of the banded mapping) whose pixel centres the bounding box can reach
        const float ylo = (r.ymin - ty0 - 7.5f) * 0.125f, yhi = (r.ymax - ty0 - 0.5f) * 0.125f;
        int k_lo = (int)ceilf(ylo), k_hi = (int)floorf(yhi);
        k_lo = k_lo < 0 ? 0 : k_lo; k_hi = k_hi > 27 ? 27 : k_hi;
        const unsigned rows = k_hi >= k_lo ? (((k_hi - k_lo + 1 >= 28) ? 0xFFFFFFFu : ((1u << (k_hi - k_lo + 1)) - 1u)) << k_lo) : 0u;
        rbands[slot] = bands | (rows << 4);
    }
    return true;
}

// Can the record cover ANY pixel centre of the rectangle [X0, X1] x [Y0, Y1] (first / last pixel centres of a block)?  A pixel is
// covered only if its three computed edge functions e_i are all >= 0 or all <= 0 (the `pos | neg` of the pixel loops).  In exact
// arithmetic E_i is affine in (fx, fy), so over the rectangle it is extremal at a corner, and E_0 + E_1 + E_2 = S is the same everywhere
// (twice the signed area).  The computed e_i (two differences, two products, one difference: the pixel loops' expression) differs from E_i
// by at most m_i = 1e-5 (B_j A_k + B_k A_j) with B, A the largest |x - fx|, |y - fy| over the rectangle - 40 times the worst-case
// rounding of that expression (4 x 2^-24).  Hence:  some edge with max over the corners of e_i < -2 m_i  ->  e_i < 0 at every pixel, no
// pixel is `pos`;  S certainly > sum m_i (computed sum at a corner > 2 sum m_i)  ->  the three e_i cannot all be <= 0 anywhere, no pixel
// is `neg`; and the mirror images.  Returns true when both are excluded: skipping the record for this rectangle changes no pixel.
// NaN / inf coordinates compare false everywhere: not excluded.  (What it buys: the two coplanar triangles of a box face or of the plate
// both have the face as bounding box and depth plane, and each covers half of it.)
__device__ __forceinline__ bool edges_exclude_rect(float x0, float y0, float x1, float y1, float x2, float y2, float X0, float X1, float Y0, float Y1) {
    const float B0 = fmaxf(fabsf(x0 - X0), fabsf(x0 - X1)), B1 = fmaxf(fabsf(x1 - X0), fabsf(x1 - X1)), B2 = fmaxf(fabsf(x2 - X0), fabsf(x2 - X1));
    const float A0 = fmaxf(fabsf(y2 - Y0), fabsf(y2 - Y1)), A1 = fmaxf(fabsf(y1 - Y0), fabsf(y1 - Y1)), A2 = fmaxf(fabsf(y0 - Y0), fabsf(y0 - Y1));
    const float m0 = 1e-5f * (B1 * A0 + B2 * A1), m1 = 1e-5f * (B2 * A2 + B0 * A0), m2 = 1e-5f * (B0 * A1 + B1 * A2);
    float hi0 = -3.0e38f, hi1 = -3.0e38f, hi2 = -3.0e38f, lo0 = 3.0e38f, lo1 = 3.0e38f, lo2 = 3.0e38f, s_lo = 3.0e38f, s_hi = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float fx = (c & 1) ? X1 : X0, fy = (c & 2) ? Y1 : Y0;
        const float a0 = y2 - fy, a1 = y1 - fy, a2 = y0 - fy;
        const float e0 = (x1 - fx) * a0 - (x2 - fx) * a1;
        const float e1 = (x2 - fx) * a2 - (x0 - fx) * a0;
        const float e2 = (x0 - fx) * a1 - (x1 - fx) * a2;
        hi0 = fmaxf(hi0, e0); hi1 = fmaxf(hi1, e1); hi2 = fmaxf(hi2, e2);
        lo0 = fminf(lo0, e0); lo1 = fminf(lo1, e1); lo2 = fminf(lo2, e2);
        const float sc = (e0 + e1) + e2;
        s_lo = fminf(s_lo, sc); s_hi = fmaxf(s_hi, sc);
    }
    const float M2 = 2.0f * ((m0 + m1) + m2);
    const bool no_pos = (hi0 < -2.0f * m0) | (hi1 < -2.0f * m1) | (hi2 < -2.0f * m2) | (s_hi < -M2);
    const bool no_neg = (lo0 > 2.0f * m0) | (lo1 > 2.0f * m1) | (lo2 > 2.0f * m2) | (s_lo > M2);
    return no_pos & no_neg;
}


// Does the record cover EVERY pixel centre of the rectangle - with its three computed edge functions all > 0 (or all < 0), i.e. `pos` (or `neg`)
// true and s != 0 at every pixel, whatever the rounding?  Same bounds as edges_exclude_rect: e_i is affine, so over the rectangle its extremes
// sit at the corner pixel centres, and a computed value is within m_i of the exact one; min over the corners of the computed e_i > 2 m_i means
// the exact e_i > m_i at every corner, hence everywhere, hence the computed e_i > 0 at every pixel.  A pixel inside the triangle is inside its
// bounding box (exact comparisons of exact coordinates), so for such a (record, rectangle) the pixel loops' coverage predicate is `true` and
// only the depth test remains.  NaN / inf compare false: not covered.
__device__ __forceinline__ bool edges_cover_rect(float x0, float y0, float x1, float y1, float x2, float y2, float X0, float X1, float Y0, float Y1) {
    const float B0 = fmaxf(fabsf(x0 - X0), fabsf(x0 - X1)), B1 = fmaxf(fabsf(x1 - X0), fabsf(x1 - X1)), B2 = fmaxf(fabsf(x2 - X0), fabsf(x2 - X1));
    const float A0 = fmaxf(fabsf(y2 - Y0), fabsf(y2 - Y1)), A1 = fmaxf(fabsf(y1 - Y0), fabsf(y1 - Y1)), A2 = fmaxf(fabsf(y0 - Y0), fabsf(y0 - Y1));
    const float m0 = 1e-5f * (B1 * A0 + B2 * A1), m1 = 1e-5f * (B2 * A2 + B0 * A0), m2 = 1e-5f * (B0 * A1 + B1 * A2);
    float hi0 = -3.0e38f, hi1 = -3.0e38f, hi2 = -3.0e38f, lo0 = 3.0e38f, lo1 = 3.0e38f, lo2 = 3.0e38f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float fx = (c & 1) ? X1 : X0, fy = (c & 2) ? Y1 : Y0;
        const float a0 = y2 - fy, a1 = y1 - fy, a2 = y0 - fy;
        const float e0 = (x1 - fx) * a0 - (x2 - fx) * a1;
        const float e1 = (x2 - fx) * a2 - (x0 - fx) * a0;
        const float e2 = (x0 - fx) * a1 - (x1 - fx) * a2;
        hi0 = fmaxf(hi0, e0); hi1 = fmaxf(hi1, e1); hi2 = fmaxf(hi2, e2);
        lo0 = fminf(lo0, e0); lo1 = fminf(lo1, e1); lo2 = fminf(lo2, e2);
    }
    const bool all_pos = (lo0 > 2.0f * m0) & (lo1 > 2.0f * m1) & (lo2 > 2.0f * m2);
    const bool all_neg = (hi0 < -2.0f * m0) & (hi1 < -2.0f * m1) & (hi2 < -2.0f * m2);
    return all_pos | all_neg;
}

// ------------------------------------------------------------------------------------------------ block raster, one wavefront per image
// k_render_blocks (tg_raster.hip) restated for ONE wavefront that draws a whole image by itself - the render half of k_step_render
// (tg_fused.hip), where the wavefront that has just stepped an env draws that env's image in the same launch.  Same set-up (one triangle per
// lane), same lane-as-record planes, same reach tests, same pixel arithmetic expression by expression; what differs is the mapping only:
//   * no workgroup barrier and no LDS masks: a record's reach mask lives in the record's lane (v_readlane at a uniform index);
//   * a lane owns ONE quad (4 x 1 pixels) of each of NB reached blocks per round - 64 lanes = the 16 x 16 block - so a (record, block) pair
//     the reach test excluded is skipped as a scalar branch per block (the four-wavefront form skips per group of four blocks);
//   * the latency the four-wavefront form hides by occupancy is hidden here by the NB independent quads of a lane (the fused kernel holds
//     the step's 512 registers anyway: one wavefront per SIMD).
// The depth test keeps the smallest d and every cull is conservative, so the image is that of k_render_blocks bit for bit
// (tests/test_gpu_fused_step.py compares the two over rollouts with resets).  recs: >= rec_cap records of LDS; count: one LDS word.
#ifndef TG_FUSED_NB
#define TG_FUSED_NB 8
#endif
template <int BW>
__device__ __forceinline__ void render_blocks_wave(const RasterParams& P, const Stimulus& S, const float (&M)[12], int n_regions,
                                                   const float* __restrict__ nodef_dep, const uint8_t* __restrict__ gray_u8,
                                                   const uint8_t* __restrict__ border, uint8_t* __restrict__ dst /* this env's image */,
                                                   unsigned long long* drawn_env /* this env's [n_regions] changed-block records, or null */,
                                                   int rec_cap, TriRec* recs, int* count) {
    constexpr int BH = 256 / BW, NBX = 128 / BW, NB = TG_FUSED_NB;
    static_assert(BW == 16, "one 16-byte word per block row; 64 lanes = one block of quads");
    const int lane = threadIdx.x & 63;
    const int n_tris = S.n_tris;
    const int regions_x = P.W / 128;
    const float eps = 1e-4f, max_pen = 0.05f;
    constexpr float kGrey = 1.5e-4f;
    for (int reg = 0; reg < n_regions; ++reg) {
        const int rx = (reg % regions_x) * 128, ry = (reg / regions_x) * 128;
        const float bmax_l = P.blockmax[reg * 64 + lane];          // lane l <-> block l: its largest undeformed depth
        unsigned long long* drawn_p = drawn_env ? drawn_env + reg : nullptr;
        unsigned long long stale = ~0ull;
        if (drawn_p) stale = *drawn_p;
        const float tx0 = (float)rx, tx1 = (float)(rx + 128), ty0 = (float)ry, ty1 = (float)(ry + 128);
        __syncthreads();                                           // (one wavefront: orders the LDS traffic of the previous image / region)
        if (lane == 0) *count = 0;
        const bool has = lane < n_tris;
        float cx[3] = {0.0f, 0.0f, 0.0f}, cy[3] = {0.0f, 0.0f, 0.0f}, cw[3] = {0.0f, 0.0f, 0.0f};
        bool beyond = true;
        if (has) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float* v = S.soup + 9 * lane + 3 * k;
                const float vx = v[0], vy = v[1], vz = v[2];
                cx[k] = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
                cy[k] = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
                cw[k] = -(((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11]);
                beyond = beyond && (cw[k] >= P.near_);
            }
        }
        const bool cull = __ballot(!beyond) == 0ull && S.closed_outward != 0;
        if (has && !(cull && back_facing(cx, cy, cw))) {
            if ((cw[0] >= P.near_) & (cw[1] >= P.near_) & (cw[2] >= P.near_)) {
                emit(recs, count, rec_cap, cx, cy, cw, 0, 1, 2, P, tx0, ty0, tx1, ty1);
            } else {   // near-plane clip, case by case (k_render_blocks: the same vertex sequences)
                const bool i0 = cw[0] >= P.near_, i1 = cw[1] >= P.near_, i2 = cw[2] >= P.near_;
                float ix[3], iy[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int k1 = (k + 1) % 3;
                    const float tt = (P.near_ - cw[k]) / (cw[k1] - cw[k]);
                    ix[k] = cx[k] + tt * (cx[k1] - cx[k]);
                    iy[k] = cy[k] + tt * (cy[k1] - cy[k]);
                }
                const float nr = P.near_;
                float ox[4] = {0.0f, 0.0f, 0.0f, 0.0f}, oy[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ow[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                int no = 0;
#define TG_V(s, k) do { ox[s] = cx[k]; oy[s] = cy[k]; ow[s] = cw[k]; } while (0)
#define TG_I(s, k) do { ox[s] = ix[k]; oy[s] = iy[k]; ow[s] = nr; } while (0)
                if (i0 & !i1 & !i2) { TG_V(0, 0); TG_I(1, 0); TG_I(2, 2); no = 3; }
                else if (!i0 & i1 & !i2) { TG_I(0, 0); TG_V(1, 1); TG_I(2, 1); no = 3; }
                else if (!i0 & !i1 & i2) { TG_I(0, 1); TG_V(1, 2); TG_I(2, 2); no = 3; }
                else if (i0 & i1 & !i2) { TG_V(0, 0); TG_V(1, 1); TG_I(2, 1); TG_I(3, 2); no = 4; }
                else if (!i0 & i1 & i2) { TG_I(0, 0); TG_V(1, 1); TG_V(2, 2); TG_I(3, 2); no = 4; }
                else if (i0 & !i1 & i2) { TG_V(0, 0); TG_I(1, 0); TG_I(2, 1); TG_V(3, 2); no = 4; }
#undef TG_V
#undef TG_I
                if (no >= 3) emit(recs, count, rec_cap, ox, oy, ow, 0, 1, 2, P, tx0, ty0, tx1, ty1);
                if (no == 4) emit(recs, count, rec_cap, ox, oy, ow, 0, 2, 3, P, tx0, ty0, tx1, ty1);
            }
        }
        __syncthreads();
        const int n = __builtin_amdgcn_readfirstlane(min(*count, rec_cap));     // <= 64

        // 1. lane-as-record
        float q_xl = 1e30f, q_xh = -1e30f, q_yl = 1e30f, q_yh = -1e30f, q_dm = 1e30f, q_x0 = 0.0f, q_y0 = 0.0f, q_d0 = 0.0f, q_A = 0.0f, q_B = 0.0f, q_mg = 1e30f;
        float q_x1 = 0.0f, q_y1 = 0.0f, q_x2 = 0.0f, q_y2 = 0.0f;
        if (lane < n) {
            const TriRec& r = recs[lane];
            q_xl = r.xmin; q_xh = r.xmax; q_yl = r.ymin; q_yh = r.ymax; q_dm = r.dmin; q_x0 = r.x0; q_y0 = r.y0; q_d0 = r.d0;
            q_x1 = r.x1; q_y1 = r.y1; q_x2 = r.x2; q_y2 = r.y2;
            const float ux = r.x1 - r.x0, uy = r.y1 - r.y0, vx = r.x2 - r.x0, vy = r.y2 - r.y0, ud = r.d1 - r.d0, vd = r.d2 - r.d0;
            const float ar = ux * vy - vx * uy;
            const float cond = ((q_xh - q_xl) * (q_yh - q_yl)) / fabsf(ar);
            if (cond < 64.0f) {
                q_A = (ud * vy - vd * uy) / ar; q_B = (ux * vd - vx * ud) / ar;
                q_mg = 2e-5f * cond;
            }
        }
        // 2. lane-as-block: which of the 64 blocks can record t change?  The ballot goes into lane t's register.
        unsigned r_lo = 0u, r_hi = 0u;                             // this lane's record: the blocks it reaches
        unsigned long long reached = 0ull;
        {
            const float X0 = (float)(rx + (lane % NBX) * BW) + 0.5f, Y0 = (float)(ry + (lane / NBX) * BH) + 0.5f;
            const float X1 = X0 + (float)(BW - 1), Y1 = Y0 + (float)(BH - 1);
#define TG_RL(v) __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), t))
            for (int t = 0; t < n; ++t) {
                const float xl = TG_RL(q_xl), xh = TG_RL(q_xh), yl = TG_RL(q_yl), yh = TG_RL(q_yh), dm = TG_RL(q_dm);
                const float x0 = TG_RL(q_x0), y0 = TG_RL(q_y0), d0 = TG_RL(q_d0), A = TG_RL(q_A), B = TG_RL(q_B), mg = TG_RL(q_mg);
                bool miss = (yh < Y0) | (yl > Y1) | (xh < X0) | (xl > X1) | (dm >= bmax_l);
                const float xa = fmaxf(X0, xl), xb = fminf(X1, xh), ya = fmaxf(Y0, yl), yb = fminf(Y1, yh);
                const float low = (d0 + A * ((A >= 0.0f ? xa : xb) - x0)) + B * ((B >= 0.0f ? ya : yb) - y0);
                miss = miss | (low - mg >= bmax_l - kGrey);
                if (kEdgeReach) miss = miss | edges_exclude_rect(x0, y0, TG_RL(q_x1), TG_RL(q_y1), TG_RL(q_x2), TG_RL(q_y2), X0, X1, Y0, Y1);
                const unsigned long long mm = __ballot(!miss);
                reached |= mm;
                if (lane == t) { r_lo = (unsigned)mm; r_hi = (unsigned)(mm >> 32); }
            }
        }
        if (drawn_p && lane == 0) *drawn_p = reached;
        // 3. blocks the previous launch drew and nothing reaches now go back to the untouched-sensor image: 16-byte words, whole 128-byte lines
        const unsigned long long restore = stale & ~reached;
        if (restore) {
            constexpr int RPI = 64 / NBX;                          // rows per instruction
            const int cbx = lane % NBX, crow0 = lane / NBX;
            for (int i = 0; i < 128 / RPI; ++i) {
                const int row = RPI * i + crow0;
                if (!((restore >> ((RPI * i) / BH * NBX)) & ((1ull << NBX) - 1ull))) continue;   // no block of this block row (uniform)
                const size_t off = (size_t)(ry + row) * P.W + (rx + cbx * BW);
                if ((restore >> ((row / BH) * NBX + cbx)) & 1ull) *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(P.tmpl + off);
            }
        }
        // 4. reached blocks, NB per round, one quad of each per lane
        const int lrow = lane >> 2, lcol = 4 * (lane & 3);
        unsigned long long left = reached;
        while (left) {
            int bq[NB];
            unsigned long long grp_any = 0ull;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                bq[j] = left ? (int)__builtin_ctzll(left) : -1;
                if (left) grp_any |= left & (~left + 1ull);
                left = left ? (left & (left - 1ull)) : 0ull;
            }
            size_t off[NB];
            int qx[NB];
            float fy[NB], z[NB][4];
            uchar4 ng[NB], bmk[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int b = bq[j] < 0 ? 0 : bq[j];
                const int y = ry + (b / NBX) * BH + lrow;
                qx[j] = rx + (b % NBX) * BW + lcol;
                fy[j] = (float)y + 0.5f;
                off[j] = (size_t)y * P.W + qx[j];
                float4 nd = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);  // no block: nothing passes the depth cull
                ng[j] = make_uchar4(0, 0, 0, 0); bmk[j] = ng[j];
                if (bq[j] >= 0) {
                    nd = *reinterpret_cast<const float4*>(nodef_dep + off[j]);
                    ng[j] = *reinterpret_cast<const uchar4*>(gray_u8 + off[j]);
                    bmk[j] = *reinterpret_cast<const uchar4*>(border + off[j]);
                }
                z[j][0] = nd.x; z[j][1] = nd.y; z[j][2] = nd.z; z[j][3] = nd.w;
            }
            for (int t = 0; t < n; ++t) {
                const unsigned long long mt = (unsigned long long)__builtin_amdgcn_readlane(r_lo, t) | ((unsigned long long)__builtin_amdgcn_readlane(r_hi, t) << 32);
                if (!(mt & grp_any)) continue;            // the record reaches none of this round's blocks
                const TriRec r = recs[t];
                const float rA = TG_RL(q_A), rB = TG_RL(q_B), rmg = TG_RL(q_mg);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (bq[j] < 0 || !((mt >> bq[j]) & 1ull)) continue;     // (uniform)
                    if (fy[j] < r.ymin || fy[j] > r.ymax) continue;
                    if ((float)qx[j] + 3.5f < r.xmin || (float)qx[j] + 0.5f > r.xmax) continue;
                    const float zmax = fmaxf(fmaxf(z[j][0], z[j][1]), fmaxf(z[j][2], z[j][3]));
                    if (r.dmin >= zmax) continue;
                    const float qlow = (r.d0 + rA * (((float)qx[j] + (rA >= 0.0f ? 0.5f : 3.5f)) - r.x0)) + rB * (fy[j] - r.y0);
                    if (qlow - rmg >= zmax) continue;
                    const float a0 = r.y2 - fy[j], a1 = r.y1 - fy[j], a2 = r.y0 - fy[j];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float fx = (float)(qx[j] + p) + 0.5f;
                        const float e0 = (r.x1 - fx) * a0 - (r.x2 - fx) * a1;
                        const float e1 = (r.x2 - fx) * a2 - (r.x0 - fx) * a0;
                        const float e2 = (r.x0 - fx) * a1 - (r.x1 - fx) * a2;
                        const bool box = (fx >= r.xmin) & (fx <= r.xmax);
                        const bool pos = (e0 >= 0.0f) & (e1 >= 0.0f) & (e2 >= 0.0f), neg = (e0 <= 0.0f) & (e1 <= 0.0f) & (e2 <= 0.0f);
                        const float s = (e0 + e1) + e2;
                        const float d = div_mid_range((e0 * r.d0 + e1 * r.d1) + e2 * r.d2, s);
                        const bool hit = box & (pos | neg) & (s != 0.0f) & (d < z[j][p]);
                        z[j][p] = hit ? d : z[j][p];
                    }
                }
            }
            float4 nd2[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                nd2[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (bq[j] >= 0) nd2[j] = *reinterpret_cast<const float4*>(nodef_dep + off[j]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (bq[j] < 0) continue;
                const float ndv[4] = {nd2[j].x, nd2[j].y, nd2[j].z, nd2[j].w};
                const uint8_t ngv[4] = {ng[j].x, ng[j].y, ng[j].z, ng[j].w}, bmv[4] = {bmk[j].x, bmk[j].y, bmk[j].z, bmk[j].w};
                uint8_t o[4] = {0, 0, 0, 0};
                if ((z[j][0] != ndv[0]) | (z[j][1] != ndv[1]) | (z[j][2] != ndv[2]) | (z[j][3] != ndv[3])) {   // an unchanged depth gives 0 below
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        float diff = z[j][p] - ndv[p];
                        if (diff >= -eps && diff <= eps) diff = 0.0f;
                        const float pen = fabsf(diff);
                        const float cl = pen < 0.0f ? 0.0f : (pen > max_pen ? max_pen : pen);
                        o[p] = (uint8_t)(div_mid_range(cl, max_pen) * 255.0f);
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (!P.turn_off_border && bmv[p] == 1) o[p] = ngv[p];
                *reinterpret_cast<uchar4*>(dst + off[j]) = make_uchar4(o[0], o[1], o[2], o[3]);
            }
        }
#undef TG_RL
    }
}

}  // namespace tg
