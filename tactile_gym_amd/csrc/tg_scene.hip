// tg_scene.hip — the envs' RGB scene camera (get_visual_obs, base_tactile_env.py:212-245) for the whole batch.
//
// Compiled with -ffp-contract=off: the raster rule below is the specification shared with oracle/minibullet.c mb_render_scene, every
// float operation is written once and evaluated in the same order on both sides, and the image is a max over 64-bit keys, so it does
// not depend on the order in which triangles are drawn.  PARITY UNPINNED against upstream (its renderer is the GL driver's / TinyRenderer
// and the checkout holds no scene image): PARITY_ASSUMPTIONS A31-A33.
//
// Mapping: one workgroup of 1024 lanes per (env, 128 x 128 tile) keeps the tile's z-buffer (8 B keys) in LDS - 128 KB of the CU's 160 KB.
// The robot's ~10^5 triangles are mostly smaller than a pixel at the reference's image sizes, so the work is triangle set-up, not fill:
// lanes stride over the triangle list (indices + attributes stream from L2, shared by every env), transform the three vertices with
// the env's eye<-frame matrix held in LDS, and test the handful of pixel centres in the triangle's bounding box with ds_max_u64.
// Triangles whose box is large (plane, table top) are queued in LDS and then filled by the whole workgroup, pixels across lanes.
// Most of the robot is outside the camera's view: triangles are sorted on the host into chunks of <= 64 neighbours (one frame each)
// with a bounding sphere, the workgroup first tests the spheres against the tile's frustum and only visible chunks are set up,
// one chunk per wavefront, one triangle per lane.
#include "tg_scene.h"

#include <math.h>
#include <stdio.h>

#include <algorithm>
#include <mutex>
#include <numeric>

namespace tg {

namespace {

constexpr int kThreads = 1024;
constexpr int kBigArea = 64;       // bounding boxes above this many pixels are filled by a whole wavefront, pixels across lanes
constexpr int kBigCap = 2048;      // queue of those, in LDS (shrunk by lds_layout when the chunk list needs the room)
constexpr int kHugeArea = 4096;    // ... and above this many by the whole workgroup (ground plane, table top)
constexpr int kHugeCap = 64;
constexpr int kMaxFrames = 16;
constexpr int kChunk = 64;
constexpr int kMaxChunks = 8192;   // visible-chunk list in LDS (u16 entries, sized by the scene's chunk count)
constexpr int kWaveQ = 256;        // per wavefront: two queues (boxes of <= kSmallArea pixels / larger) of triangles that passed the shared-vertex tests,
                                   // each < 64 carried over + <= 64 new
constexpr int kSmallArea = 4;
constexpr int kMaxSpheres = 16;   // translucent spheres per env (SceneParams::spheres): object_push's trajectory markers are the most (<= TG_MAX_TRAJ_POINTS)

#ifdef TG_SCENE_STATS
__device__ unsigned long long g_stats[24];   // 0 workgroups, 1 visible chunks, 2 queued survivors, 3 set-ups that drew, 4 big, 5 huge, 6 big pixels, 7 huge pixels
#define TG_STAT(i, v) atomicAdd(&g_stats[i], (unsigned long long)(v))
#else
#define TG_STAT(i, v)
#endif

// single-instruction forms (the operands here are never NaN, so fminf / fmaxf's canonicalising pre-pass is not needed; min, max and clamp are exact
// operations: the values are those of the nested fminf / fmaxf in setup_verts)
__device__ __forceinline__ float min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi)); return r; }
__device__ __forceinline__ float lane_read(float v, int byte_addr) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v))); }

struct TriSetup {
    float a0, b0, c0, a1, b1, c1, a2, b2, c2, sg, rdet;
    uint32_t rgb;
    int x0, x1, y0, y1;            // pixel box (inclusive), empty if x0 > x1
};

// Everything about a triangle that does not depend on the pixel: vertices (frame coordinates), eye<-frame transform M, base colour.
// Returns false when nothing can be drawn.
__device__ __forceinline__ bool setup_verts(const SceneParams& P, const float* __restrict__ M, const float (&vxs)[3], const float (&vys)[3],
                                            const float (&vzs)[3], uint32_t attr, const int (&tile)[4], TriSetup& S) {
    float ex[3], ey[3], ez[3], w[3], X[3], Y[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float vx = vxs[k], vy = vys[k], vz = vzs[k];
        ex[k] = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
        ey[k] = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
        ez[k] = ((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11];
        w[k] = -ez[k];
        X[k] = P.kx * ex[k] + P.hw * w[k];
        Y[k] = P.hh * w[k] - P.ky * ey[k];
    }
    if (w[0] < P.near_ && w[1] < P.near_ && w[2] < P.near_) return false;
    if (w[0] > P.far_ && w[1] > P.far_ && w[2] > P.far_) return false;
    S.x0 = 0; S.x1 = P.W - 1; S.y0 = 0; S.y1 = P.H - 1;
    if (w[0] >= P.near_ && w[1] >= P.near_ && w[2] >= P.near_) {   // most triangles end here: no pixel centre in their box
        const float r0 = 1.0f / w[0], r1 = 1.0f / w[1], r2 = 1.0f / w[2];
        const float sx0 = X[0] * r0, sx1 = X[1] * r1, sx2 = X[2] * r2, sy0 = Y[0] * r0, sy1 = Y[1] * r1, sy2 = Y[2] * r2;
        float minx = fminf(sx0, fminf(sx1, sx2)), maxx = fmaxf(sx0, fmaxf(sx1, sx2)), miny = fminf(sy0, fminf(sy1, sy2)), maxy = fmaxf(sy0, fmaxf(sy1, sy2));
        minx = fminf(fmaxf(minx, -1.0f), (float)P.W + 1.0f); maxx = fminf(fmaxf(maxx, -1.0f), (float)P.W + 1.0f);
        miny = fminf(fmaxf(miny, -1.0f), (float)P.H + 1.0f); maxy = fminf(fmaxf(maxy, -1.0f), (float)P.H + 1.0f);
        S.x0 = max(0, (int)ceilf(minx - 0.515625f)); S.x1 = min(P.W - 1, (int)floorf(maxx - 0.484375f));
        S.y0 = max(0, (int)ceilf(miny - 0.515625f)); S.y1 = min(P.H - 1, (int)floorf(maxy - 0.484375f));
        if (S.x0 > S.x1 || S.y0 > S.y1) return false;
    }
    if (S.x0 > tile[2] || S.x1 < tile[0] || S.y0 > tile[3] || S.y1 < tile[1]) return false;
    S.a0 = Y[1] * w[2] - Y[2] * w[1]; S.b0 = w[1] * X[2] - w[2] * X[1]; S.c0 = X[1] * Y[2] - X[2] * Y[1];
    S.a1 = Y[2] * w[0] - Y[0] * w[2]; S.b1 = w[2] * X[0] - w[0] * X[2]; S.c1 = X[2] * Y[0] - X[0] * Y[2];
    S.a2 = Y[0] * w[1] - Y[1] * w[0]; S.b2 = w[0] * X[1] - w[1] * X[0]; S.c2 = X[0] * Y[1] - X[1] * Y[0];
    const float det = (S.c0 * w[0] + S.c1 * w[1]) + S.c2 * w[2];
    if (det == 0.0f) return false;
    S.sg = det > 0.0f ? 1.0f : -1.0f;
    S.rdet = 1.0f / det;
    // flat shade: n = (e1 - e0) x (e2 - e0) turned towards the eye, 0.6 ambient + 0.35 diffuse [A32]
    const float ux = ex[1] - ex[0], uy = ey[1] - ey[0], uz = ez[1] - ez[0], vx = ex[2] - ex[0], vy = ey[2] - ey[0], vz = ez[2] - ez[0];
    const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const float nn = sqrtf((nx * nx + ny * ny) + nz * nz);
    float ndl = 0.0f;
    if (nn > 0.0f) {
        ndl = ((nx * P.light_eye[0] + ny * P.light_eye[1]) + nz * P.light_eye[2]) / nn;
        if ((nx * ex[0] + ny * ey[0]) + nz * ez[0] > 0.0f) ndl = -ndl;
        if (ndl < 0.0f) ndl = 0.0f;
    }
    const float inten = 0.6f + 0.35f * ndl;
    const uint32_t r = (uint32_t)((float)((attr >> 16) & 255u) * inten + 0.5f), g = (uint32_t)((float)((attr >> 8) & 255u) * inten + 0.5f),
                   b = (uint32_t)((float)(attr & 255u) * inten + 0.5f);
    S.rgb = (r << 16) | (g << 8) | b;
    return true;
}
// triangle t of the shared indexed set (t < n_tris), or triangle t - n_tris of this env's heightfield (2 per grid cell)
__device__ __forceinline__ bool setup_tri(const SceneParams& P, const float* __restrict__ sxf, int env, int t, const int (&tile)[4], TriSetup& S) {
    float vx[3], vy[3], vz[3];
    if (t < P.n_tris) {
        const uint32_t attr = P.tri_attr[t];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = P.tris[3 * t + k];
            vx[k] = P.verts[3 * i + 0]; vy[k] = P.verts[3 * i + 1]; vz[k] = P.verts[3 * i + 2];
        }
        return setup_verts(P, sxf + 12 * (attr >> 24), vx, vy, vz, attr, tile, S);
    }
    const int h = t - P.n_tris, cell = h >> 1, half = h & 1;
    const int ci = cell % (P.hf_rows - 1), cj = cell / (P.hf_rows - 1);
    const size_t hidx = P.hf_sel != nullptr ? (size_t)(P.hf_sel[env] & 3) * P.hf_n + env : (size_t)env;
    const double* H = P.hf_heights + hidx * P.hf_rows * P.hf_cols;
    const float zoff = P.hf_zoff[hidx], cx = 0.5f * (float)(P.hf_rows - 1), cy = 0.5f * (float)(P.hf_cols - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // half 0: (i,j),(i,j+1),(i+1,j)   half 1: (i+1,j),(i,j+1),(i+1,j+1)
        const int di = half == 0 ? (k == 2) : (k != 1), dj = half == 0 ? (k == 1) : (k != 0);
        const int vi = ci + di, vj = cj + dj;
        vx[k] = ((float)vi - cx) * P.hf_scale;
        vy[k] = ((float)vj - cy) * P.hf_scale;
        vz[k] = (float)H[(size_t)vj * P.hf_rows + vi] - zoff;
    }
    return setup_verts(P, sxf + 12 * (P.n_frames - 1), vx, vy, vz, P.hf_rgb, tile, S);
}

__device__ __forceinline__ void shade_pixel(const SceneParams& P, const TriSetup& S, int px, int py, unsigned long long* zb, int tx0, int ty0, int tw) {
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float e0 = (S.a0 * fx + S.b0 * fy) + S.c0, e1 = (S.a1 * fx + S.b1 * fy) + S.c1, e2 = (S.a2 * fx + S.b2 * fy) + S.c2;
    if (!(S.sg * e0 >= 0.0f && S.sg * e1 >= 0.0f && S.sg * e2 >= 0.0f)) return;
    const float iw = ((e0 + e1) + e2) * S.rdet;
    if (!(iw >= P.inv_far && iw <= P.inv_near)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(iw) << 32) | S.rgb;
    atomicMax(&zb[(py - ty0) * tw + (px - tx0)], key);
}

__global__ __launch_bounds__(kThreads) void k_scene(SceneParams P, const float* __restrict__ xf, const uint8_t* __restrict__ mask,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ save_prev, int tw, int th, int big_cap,
                                                    unsigned long long* __restrict__ static_out) {
    // static_out != nullptr: the one-off pass over the world frame (frame 0: ground plane, table - the same for every env and step, the
    // camera is fixed): its keys are stored per tile and every later draw starts its z-buffer from them instead of drawing those triangles again
    // (the image is a max over keys, so the split changes nothing).
    const bool static_pass = static_out != nullptr;
    const int env = blockIdx.y;
    if (mask != nullptr && mask[env] == 0) return;
    extern __shared__ unsigned long long zb[];                     // [th][tw] keys, then the queue of large triangles
    __shared__ float sxf[kMaxFrames * 12];
    __shared__ int big_n, vis_n, huge_n;
    __shared__ int huge[kHugeCap];
    int* big = reinterpret_cast<int*>(zb + tw * th);
    volatile int* wq = big + big_cap + (threadIdx.x >> 6) * kWaveQ;   // this wavefront's queue
    uint16_t* vis_list = reinterpret_cast<uint16_t*>(big + big_cap + (kThreads / 64) * kWaveQ);
    const int tiles_x = P.W / tw;
    const int tx0 = (blockIdx.x % tiles_x) * tw, ty0 = (blockIdx.x / tiles_x) * th;
    const int tid = threadIdx.x;
    uint8_t* img = out + (size_t)env * P.W * P.H * 3;
    if (save_prev != nullptr && !static_pass) {
        uint8_t* dst = save_prev + (size_t)env * P.W * P.H * 3;
        for (int p = tid; p < tw * th; p += kThreads) {
            const size_t o = ((size_t)(ty0 + p / tw) * P.W + (tx0 + p % tw)) * 3;
            dst[o] = img[o]; dst[o + 1] = img[o + 1]; dst[o + 2] = img[o + 2];
        }
    }
    if (static_pass || P.static_keys == nullptr) { for (int p = tid; p < tw * th; p += kThreads) zb[p] = 0ull; }
    else { const unsigned long long* sk = P.static_keys + (size_t)blockIdx.x * tw * th; for (int p = tid; p < tw * th; p += kThreads) zb[p] = sk[p]; }
    const bool skip_world = !static_pass && P.static_keys != nullptr;
    for (int p = tid; p < P.n_frames * 12; p += kThreads) sxf[p] = xf[(size_t)env * P.n_frames * 12 + p];
    if (tid == 0) { big_n = 0; vis_n = 0; huge_n = 0; }
    __syncthreads();
    const int bx1 = tx0 + tw - 1, by1 = ty0 + th - 1;
    const int tile[4] = {tx0, ty0, bx1, by1};
    // chunk spheres against the tile's frustum (eye space; conservative: a chunk that fails cannot touch a pixel centre of the tile).
    // Side planes pass through the eye: pixel column c is the plane kx x + (hw - c) w = 0, row r is (hh - r) w - ky y = 0.
    {
        const float nl = sqrtf(P.kx * P.kx + (P.hw - (float)tx0) * (P.hw - (float)tx0)), nr = sqrtf(P.kx * P.kx + (P.hw - (float)(bx1 + 1)) * (P.hw - (float)(bx1 + 1)));
        const float nt = sqrtf(P.ky * P.ky + (P.hh - (float)ty0) * (P.hh - (float)ty0)), nb_ = sqrtf(P.ky * P.ky + (P.hh - (float)(by1 + 1)) * (P.hh - (float)(by1 + 1)));
        for (int ci = tid; ci < P.n_chunks; ci += kThreads) {
            const SceneChunk ch = P.chunks[ci];
            if (static_pass ? ch.frame != 0 : (skip_world && ch.frame == 0)) continue;
            const float* M = sxf + 12 * ch.frame;
            const float x = M[0] * ch.cx + M[1] * ch.cy + M[2] * ch.cz + M[9], y = M[3] * ch.cx + M[4] * ch.cy + M[5] * ch.cz + M[10];
            const float w = -(M[6] * ch.cx + M[7] * ch.cy + M[8] * ch.cz + M[11]);
            const float sc = sqrtf(M[0] * M[0] + M[3] * M[3] + M[6] * M[6]);                 // object_roll: the marble's frame is scaled
            const float r = ch.r * sc * 1.001f + 1e-6f;
            bool vis = w + r >= P.near_ && w - r <= P.far_;
            vis = vis && (P.kx * x + (P.hw - (float)tx0) * w) >= -r * nl && ((float)(bx1 + 1) - P.hw) * w - P.kx * x >= -r * nr;
            vis = vis && ((P.hh - (float)ty0) * w - P.ky * y) >= -r * nt && (P.ky * y - (P.hh - (float)(by1 + 1)) * w) >= -r * nb_;
            if (vis) { const int slot = atomicAdd(&vis_n, 1); if (slot < kMaxChunks) vis_list[slot] = (uint16_t)ci; }
        }
    }
    __syncthreads();
    const int nvis = min(vis_n, kMaxChunks);
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) { TG_STAT(0, 1); TG_STAT(1, nvis); }
    // One triangle of the lane: full set-up; boxes above kBigArea / kHugeArea pixels are queued for a wavefront / the workgroup, the rest
    // (a handful of pixel centres) is tested here.
    auto draw = [&](int t) {
        TriSetup S;
        TG_STAT(2, 1);
        if (!setup_tri(P, sxf, env, t, tile, S)) return;
        TG_STAT(3, 1);
        const int x0 = max(S.x0, tx0), x1 = min(S.x1, bx1), y0 = max(S.y0, ty0), y1 = min(S.y1, by1);
        const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
        if (area > kHugeArea) {
            const int slot = atomicAdd(&huge_n, 1);
            if (slot < kHugeCap) { huge[slot] = t; return; }
        }
        if (area > kBigArea) {
            const int slot = atomicAdd(&big_n, 1);
            if (slot < big_cap) { big[slot] = t; return; }
        }
#ifdef TG_SCENE_STATS
        { int b = 0; while ((1 << b) < area) ++b; TG_STAT(8 + b, 1); }      // 8..14: area <= 1, 2, 4, 8, 16, 32, 64
        { int mx = area; for (int o = 32; o; o >>= 1) mx = max(mx, __shfl_xor(mx, o)); if (area == mx) TG_STAT(16, 1); TG_STAT(17, area); if ((threadIdx.x & 63) == __ffsll(__ballot(1)) - 1) TG_STAT(18, mx); }
#endif
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) shade_pixel(P, S, px, py, zb, tx0, ty0, tw);
    };
    // Visible chunks, one per wavefront and pass.  Most of the robot's triangles are smaller than a pixel and end at "no pixel centre in
    // the box" (setup_verts): that test is run on vertices transformed ONCE per chunk (a lane per distinct vertex, the triangle lanes pick
    // their corners with ds_bpermute), with the very expressions of setup_verts, so its outcome is the same bit for bit; the few triangles
    // that pass are compacted into the wavefront's queue and set up in full on dense lanes.
    // Two queues by box size: a dense pass lasts as long as its largest box, so one-pixel boxes (60 % of the survivors) do not wait for 8 x 8 ones.
    int qn = 0, qn2 = 0;                                          // wave-uniform fills of wq[0 .. 127] (small boxes), wq[128 .. 255]
    for (int vi = wave; vi < nvis; vi += kThreads / 64) {
        const SceneChunk ch = P.chunks[vis_list[vi]];
        const float* M = sxf + 12 * ch.frame;
        float vw = 0.0f, vsx = 0.0f, vsy = 0.0f;
        if (lane < ch.vcount) {
            const float* v = P.verts + 3 * (size_t)(ch.vstart + lane);
            const float vx = v[0], vy = v[1], vz = v[2];
            const float ex = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
            const float ey = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
            const float ez = ((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11];
            vw = -ez;
            const float X = P.kx * ex + P.hw * vw, Y = P.hh * vw - P.ky * ey;
            const float r = 1.0f / vw;                             // used only by triangles whose three corners are beyond the near plane
            vsx = X * r; vsy = Y * r;
        }
        const uint32_t loc = lane < ch.count ? P.tri_local[ch.start + lane] : 0u;
        const int i0 = (int)((loc & 255u) << 2), i1 = (int)((loc >> 6) & 1020u), i2 = (int)((loc >> 14) & 1020u);   // ds_bpermute byte addresses
        const float w0 = lane_read(vw, i0), w1 = lane_read(vw, i1), w2 = lane_read(vw, i2);
        const float sx0 = lane_read(vsx, i0), sx1 = lane_read(vsx, i1), sx2 = lane_read(vsx, i2);
        const float sy0 = lane_read(vsy, i0), sy1 = lane_read(vsy, i1), sy2 = lane_read(vsy, i2);
        const float wmin = min3(w0, w1, w2), wmax = max3(w0, w1, w2);
        bool alive = lane < ch.count && !(wmax < P.near_) && !(wmin > P.far_);      // all three behind the near / beyond the far plane
        bool small = false;
        if (wmin >= P.near_) {
            const float fw = (float)P.W + 1.0f, fh = (float)P.H + 1.0f;
            const float minx = clampf(min3(sx0, sx1, sx2), -1.0f, fw), maxx = clampf(max3(sx0, sx1, sx2), -1.0f, fw);
            const float miny = clampf(min3(sy0, sy1, sy2), -1.0f, fh), maxy = clampf(max3(sy0, sy1, sy2), -1.0f, fh);
            const int bx0 = max(0, (int)ceilf(minx - 0.515625f)), bx1_ = min(P.W - 1, (int)floorf(maxx - 0.484375f));
            const int by0 = max(0, (int)ceilf(miny - 0.515625f)), by1_ = min(P.H - 1, (int)floorf(maxy - 0.484375f));
            alive = alive && bx0 <= bx1_ && by0 <= by1_ && !(bx0 > tile[2] || bx1_ < tile[0] || by0 > tile[3] || by1_ < tile[1]);
            small = (bx1_ - bx0 + 1) * (by1_ - by0 + 1) <= kSmallArea;
        }
        const unsigned long long m = __ballot(alive);
        if (m == 0ull) continue;
        const unsigned long long m1 = __ballot(alive && small), m2 = m & ~m1;
        if (alive && small) wq[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = ch.start + lane;
        if (alive && !small) wq[128 + qn2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u))] = ch.start + lane;
        qn += __popcll(m1); qn2 += __popcll(m2);
        if (qn >= 64) {
            qn -= 64;
            draw(wq[qn + lane]);
        }
        if (qn2 >= 64) {
            qn2 -= 64;
            draw(wq[128 + qn2 + lane]);
        }
    }
    if (lane < qn) draw(wq[lane]);
    if (lane < qn2) draw(wq[128 + lane]);
    if (P.hf_heights != nullptr && !static_pass) {                // this env's heightfield: two triangles per grid cell, a lane per triangle
        const int n_hf = 2 * (P.hf_rows - 1) * (P.hf_cols - 1);
        for (int h = tid; h < n_hf; h += kThreads) draw(P.n_tris + h);
    }
    __syncthreads();
    // Queued large triangles.  Their set-up runs on dense lanes (a lane per queued triangle), then one triangle at a time is handed to the
    // whole wavefront (workgroup) through v_readlane - its coefficients become scalar operands - and the pixels of its box go across the lanes.
    auto bcast_setup = [](const TriSetup& S, int src, TriSetup& B) {
        auto rf = [src](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
        B.a0 = rf(S.a0); B.b0 = rf(S.b0); B.c0 = rf(S.c0); B.a1 = rf(S.a1); B.b1 = rf(S.b1); B.c1 = rf(S.c1);
        B.a2 = rf(S.a2); B.b2 = rf(S.b2); B.c2 = rf(S.c2); B.sg = rf(S.sg); B.rdet = rf(S.rdet);
        B.rgb = (uint32_t)__builtin_amdgcn_readlane((int)S.rgb, src);
        B.x0 = __builtin_amdgcn_readlane(S.x0, src); B.x1 = __builtin_amdgcn_readlane(S.x1, src);
        B.y0 = __builtin_amdgcn_readlane(S.y0, src); B.y1 = __builtin_amdgcn_readlane(S.y1, src);
    };
    const int nb = min(big_n, big_cap);
    const int per = min(64, (nb + kThreads / 64 - 1) / (kThreads / 64));      // queued triangles per wavefront and pass
    for (int base = wave * per; base < nb; base += (kThreads / 64) * per) {
        TriSetup S;
        bool ok = lane < per && base + lane < nb;
        if (ok) ok = setup_tri(P, sxf, env, big[base + lane], tile, S);
        if (ok) { S.x0 = max(S.x0, tx0); S.x1 = min(S.x1, bx1); S.y0 = max(S.y0, ty0); S.y1 = min(S.y1, by1); }
        unsigned long long m = __ballot(ok);
        while (m != 0ull) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            m &= m - 1ull;
            TriSetup B;
            bcast_setup(S, src, B);
            const int bw = B.x1 - B.x0 + 1, np = bw * (B.y1 - B.y0 + 1);
            if (lane == 0) { TG_STAT(4, 1); TG_STAT(6, np); }
            const float rbw = 1.0f / (float)bw;       // row = floor((p + 0.5) / bw): exact for p < 2^14, bw <= 128 (the fraction is >= 0.5 / 128 from an integer)
            for (int p = lane; p < np; p += 64) { const int row = (int)(((float)p + 0.5f) * rbw); shade_pixel(P, B, B.x0 + (p - row * bw), B.y0 + row, zb, tx0, ty0, tw); }
        }
    }
    const int nh = min(huge_n, kHugeCap);
    {
        TriSetup S;
        bool ok = lane < nh;                                      // every wavefront sets the (<= 64) huge triangles up for itself
        if (ok) ok = setup_tri(P, sxf, env, huge[lane], tile, S);
        if (ok) { S.x0 = max(S.x0, tx0); S.x1 = min(S.x1, bx1); S.y0 = max(S.y0, ty0); S.y1 = min(S.y1, by1); }
        unsigned long long m = __ballot(ok);
        while (m != 0ull) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            m &= m - 1ull;
            TriSetup B;
            bcast_setup(S, src, B);
            const int bw = B.x1 - B.x0 + 1, np = bw * (B.y1 - B.y0 + 1);
            if (tid == 0) { TG_STAT(5, 1); TG_STAT(7, np); }
            const float rbw = 1.0f / (float)bw;
            for (int p = tid; p < np; p += kThreads) { const int row = (int)(((float)p + 0.5f) * rbw); shade_pixel(P, B, B.x0 + (p - row * bw), B.y0 + row, zb, tx0, ty0, tw); }
        }
    }
    __syncthreads();
    if (static_pass) {
        for (int p = tid; p < tw * th; p += kThreads) static_out[(size_t)blockIdx.x * tw * th + p] = zb[p];
        return;
    }
    const int ns = P.spheres != nullptr ? (P.n_spheres < kMaxSpheres ? P.n_spheres : kMaxSpheres) : 0;
    float* ssp = reinterpret_cast<float*>(big);                     // the queues are done with: this env's spheres take their place
    if (ns > 0) {
        for (int p = tid; p < ns * 8; p += kThreads) ssp[p] = P.spheres[((size_t)env * P.n_spheres) * 8 + p];
        __syncthreads();
    }
    for (int p = tid; p < tw * th; p += kThreads) {
        const unsigned long long k = zb[p];
        const int px = tx0 + p % tw, py = ty0 + p / tw;
        const size_t o = ((size_t)py * P.W + px) * 3;
        uint32_t c0 = k ? (uint32_t)((k >> 16) & 255u) : P.background[0], c1 = k ? (uint32_t)((k >> 8) & 255u) : P.background[1], c2 = k ? (uint32_t)(k & 255u) : P.background[2];
        if (ns > 0) {                                               // translucent spheres, in list order (tg_scene.h; oracle: mb_blend_spheres)
            const float iw_o = k ? __uint_as_float((uint32_t)(k >> 32)) : 0.0f;
            const float dx = (((float)px + 0.5f) - P.hw) / P.kx, dy = (P.hh - ((float)py + 0.5f)) / P.ky;
            const float A = (dx * dx + dy * dy) + 1.0f;
            for (int s = 0; s < ns; ++s) {
                const float* S = ssp + 8 * s;
                const float alpha = S[7];
                if (!(alpha > 0.0f)) continue;
                const float cx = S[0], cy = S[1], cz = S[2], r = S[3];
                const float B = (dx * cx + dy * cy) - cz;
                const float Cc = ((cx * cx + cy * cy) + cz * cz) - r * r;
                const float disc = B * B - A * Cc;
                if (!(disc >= 0.0f)) continue;
                const float w = (B - sqrtf(disc)) / A;
                if (!(w >= P.near_ && w <= P.far_)) continue;
                const float iw = 1.0f / w;
                if (!(iw > iw_o)) continue;
                const float nx = (w * dx - cx) / r, ny = (w * dy - cy) / r, nz = (-w - cz) / r;
                float ndl = (nx * P.light_eye[0] + ny * P.light_eye[1]) + nz * P.light_eye[2];
                if (!(ndl > 0.0f)) ndl = 0.0f;
                const float inten = 0.6f + 0.35f * ndl;
                const float s0 = (float)(uint32_t)(S[4] * inten + 0.5f), s1 = (float)(uint32_t)(S[5] * inten + 0.5f), s2 = (float)(uint32_t)(S[6] * inten + 0.5f);
                c0 = (uint32_t)((alpha * s0 + (1.0f - alpha) * (float)c0) + 0.5f) & 255u;
                c1 = (uint32_t)((alpha * s1 + (1.0f - alpha) * (float)c1) + 0.5f) & 255u;
                c2 = (uint32_t)((alpha * s2 + (1.0f - alpha) * (float)c2) + 0.5f) & 255u;
            }
        }
        img[o + 0] = (uint8_t)c0;
        img[o + 1] = (uint8_t)c1;
        img[o + 2] = (uint8_t)c2;
    }
}

// Dynamic LDS: z keys | big-triangle queue | the wavefronts' queues | visible-chunk list.  The CU has 160 KB; kLdsStatic covers the
// kernel's static arrays.  The big-triangle queue gives way when a scene has many chunks (an overflowing entry is drawn by its lane).
constexpr int kLdsBudget = 160 * 1024, kLdsStatic = 2048;
struct LdsLayout { int big_cap; size_t bytes; };
LdsLayout lds_layout(int tw, int th, int n_chunks) {
    const size_t fixed = (size_t)tw * th * 8 + (size_t)(kThreads / 64) * kWaveQ * 4 + (((size_t)n_chunks * 2 + 7) & ~(size_t)7);
    const long room = (long)kLdsBudget - kLdsStatic - (long)fixed;
    LdsLayout L;
    L.big_cap = (int)std::min<long>(kBigCap, room / 4);
    L.bytes = fixed + (size_t)std::max(L.big_cap, 0) * 4;
    return L;
}

}  // namespace

SceneParams make_scene_params(int W, int H, double fov_deg, double near_, double far_) {
    SceneParams P{};
    P.W = W; P.H = H;
    const double ys = 1.0 / tan(0.5 * fov_deg * (3.14159265358979323846 / 180.0));
    P.kx = (float)(ys * 0.5 * H); P.ky = (float)(ys * 0.5 * H);   // xScale = yScale / aspect with aspect = W / H
    P.hw = 0.5f * (float)W; P.hh = 0.5f * (float)H;
    P.near_ = (float)near_; P.far_ = (float)far_;
    P.inv_near = 1.0f / P.near_; P.inv_far = 1.0f / P.far_;
    return P;
}

void build_scene_chunks(const float* verts, int32_t* tris, uint32_t* attr, int n_tris, std::vector<SceneChunk>& chunks, std::vector<float>& cverts,
                        std::vector<uint32_t>& tri_local) {
    // per frame: bounding box of the centroids -> 10 bits per axis Morton code; stable sort by (frame, code)
    float lo[kMaxFrames][3], hi[kMaxFrames][3];
    for (int f = 0; f < kMaxFrames; ++f) for (int k = 0; k < 3; ++k) { lo[f][k] = 3.4e38f; hi[f][k] = -3.4e38f; }
    std::vector<float> cen((size_t)n_tris * 3);
    for (int t = 0; t < n_tris; ++t) {
        const int f = (int)(attr[t] >> 24);
        for (int k = 0; k < 3; ++k) {
            const float c = (verts[3 * tris[3 * t] + k] + verts[3 * tris[3 * t + 1] + k] + verts[3 * tris[3 * t + 2] + k]) / 3.0f;
            cen[(size_t)3 * t + k] = c;
            lo[f][k] = std::min(lo[f][k], c); hi[f][k] = std::max(hi[f][k], c);
        }
    }
    auto spread = [](uint32_t v) { v &= 1023u; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; return (v | (v << 2)) & 0x09249249u; };
    std::vector<uint64_t> key(n_tris);
    for (int t = 0; t < n_tris; ++t) {
        const int f = (int)(attr[t] >> 24);
        // the world frame holds a 200 m plane next to centimetre parts: quantise on a cube root scale there would be overkill; the plane's
        // two triangles simply form chunks of their own (huge radius, always visible)
        uint32_t code = 0;
        for (int k = 0; k < 3; ++k) {
            const float ext = hi[f][k] - lo[f][k];
            const uint32_t q = ext > 0.0f ? (uint32_t)std::min(1023.0f, (cen[(size_t)3 * t + k] - lo[f][k]) / ext * 1023.0f) : 0u;
            code |= spread(q) << k;
        }
        key[t] = ((uint64_t)f << 32) | code;
    }
    std::vector<int> order(n_tris);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    std::vector<int32_t> t2((size_t)n_tris * 3);
    std::vector<uint32_t> a2(n_tris);
    for (int i = 0; i < n_tris; ++i) {
        for (int k = 0; k < 3; ++k) t2[(size_t)3 * i + k] = tris[(size_t)3 * order[i] + k];
        a2[i] = attr[order[i]];
    }
    std::copy(a2.begin(), a2.end(), attr);
    chunks.clear(); cverts.clear(); tri_local.assign(n_tris, 0u);
    // a chunk ends after kChunk triangles or kChunk distinct vertices, at a frame change, or when one more triangle would make its sphere
    // much larger than the rest (keeps the ground plane and the table slabs from poisoning a chunk of small parts)
    int start = 0;
    std::vector<int32_t> local;                                   // the chunk's distinct vertices (indices into verts), in order of first use
    while (start < n_tris) {
        const int f = (int)(attr[start] >> 24);
        float blo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, bhi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        int count = 0;
        local.clear();
        while (start + count < n_tris && count < kChunk && (int)(attr[start + count] >> 24) == f) {
            const int32_t* tv = &t2[(size_t)3 * (start + count)];
            float nlo[3], nhi[3];
            for (int k = 0; k < 3; ++k) { nlo[k] = blo[k]; nhi[k] = bhi[k]; }
            int fresh = 0;
            for (int j = 0; j < 3; ++j) {
                for (int k = 0; k < 3; ++k) {
                    const float v = verts[3 * tv[j] + k];
                    nlo[k] = std::min(nlo[k], v); nhi[k] = std::max(nhi[k], v);
                }
                bool seen = std::find(local.begin(), local.end(), tv[j]) != local.end();
                for (int j2 = 0; j2 < j; ++j2) seen = seen || tv[j2] == tv[j];
                fresh += seen ? 0 : 1;
            }
            if ((int)local.size() + fresh > kChunk) break;
            const float dn = std::max({nhi[0] - nlo[0], nhi[1] - nlo[1], nhi[2] - nlo[2]}), d0 = std::max({bhi[0] - blo[0], bhi[1] - blo[1], bhi[2] - blo[2]});
            if (count >= 8 && dn > 4.0f * d0 && dn > 0.02f) break;
            for (int k = 0; k < 3; ++k) { blo[k] = nlo[k]; bhi[k] = nhi[k]; }
            uint32_t loc = 0;
            for (int j = 0; j < 3; ++j) {
                auto it = std::find(local.begin(), local.end(), tv[j]);
                if (it == local.end()) { local.push_back(tv[j]); it = local.end() - 1; }
                loc |= (uint32_t)(it - local.begin()) << (8 * j);
            }
            tri_local[start + count] = loc;
            ++count;
        }
        SceneChunk ch{};
        ch.cx = 0.5f * (blo[0] + bhi[0]); ch.cy = 0.5f * (blo[1] + bhi[1]); ch.cz = 0.5f * (blo[2] + bhi[2]);
        float r2 = 0.0f;
        for (int32_t vi : local) {
            const float* v = verts + 3 * vi;
            const float dx = v[0] - ch.cx, dy = v[1] - ch.cy, dz = v[2] - ch.cz;
            r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
        }
        ch.r = sqrtf(r2) * 1.0001f;
        ch.start = start; ch.count = (uint16_t)count; ch.frame = f;
        ch.vstart = (int)(cverts.size() / 3); ch.vcount = (uint16_t)local.size();
        for (int32_t vi : local) for (int k = 0; k < 3; ++k) cverts.push_back(verts[3 * vi + k]);
        for (int i = 0; i < count; ++i)
            for (int j = 0; j < 3; ++j) tris[(size_t)3 * (start + i) + j] = ch.vstart + (int32_t)((tri_local[start + i] >> (8 * j)) & 255u);
        chunks.push_back(ch);
        start += count;
    }
}

int scene_prepare(const SceneParams& P) {
    const int tw = P.W < 128 ? P.W : 128, th = P.H < 128 ? P.H : 128;
    const LdsLayout L = lds_layout(tw, th, P.n_chunks);
    if (P.n_chunks > kMaxChunks || L.big_cap < 64) return -1;
    // the attribute belongs to the function (per device), not to a context: another context of this process may launch k_scene with a larger
    // tile, so the limit is only ever raised
    static std::mutex mu;
    static int raised[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && (int)L.bytes <= raised[dev]) return 0;
    const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scene), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes);
    if (rc == 0 && dev >= 0 && dev < 64) raised[dev] = (int)L.bytes;
    return rc;
}

#ifdef TG_SCENE_STATS
void scene_debug_stats() {
    unsigned long long h[24]; if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stats), sizeof h) != hipSuccess || h[0] == 0) return;
    fprintf(stderr, "k_scene per workgroup: visible chunks %.1f, queued %.1f, drew %.1f, big %.1f (%.0f px), huge %.1f (%.0f px)\n", (double)h[1] / h[0],
            (double)h[2] / h[0], (double)h[3] / h[0], (double)h[4] / h[0], (double)h[6] / h[0], (double)h[5] / h[0], (double)h[7] / h[0]);
    fprintf(stderr, "  small-path box areas <=1 %.0f <=2 %.0f <=4 %.0f <=8 %.0f <=16 %.0f <=32 %.0f <=64 %.0f; sum of areas %.0f, sum over passes of the max area %.0f\n", (double)h[8] / h[0],
            (double)h[9] / h[0], (double)h[10] / h[0], (double)h[11] / h[0], (double)h[12] / h[0], (double)h[13] / h[0], (double)h[14] / h[0], (double)h[17] / h[0], (double)h[18] / h[0]); }
#else
void scene_debug_stats() {}
#endif
void launch_scene(const SceneParams& P, const float* xf, int n_envs, const uint8_t* mask, uint8_t* out, uint8_t* save_prev, hipStream_t stream) {
    const int tw = P.W < 128 ? P.W : 128, th = P.H < 128 ? P.H : 128;
    dim3 grid((P.W / tw) * (P.H / th), n_envs);
    const LdsLayout L = lds_layout(tw, th, P.n_chunks);
    hipLaunchKernelGGL(k_scene, grid, dim3(kThreads), L.bytes, stream, P, xf, mask, out, save_prev, tw, th, L.big_cap, (unsigned long long*)nullptr);
}

void launch_scene_static(const SceneParams& P, const float* xf_env0, unsigned long long* static_keys, hipStream_t stream) {
    const int tw = P.W < 128 ? P.W : 128, th = P.H < 128 ? P.H : 128;
    dim3 grid((P.W / tw) * (P.H / th), 1);
    const LdsLayout L = lds_layout(tw, th, P.n_chunks);
    SceneParams Q = P;
    Q.static_keys = nullptr;
    hipLaunchKernelGGL(k_scene, grid, dim3(kThreads), L.bytes, stream, Q, xf_env0, (const uint8_t*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr, tw, th, L.big_cap, static_keys);
}

}  // namespace tg
