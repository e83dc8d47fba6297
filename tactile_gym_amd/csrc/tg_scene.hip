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
#include "tg_scene.h"

#include <math.h>

namespace tg {

namespace {

constexpr int kThreads = 1024;
constexpr int kBigArea = 64;       // bounding boxes above this many pixels are filled cooperatively
constexpr int kBigCap = 2048;
constexpr int kMaxFrames = 16;

struct TriSetup {
    float a0, b0, c0, a1, b1, c1, a2, b2, c2, sg, rdet;
    uint32_t rgb;
    int x0, x1, y0, y1;            // pixel box (inclusive), empty if x0 > x1
};

// Everything about triangle t that does not depend on the pixel.  Returns false when nothing can be drawn.
__device__ __forceinline__ bool setup_tri(const SceneParams& P, const float* __restrict__ sxf, int t, TriSetup& S) {
    const int i0 = P.tris[3 * t + 0], i1 = P.tris[3 * t + 1], i2 = P.tris[3 * t + 2];
    const uint32_t attr = P.tri_attr[t];
    const float* M = sxf + 12 * (attr >> 24);
    const int idx[3] = {i0, i1, i2};
    float ex[3], ey[3], ez[3], w[3], X[3], Y[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float vx = P.verts[3 * idx[k] + 0], vy = P.verts[3 * idx[k] + 1], vz = P.verts[3 * idx[k] + 2];
        ex[k] = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
        ey[k] = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
        ez[k] = ((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11];
        w[k] = -ez[k];
        X[k] = P.kx * ex[k] + P.hw * w[k];
        Y[k] = P.hh * w[k] - P.ky * ey[k];
    }
    if (w[0] < P.near_ && w[1] < P.near_ && w[2] < P.near_) return false;
    if (w[0] > P.far_ && w[1] > P.far_ && w[2] > P.far_) return false;
    S.a0 = Y[1] * w[2] - Y[2] * w[1]; S.b0 = w[1] * X[2] - w[2] * X[1]; S.c0 = X[1] * Y[2] - X[2] * Y[1];
    S.a1 = Y[2] * w[0] - Y[0] * w[2]; S.b1 = w[2] * X[0] - w[0] * X[2]; S.c1 = X[2] * Y[0] - X[0] * Y[2];
    S.a2 = Y[0] * w[1] - Y[1] * w[0]; S.b2 = w[0] * X[1] - w[1] * X[0]; S.c2 = X[0] * Y[1] - X[1] * Y[0];
    const float det = (S.c0 * w[0] + S.c1 * w[1]) + S.c2 * w[2];
    if (det == 0.0f) return false;
    S.sg = det > 0.0f ? 1.0f : -1.0f;
    S.rdet = 1.0f / det;
    S.x0 = 0; S.x1 = P.W - 1; S.y0 = 0; S.y1 = P.H - 1;
    if (w[0] >= P.near_ && w[1] >= P.near_ && w[2] >= P.near_) {
        const float sx0 = X[0] / w[0], sx1 = X[1] / w[1], sx2 = X[2] / w[2], sy0 = Y[0] / w[0], sy1 = Y[1] / w[1], sy2 = Y[2] / w[2];
        float minx = fminf(sx0, fminf(sx1, sx2)), maxx = fmaxf(sx0, fmaxf(sx1, sx2)), miny = fminf(sy0, fminf(sy1, sy2)), maxy = fmaxf(sy0, fmaxf(sy1, sy2));
        minx = fminf(fmaxf(minx, -1.0f), (float)P.W + 1.0f); maxx = fminf(fmaxf(maxx, -1.0f), (float)P.W + 1.0f);
        miny = fminf(fmaxf(miny, -1.0f), (float)P.H + 1.0f); maxy = fminf(fmaxf(maxy, -1.0f), (float)P.H + 1.0f);
        S.x0 = max(0, (int)ceilf(minx - 0.515625f)); S.x1 = min(P.W - 1, (int)floorf(maxx - 0.484375f));
        S.y0 = max(0, (int)ceilf(miny - 0.515625f)); S.y1 = min(P.H - 1, (int)floorf(maxy - 0.484375f));
        if (S.x0 > S.x1 || S.y0 > S.y1) return false;
    }
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
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ save_prev, int tw, int th) {
    const int env = blockIdx.y;
    if (mask != nullptr && mask[env] == 0) return;
    extern __shared__ unsigned long long zb[];                     // [th][tw] keys, then the queue of large triangles
    __shared__ float sxf[kMaxFrames * 12];
    __shared__ int big_n;
    int* big = reinterpret_cast<int*>(zb + tw * th);
    const int tiles_x = P.W / tw;
    const int tx0 = (blockIdx.x % tiles_x) * tw, ty0 = (blockIdx.x / tiles_x) * th;
    const int tid = threadIdx.x;
    uint8_t* img = out + (size_t)env * P.W * P.H * 3;
    if (save_prev != nullptr) {
        uint8_t* dst = save_prev + (size_t)env * P.W * P.H * 3;
        for (int p = tid; p < tw * th; p += kThreads) {
            const size_t o = ((size_t)(ty0 + p / tw) * P.W + (tx0 + p % tw)) * 3;
            dst[o] = img[o]; dst[o + 1] = img[o + 1]; dst[o + 2] = img[o + 2];
        }
    }
    for (int p = tid; p < tw * th; p += kThreads) zb[p] = 0ull;
    for (int p = tid; p < P.n_frames * 12; p += kThreads) sxf[p] = xf[(size_t)env * P.n_frames * 12 + p];
    if (tid == 0) big_n = 0;
    __syncthreads();
    const int bx1 = tx0 + tw - 1, by1 = ty0 + th - 1;
    for (int t = tid; t < P.n_tris; t += kThreads) {
        TriSetup S;
        if (!setup_tri(P, sxf, t, S)) continue;
        const int x0 = max(S.x0, tx0), x1 = min(S.x1, bx1), y0 = max(S.y0, ty0), y1 = min(S.y1, by1);
        if (x0 > x1 || y0 > y1) continue;
        if ((x1 - x0 + 1) * (y1 - y0 + 1) > kBigArea) {
            const int slot = atomicAdd(&big_n, 1);
            if (slot < kBigCap) { big[slot] = t; continue; }
        }
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) shade_pixel(P, S, px, py, zb, tx0, ty0, tw);
    }
    __syncthreads();
    const int nb = min(big_n, kBigCap);
    for (int i = 0; i < nb; ++i) {
        TriSetup S;
        if (!setup_tri(P, sxf, big[i], S)) continue;               // workgroup-uniform
        const int x0 = max(S.x0, tx0), x1 = min(S.x1, bx1), y0 = max(S.y0, ty0), y1 = min(S.y1, by1);
        const int bw = x1 - x0 + 1, np = bw * (y1 - y0 + 1);
        for (int p = tid; p < np; p += kThreads) shade_pixel(P, S, x0 + p % bw, y0 + p / bw, zb, tx0, ty0, tw);
    }
    __syncthreads();
    for (int p = tid; p < tw * th; p += kThreads) {
        const unsigned long long k = zb[p];
        const size_t o = ((size_t)(ty0 + p / tw) * P.W + (tx0 + p % tw)) * 3;
        img[o + 0] = k ? (uint8_t)(k >> 16) : P.background[0];
        img[o + 1] = k ? (uint8_t)(k >> 8) : P.background[1];
        img[o + 2] = k ? (uint8_t)k : P.background[2];
    }
}

constexpr size_t lds_bytes(int tw, int th) { return (size_t)tw * th * 8 + (size_t)kBigCap * 4; }

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

int scene_prepare() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scene), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(128, 128));
}

void launch_scene(const SceneParams& P, const float* xf, int n_envs, const uint8_t* mask, uint8_t* out, uint8_t* save_prev, hipStream_t stream) {
    const int tw = P.W < 128 ? P.W : 128, th = P.H < 128 ? P.H : 128;
    dim3 grid((P.W / tw) * (P.H / th), n_envs);
    hipLaunchKernelGGL(k_scene, grid, dim3(kThreads), lds_bytes(tw, th), stream, P, xf, mask, out, save_prev, tw, th);
}

}  // namespace tg
