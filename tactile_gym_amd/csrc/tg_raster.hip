// tg_raster.hip — tactile depth raster + t_s_camera post-process for gfx950 (MI355X).
//
// Replaces, per env and per step, PyBullet's getCameraImage(...) depth channel and the numpy post-process of
// TactileSensor.t_s_camera (reference tactile_gym/sensors/tactile_sensor.py:239-246, 261-294).
//
// THIS FILE IS COMPILED WITH -ffp-contract=off.  Every float expression below follows the raster specification in
// DESIGN.md operation by operation (same association, no fused multiply-adds, IEEE division) so the image is
// bit-identical to the CPU oracle's scalar restatement (oracle/minibullet.c: mb_render_depth, mb_t_s_camera).
//
// Mapping to the hardware
//   * one 256-thread workgroup (4 wavefronts) per (env, image tile); a launch of 1024 envs is 2048-8192 workgroups >> 256 CUs, and
//     consecutive workgroups land on different XCDs (block b -> XCD b % 8) while the only shared data (reference images, stimulus
//     mesh: < 200 KB) is read-only and L2-resident on every XCD;
//   * triangle set-up (transform, near clip, project) is done once per workgroup, one lane per input triangle, and staged as compact
//     records in LDS; the pixel phase then reads each record as an LDS broadcast;
//   * each lane owns quads of 4 horizontally adjacent pixels whose depth values stay in VGPRs (the z-buffer): depth reduction is a
//     register min, the reference images are read as 16-byte vectors and the uint8 image is written as 4-byte vectors;
//   * four kernels, chosen per stimulus by launch_render:
//       k_render_blocks<16>        meshes of up to 32 triangles whose image is mostly the untouched sensor's (edge, cube): ONE workgroup per
//                                  128 x 128 region, 16 x 16 pixel blocks, depth-plane block culls, only changed blocks written (see there);
//       k_render_small<128,64,2>   the pole's plate (fills the view) and meshes of 33-256 triangles: 128 x 64 tiles, the pixel phase in two
//                                  passes over disjoint row groups, a wavefront visits its 32-pixel column band as 16 x 16 pixel blocks so
//                                  that passes which miss a record are skipped by the whole wavefront;
//       k_render_tactile<128,64,true>   per-env heightfields: only the grid window the truncated view frustum can reach is staged
//                                  (window-local LDS arrays) and culled, survivors are compacted, wavefront = 32-pixel band with per-record
//                                  band / row-group masks (scalar skips), 3 wavefronts per SIMD;
//       k_render_scatter / k_render_tactile<128,128,false> / <64,64,false>   everything else (the 960-triangle marble: triangle-parallel
//                                  into an LDS z-buffer; 64 x 64 images): full-width rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "tg_raster_dev.hpp"

namespace tg {


// The per-quad edge-function reject of k_render_tactile's pixel loop (a conservative cull, the image does not depend on it): off - it saves
// fewer instructions than its registers and its ~45 instructions per record x quad row cost (surface_follow render 66.4 -> 62.3 us without;
// -DTG_HF_QREJ brings it back for A/B).
#ifdef TG_HF_QREJ
constexpr bool kHfQuadReject = true;
#else
constexpr bool kHfQuadReject = false;
#endif
#ifndef TG_HF_WAVES
#define TG_HF_WAVES 3   // 3 wavefronts per SIMD for the 128 x 64 heightfield kernel (167 VGPRs, 64 B of scratch): 0.098 -> 0.078 ms; needs its windowed LDS (< 53 KB)
#endif
constexpr int kThreads = 256;
#ifndef TG_EDGE_CELL_MAX
#define TG_EDGE_CELL_MAX 256
#endif
constexpr int kEdgeCellMax = TG_EDGE_CELL_MAX;   // heightfield: the per-record cell test costs ~80 vector instructions per record on one wavefront; it pays while the records are few
                                    // (DIGIT on the horizontal surface: 56 us against 63), not for a view of hundreds (TacTip on the vertical one: 159 against 126)
constexpr int kCellsWinSide = 24;   // heightfield window (launch_render) up to which the cell masks are used: DIGIT 16, DigiTac 18 (TacTip: 32)
constexpr int kBatch = 1024;         // most triangle records staged in LDS per pass (56 KB); a small mesh allocates 2 * n_tris records only, so
                                     // that the 12-triangle edge does not hold the occupancy at 2 workgroups per CU (LDS-bound)

RasterParams make_raster_params(int W, int H, double fov_deg, double near_, double far_, int turn_off_border, const float* nodef_dep_host) {
    RasterParams P;
    const double ys = 1.0 / tan(0.5 * fov_deg * (3.14159265358979323846 / 180.0));
    P.W = W; P.H = H;
    P.kx = (float)(ys * 0.5 * W); P.ky = (float)(ys * 0.5 * H);
    P.hw = 0.5f * (float)W; P.hh = 0.5f * (float)H;
    P.C0 = (float)(far_ / (far_ - near_));
    P.C1 = (float)(-(near_ * far_) / (far_ - near_));
    P.near_ = (float)near_;
    P.turn_off_border = turn_off_border;
    { const char* e = getenv("TG_TERM_LAYER"); P.term_layer = (e != nullptr && e[0] == '0') ? 0 : 1; }   // A/B switch (measurements)
    P.blockmax = nullptr; P.tmpl = nullptr; P.drawn = nullptr; P.kt = nullptr;
    P.zcull = 0.0f;
    for (int i = 0; i < W * H; ++i) P.zcull = nodef_dep_host[i] > P.zcull ? nodef_dep_host[i] : P.zcull;
    return P;
}

int make_block_tables(RasterParams& P, const float* nodef_dep_host, const float* nodef_gray_host, const uint8_t* border_host, int n_envs, void** d_mem) {
    *d_mem = nullptr;
    if (P.W % 128 != 0 || P.H % 128 != 0) return 0;      // only the 128-multiple image sizes have the block kernel
    const int W = P.W, H = P.H, rxn = W / 128, ryn = H / 128, nbx = 128 / kBlockW, bh = 256 / kBlockW;
    const size_t npix = (size_t)W * H, nblk = (size_t)rxn * ryn * 64;
    std::vector<float> bmax(nblk, -1.0f);     // a block of pasted pixels only is reached by nothing (depths are >= 0)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int reg = (y / 128) * rxn + x / 128, b = ((y % 128) / bh) * nbx + (x % 128) / kBlockW;
            float& m = bmax[(size_t)reg * 64 + b];
            if (!P.turn_off_border && border_host[(size_t)y * W + x] == 1) continue;   // a pasted pixel does not depend on its depth
            const float v = nodef_dep_host[(size_t)y * W + x];
            m = v > m ? v : m;
        }
    std::vector<uint8_t> tmpl(npix, 0);
    if (!P.turn_off_border)
        for (size_t p = 0; p < npix; ++p) tmpl[p] = border_host[p] == 1 ? (uint8_t)nodef_gray_host[p] : 0;
    static const bool rewrite_all = getenv("TG_RASTER_REWRITE_ALL") != nullptr;   // every launch writes every block (a caller that scribbles on the image buffer)
    const size_t drawn_bytes = rewrite_all ? 0 : (((size_t)n_envs * rxn * ryn * 8 + 15) & ~(size_t)15), off_tmpl = drawn_bytes + nblk * 4;
    uint8_t* mem = nullptr;
    if (hipMalloc(&mem, off_tmpl + npix) != hipSuccess) return -1;
    if (hipMemset(mem, 0xFF, drawn_bytes ? drawn_bytes : 1) != hipSuccess ||      // nothing holds the untouched-sensor image yet
        hipMemcpy(mem + drawn_bytes, bmax.data(), nblk * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(mem + off_tmpl, tmpl.data(), npix, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(mem); return -1; }
    P.drawn = drawn_bytes ? reinterpret_cast<unsigned long long*>(mem) : nullptr;
    P.blockmax = reinterpret_cast<const float*>(mem + drawn_bytes);
    P.tmpl = mem + off_tmpl;
    *d_mem = mem;
    return 0;
}


// grid: (tiles_x * tiles_y, num_envs, 1 or 2); block: 256.  TW x TH = tile (128 x 128; 128 x 64 for small meshes: half the rows
// per lane halves the z-buffer registers, 4 instead of 2 workgroups fit a CU, and the per-workgroup set-up of a dozen triangles is
// negligible; 64 x 64 for 64x64 images).
// BAND (heightfield stimuli: hundreds of small triangles per tile): wavefront w owns the w-th 32-pixel column band of the tile (8 quad
// columns x 8 rows per pass) instead of two full rows, and skips - as one scalar branch - every record whose bounding box misses the band.
template <int TW, int TH, bool BAND, bool CELLS = false /* BAND only: per-record (band, pass) cell masks from the edge functions */>
__global__ __launch_bounds__(kThreads, (BAND && TH == 64) ? TG_HF_WAVES : 1) void k_render_tactile(RasterParams P, Stimulus S, const float* __restrict__ xform /*[12][n] SoA or [n][12] AoS*/,
                                                             int xform_soa, int n_envs, const uint8_t* __restrict__ mask,
                                                             const float* __restrict__ nodef_dep, const uint8_t* __restrict__ gray_u8,
                                                             const uint8_t* __restrict__ border, uint8_t* __restrict__ out,
                                                             uint8_t* __restrict__ save_prev /*nullable: copy old image here first*/,
                                                             int rec_cap /*records of dynamic LDS, even*/,
                                                             const float* __restrict__ term_xform, const uint8_t* __restrict__ term_mask,
                                                             uint8_t* __restrict__ term_out /*fused auto-reset: all three or none*/) {
    KtScope kt_scope_(P.kt);
    constexpr int QPR = TW / 4;            // pixel quads per tile row
    constexpr int RPP = kThreads / QPR;    // tile rows covered per pass of the workgroup
    constexpr int NK = TH / RPP;           // rows owned by each lane
    extern __shared__ TriRec recs[];                       // rec_cap records, then (heightfield stimulus) per vertex: height, projected
    float* hfl = reinterpret_cast<float*>(recs + rec_cap); // depth (-1: behind the near plane), tile outcode; then the survivor list.
    const int wcap = S.kind == 1 ? ((S.win_side * S.win_side + 3) & ~3) : 0;   // All of it indexed within the frustum window (<= win_side^2)
    float* hvd = hfl + wcap;
    unsigned short* surv = reinterpret_cast<unsigned short*>(hvd + wcap);      // 2 triangles per window cell
    uint8_t* hcode = reinterpret_cast<uint8_t*>(surv + 2 * wcap);
    unsigned* rbands = reinterpret_cast<unsigned*>(hcode + wcap);              // BAND: rec_cap words
    __shared__ int count, next_start, n_surv;
    const int env = blockIdx.y;
    if (mask != nullptr && mask[env] == 0) return;
    const int n_tris = S.n_tris;
    // (the heightfield's slot: the live one; the fused auto-reset's terminal layer draws the surface the finished episode ran on, which
    //  sits in the slot the env has just left: bits 2-3 of its hsel byte)
    const int hslot = (S.kind == 1 && S.hsel != nullptr) ? ((term_xform != nullptr && (int)blockIdx.z == P.term_layer) ? (S.hsel[env] >> 2) & 3 : S.hsel[env] & 3) : 0;
    const size_t hidx = (size_t)hslot * n_envs + env;
    const float hf_zoff = (S.kind == 1) ? S.zoff[hidx] : 0.0f;
    const double* hf = (S.kind == 1) ? S.heights + hidx * S.rows * S.cols : nullptr;
    const float hf_cx = 0.5f * (float)(S.rows - 1), hf_cy = 0.5f * (float)(S.cols - 1);
    const int tiles_x = P.W / TW;
    const int tile_x = (blockIdx.x % tiles_x) * TW, tile_y = (blockIdx.x / tiles_x) * TH;
    const int tid = threadIdx.x;

    // fused auto-reset: grid z = P.term_layer (1) draws the terminal observation (term_xform -> term_out) of the envs flagged in term_mask and is
    // empty for every other env; the other layer is the regular image.  (Terminal layer first - its few real tiles dispatched before everything
    // else instead of after - was measured in round 6, 1024 envs: edge_follow 42.7 against 42.4 us per step, surface_follow 102.4 against 100.4 us
    // aligned and 111.6 against 110.7 us with the episodes out of phase: last stays.)
    const int pass = (term_xform != nullptr && (int)blockIdx.z == P.term_layer) ? 0 : 1;
    if (pass == 0 && term_mask[env] == 0) return;
    const float* __restrict__ xf = pass == 0 ? term_xform : xform;
    uint8_t* __restrict__ img = pass == 0 ? term_out : out;
    float M[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = xform_soa ? xf[(size_t)k * n_envs + env] : xf[(size_t)env * 12 + k];

    // Lane -> pixel mapping.  Default: a pass of the workgroup covers RPP full tile rows (a wavefront: a 128 x 2 strip).  BAND (heightfield):
    // wavefront w owns the w-th 32-pixel column band (8 quad columns x 8 rows per pass) and skips records whose bounding box misses it.
    // (Meshes of a few large triangles go to k_render_small and its 16 x 16 pass blocks; for many small triangles the per-pass x test
    // that mapping needs costs more than it saves: heightfield 0.15 -> 0.23 ms, marble 0.80 -> 1.50 ms.)
    static_assert(!BAND || (TW == 128 && RPP == 8), "banded record skip: 4 wavefronts x 32-pixel bands");
    const int band = __builtin_amdgcn_readfirstlane(tid / 64);
    const int qx0 = BAND ? tile_x + 32 * band + 4 * (tid % 8) : tile_x + 4 * (tid % QPR);
    const int ry0 = BAND ? tile_y + ((tid % 64) / 8) : tile_y + (tid / QPR);
#define TG_QX(kk) (qx0)
#define TG_RY(kk) (ry0 + RPP * (kk))
    float z[NK][4];
    unsigned touched = 0;      // bit k: some triangle lowered a depth of row k of this lane's quad column
    {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float4 nd = *reinterpret_cast<const float4*>(nodef_dep + (size_t)TG_RY(k) * P.W + TG_QX(k));
            z[k][0] = nd.x; z[k][1] = nd.y; z[k][2] = nd.z; z[k][3] = nd.w;
        }
    }
    const float tx0 = (float)tile_x, ty0 = (float)tile_y, tx1 = (float)(tile_x + TW), ty1 = (float)(tile_y + TH);
    int win_i0 = 0, win_i1 = -1, win_j0 = 0, win_j1 = -1;   // heightfield: window of grid vertices that can reach the image
    if (S.kind == 1) {   // stage the env's vertices in LDS once (one coalesced 32 KB read instead of three dependent global loads per
                         // triangle): the height, same expression as a direct fetch, (float)h - zoff, and the projected depth of the
                         // vertex and its window position (depth -1 when it is behind the near plane), which lets the triangle loop apply emit()'s
                         // depth and tile culls to the 7 938 triangles with a handful of LDS reads, before any transform or clipping
        // (the height samples are fetched in batches of 8 per lane before any of them is used: issued one per iteration each paid its own
        // HBM / L2 round trip - 8.6 us of a 43 us workgroup)
        // Only a window of the grid can reach the image.  Everything emit() keeps lies inside the view frustum in front of the near plane
        // and closer than w_cull, the eye depth at which the depth value reaches zcull (beyond it min d_k - slack >= zcull culls the
        // triangle); that truncated pyramid is convex, so its xy bounding box in the heightfield frame - through its apex and the four far
        // corners - contains the xy of every visible point, and a grid triangle (one cell across) that touches it has all its vertices within
        // one cell of the box.  The window is the box widened by two cells; the rest of the 4 096 vertices / 7 938 triangles is never touched
        // (a DIGIT sees about 6 x 6 cells).  Conservative, hence exact.
        {
            const float w_cull = 1.01f * (P.C1 / ((P.zcull + 2.0f * kDepthSlack) - P.C0));   // d = C0 + C1 / w, C1 < 0
            const float ex = w_cull * (P.hw / P.kx), ey = w_cull * (P.hh / P.ky);
            float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
#pragma unroll
            for (int cnr = 0; cnr < 5; ++cnr) {
                const float pcx = cnr == 0 ? 0.0f : ((cnr & 1) ? ex : -ex), pcy = cnr == 0 ? 0.0f : ((cnr & 2) ? ey : -ey);
                const float pcz = cnr == 0 ? 0.0f : -w_cull;
                const float qx_ = pcx - M[9], qy_ = pcy - M[10], qz_ = pcz - M[11];              // object = R^T (camera - t)
                const float ox_ = (M[0] * qx_ + M[3] * qy_) + M[6] * qz_, oy_ = (M[1] * qx_ + M[4] * qy_) + M[7] * qz_;
                xlo = fminf(xlo, ox_); xhi = fmaxf(xhi, ox_); ylo = fminf(ylo, oy_); yhi = fmaxf(yhi, oy_);
            }
            const float inv = 1.0f / S.scale;
            int i0 = (int)floorf(xlo * inv + hf_cx) - 2, i1 = (int)ceilf(xhi * inv + hf_cx) + 2;
            int j0 = (int)floorf(ylo * inv + hf_cy) - 2, j1 = (int)ceilf(yhi * inv + hf_cy) + 2;
            if (!(xlo <= xhi)) { i0 = 0; i1 = S.rows - 1; j0 = 0; j1 = S.cols - 1; }           // NaN in the transform: take everything
            win_i0 = max(i0, 0); win_i1 = min(i1, S.rows - 1); win_j0 = max(j0, 0); win_j1 = min(j1, S.cols - 1);
            // win_side (host) bounds the window for every camera orientation: the pyramid fits the sphere of radius |far corner| about its
            // apex.  The clamp below can therefore not cut anything; it only keeps a wrong bound from overrunning the LDS arrays.
            win_i1 = min(win_i1, win_i0 + S.win_side - 1); win_j1 = min(win_j1, win_j0 + S.win_side - 1);
        }
        const int wi = win_i1 - win_i0 + 1, wj = win_j1 - win_j0 + 1;
        const int nv = (wi > 0 && wj > 0) ? wi * wj : 0;
        // (the height samples are fetched in batches of 4 per lane before any of them is used)
        for (int base = 0; base < nv; base += 4 * kThreads) {
            double hh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = base + u * kThreads + tid;
                hh[u] = w < nv ? hf[(win_j0 + w / wi) * S.rows + (win_i0 + w % wi)] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = base + u * kThreads + tid;
                if (w >= nv) continue;
                const int vi = win_i0 + w % wi, vj = win_j0 + w / wi, i = w;   // window-local vertex index
                const float vz = (float)hh[u] - hf_zoff;
                hfl[i] = vz;
                const float vx = ((float)vi - hf_cx) * S.scale, vy = ((float)vj - hf_cy) * S.scale;
                const float cx = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
                const float cy = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
                const float cw = -(((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11]);
                float d = -1.0f, sx = 0.0f, sy = 0.0f;
                if (cw >= P.near_) project_vertex(cx, cy, cw, P, sx, sy, d);
                hvd[i] = d;
                hcode[i] = (uint8_t)((sx < tx0 ? 1 : 0) | (sx > tx1 ? 2 : 0) | (sy < ty0 ? 4 : 0) | (sy > ty1 ? 8 : 0));   // tile outcode
            }
        }
        __syncthreads();
    }

    // Heightfield: first pass over all triangles with the cheap exact culls on the staged per-vertex data; the survivors (mostly the
    // triangles that straddle the near plane, a few hundred of 7 938) go to a list, so that the expensive path below runs on dense
    // wavefronts of survivors instead of once per loop iteration for whichever single lane happened to need it.
    int total = n_tris;
    if (S.kind == 1) {
        if (tid == 0) n_surv = 0;
        __syncthreads();
        // one lane per grid cell: its two triangles share the four corner vertices (half as many LDS reads as one lane per triangle)
        const int wci = win_i1 - win_i0, wcj = win_j1 - win_j0;              // cells of the window
        const int n_cells = (wci > 0 && wcj > 0) ? wci * wcj : 0;
        for (int wc = tid; wc < n_cells; wc += kThreads) {
            const int cell = wc;                                                                    // window-local cell index
            const int v00 = (wc / wci) * (wci + 1) + wc % wci, v10 = v00 + 1, v01 = v00 + (wci + 1), v11 = v01 + 1;   // (i,j) (i+1,j) (i,j+1) (i+1,j+1)
            const float d00 = hvd[v00], d10 = hvd[v10], d01 = hvd[v01], d11 = hvd[v11];
            const unsigned c00 = hcode[v00], c10 = hcode[v10], c01 = hcode[v01], c11 = hcode[v11];
#pragma unroll
            for (int half = 0; half < 2; ++half) {   // half 0: (i,j),(i,j+1),(i+1,j)   half 1: (i+1,j),(i,j+1),(i+1,j+1)
                const float d0 = half == 0 ? d00 : d10, d1 = d01, d2 = half == 0 ? d10 : d11;
                const unsigned o0 = half == 0 ? c00 : c10, o1 = c01, o2 = half == 0 ? c10 : c11;
                if (d0 < 0.0f && d1 < 0.0f && d2 < 0.0f) continue;   // wholly behind the near plane (the hills of the surface rise past the
                                                                     // camera elsewhere): the clipper would return no polygon
                if (d0 >= 0.0f && d1 >= 0.0f && d2 >= 0.0f) {        // not clipped: the very culls of emit()
                    if (fminf(d0, fminf(d1, d2)) - kDepthSlack >= P.zcull) continue;
                    if ((o0 & o1 & o2) != 0) continue;               // all three on one outer side of the tile = emit()'s bbox test
                }
                surv[atomicAdd(&n_surv, 1)] = (unsigned short)(2 * cell + half);
            }
        }
        __syncthreads();
        total = n_surv;
    }

    // Rounds: every lane takes work items (triangles, or survivors of the first pass) from `start`; a projected triangle that passes
    // emit()'s culls takes a record slot; when the buffer is full the smallest item index that found no room becomes the next round's
    // start (re-emitting a triangle is harmless for a depth-min).  Normally one round: two barriers.
    bool cull = false;            // back-face cull of a closed, outward-wound mesh that lies wholly beyond the near plane (see back_facing)
    if (S.kind == 0 && S.closed_outward) {
        bool beyond = true;
        for (int t = tid; t < n_tris; t += kThreads) beyond = beyond && tri_beyond_near(S.soup, t, M, P.near_);
        cull = __syncthreads_and(beyond ? 1 : 0) != 0;
    }
    for (int start = 0; start < total;) {
        if (tid == 0) { count = 0; next_start = total; }
        __syncthreads();
        for (int it = start + tid; it < total; it += kThreads) {
            const int t = S.kind == 1 ? (int)surv[it] : it;
            float cx[3], cy[3], cw[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float vx, vy, vz;
                if (S.kind == 1) {
                    const int cell = t >> 1, half = t & 1;              // window-local cell
                    const int wci = win_i1 - win_i0;
                    const int lci = cell % wci, lcj = cell / wci;
                    // half 0: (i,j),(i,j+1),(i+1,j)   half 1: (i+1,j),(i,j+1),(i+1,j+1)
                    const int di = half == 0 ? (k == 2) : (k != 1);
                    const int dj = half == 0 ? (k == 1) : (k != 0);
                    const int vi = win_i0 + lci + di, vj = win_j0 + lcj + dj;
                    vx = ((float)vi - hf_cx) * S.scale;
                    vy = ((float)vj - hf_cy) * S.scale;
                    vz = hfl[(lcj + dj) * (wci + 1) + (lci + di)];
                } else {
                    const float* v = S.soup + 9 * t + 3 * k;   // pre-expanded triangles: one round trip instead of index -> vertex
                    vx = v[0]; vy = v[1]; vz = v[2];
                }
                cx[k] = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
                cy[k] = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
                cw[k] = -(((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11]);
            }
            if (S.kind == 0 && cull && back_facing(cx, cy, cw)) continue;
            // near-plane clip (Sutherland-Hodgman on w >= near), vertex order 0,1,2
            float ox[4], oy[4], ow[4];
            int no = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int k1 = (k + 1) % 3;
                const bool ain = cw[k] >= P.near_, bin = cw[k1] >= P.near_;
                if (ain) { ox[no] = cx[k]; oy[no] = cy[k]; ow[no] = cw[k]; ++no; }
                if (ain != bin) {
                    const float tt = (P.near_ - cw[k]) / (cw[k1] - cw[k]);
                    ox[no] = cx[k] + tt * (cx[k1] - cx[k]);
                    oy[no] = cy[k] + tt * (cy[k1] - cy[k]);
                    ow[no] = P.near_;
                    ++no;
                }
            }
            bool ok = true;
            if (no >= 3) ok = emit(recs, &count, rec_cap, ox, oy, ow, 0, 1, 2, P, tx0, ty0, tx1, ty1, BAND ? rbands : nullptr);
            if (ok && no == 4) ok = emit(recs, &count, rec_cap, ox, oy, ow, 0, 2, 3, P, tx0, ty0, tx1, ty1, BAND ? rbands : nullptr);
            if (!ok) { atomicMin(&next_start, it); break; }
        }
        __syncthreads();
        const int n = min(count, rec_cap);
        start = next_start;
        const bool cells = BAND && CELLS && n <= kEdgeCellMax;   // (wave-uniform)
        if (cells) {
            // per record: the (band, pass) cells - 32 x 8 pixel rectangles, bit 4 k + band - whose pixel centres the TRIANGLE can cover, not just
            // its bounding box (edges_exclude_rect: rigorous).  Lane l < 32 of each wavefront stands for cell (band l & 3, pass l >> 2).
            const int l = tid & 63, cb = l & 3, ck = (l >> 2) & 7;
            const float X0 = tx0 + 32.0f * (float)cb + 0.5f, X1 = X0 + 31.0f, Y0 = ty0 + 8.0f * (float)ck + 0.5f, Y1 = Y0 + 7.0f;
            for (int t = band; t < n; t += kThreads / 64) {
                const TriRec r = recs[t];
                const unsigned rb0 = rbands[t];
                bool miss = !((rb0 >> cb) & 1u) | !((rb0 >> (4 + ck)) & 1u);
                miss = miss | edges_exclude_rect(r.x0, r.y0, r.x1, r.y1, r.x2, r.y2, X0, X1, Y0, Y1);
                const unsigned long long m = __ballot(!miss);
                if (l == 0) rbands[t] = (unsigned)m;
            }
            __syncthreads();
        }
        for (int t = 0; t < n; ++t) {
            unsigned rb = 0u;
            if (BAND) {
                rb = __builtin_amdgcn_readfirstlane(rbands[t]);
                if (cells ? !((rb >> band) & 0x11111111u) : !((rb >> band) & 1u)) continue;   // wave-uniform
            }
            const TriRec r = recs[t];   // same address on every lane: LDS broadcast
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (BAND && (cells ? !((rb >> (4 * k + band)) & 1u) : !((rb >> (4 + k)) & 1u))) continue;   // scalar: this pass's 32 x 8 cell is out of the record's reach
                const int qx = TG_QX(k);
                const float fy = (float)TG_RY(k) + 0.5f;
                if (fy < r.ymin || fy > r.ymax) continue;
                if ((float)qx + 3.5f < r.xmin || (float)qx + 0.5f > r.xmax) continue;
                if (r.dmin >= fmaxf(fmaxf(z[k][0], z[k][1]), fmaxf(z[k][2], z[k][3]))) continue;
                const float a0 = r.y2 - fy, a1 = r.y1 - fy, a2 = r.y0 - fy;
                if (kHfQuadReject) {   // conservative reject of the 4-pixel quad: e_i is affine in fx (slope y_j - y_k along a row), so its value at the quad
                    // centre plus 1.5 |slope| plus a bound on the float rounding of the per-pixel expression bounds it over the quad.  e0 + e1 + e2
                    // is the same at every pixel (twice the signed area): when its sign is certain, a covered pixel needs all three e_i on
                    // that side, so one edge function provably on the other side rejects the quad.  Skipping changes no pixel.
                    const float fc = (float)qx + 2.0f;
                    const float b0 = r.x0 - fc, b1 = r.x1 - fc, b2 = r.x2 - fc;
                    const float c0 = b1 * a0 - b2 * a1, c1 = b2 * a2 - b0 * a0, c2 = b0 * a1 - b1 * a2;
                    const float A0 = fabsf(a0), A1 = fabsf(a1), A2 = fabsf(a2);
                    const float B0 = fabsf(b0) + 1.5f, B1 = fabsf(b1) + 1.5f, B2 = fabsf(b2) + 1.5f;
                    const float m0 = 1e-5f * (B1 * A0 + B2 * A1), m1 = 1e-5f * (B2 * A2 + B0 * A0), m2 = 1e-5f * (B0 * A1 + B1 * A2);
                    const float h0 = 1.5f * fabsf(r.y1 - r.y2) + m0, h1 = 1.5f * fabsf(r.y2 - r.y0) + m1, h2 = 1.5f * fabsf(r.y0 - r.y1) + m2;
                    const float sc = (c0 + c1) + c2, ms = 4.0f * ((m0 + m1) + m2);
                    const bool out_pos = (sc > ms) & ((c0 < -h0) | (c1 < -h1) | (c2 < -h2));
                    const bool out_neg = (sc < -ms) & ((c0 > h0) | (c1 > h1) | (c2 > h2));
                    if (out_pos | out_neg) continue;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float fx = (float)(qx + p) + 0.5f;
                    const float e0 = (r.x1 - fx) * a0 - (r.x2 - fx) * a1;
                    const float e1 = (r.x2 - fx) * a2 - (r.x0 - fx) * a0;
                    const float e2 = (r.x0 - fx) * a1 - (r.x1 - fx) * a2;
                    // bitwise, not short-circuit, and the depth evaluated unconditionally (discarded, possibly inf/NaN, when the
                    // pixel is not covered): one straight line of VALU work instead of a scalar branch per clause
                    const bool box = (fx >= r.xmin) & (fx <= r.xmax);
                    const bool pos = (e0 >= 0.0f) & (e1 >= 0.0f) & (e2 >= 0.0f), neg = (e0 <= 0.0f) & (e1 <= 0.0f) & (e2 <= 0.0f);
                    const float s = (e0 + e1) + e2;
                    const float d = div_mid_range((e0 * r.d0 + e1 * r.d1) + e2 * r.d2, s);
                    const bool hit = box & (pos | neg) & (s != 0.0f) & (d < z[k][p]);
                    z[k][p] = hit ? d : z[k][p];
                    touched |= hit ? (1u << k) : 0u;
                }
            }
        }
        __syncthreads();
    }

    // t_s_camera: depth difference -> uint8 penetration image, border paste (tactile_sensor.py:271-292).  A pixel no triangle
    // lowered still holds nodef_dep bit for bit, so its difference is exactly 0: the reference depth image is read a second time only
    // for the rows this lane actually touched (the kernel is bound by L2 traffic: 208 KB of reference images per workgroup before,
    // 64 KB + 32 KB + the touched rows now).  gray_u8 = uint8(nodef_gray), converted once on the host.
    const float eps = 1e-4f, max_pen = 0.05f;
    uint8_t* dst = img + (size_t)env * P.W * P.H;
    uint8_t* prev = (save_prev && pass == 1) ? save_prev + (size_t)env * P.W * P.H : nullptr;
    // all loads of the post-process first (they are independent of one another), then the arithmetic and the stores: issued row by
    // row, each row paid its own L2 round trip (measured with s_memtime: 18.7 k of the 54 k cycles of a workgroup)
    uchar4 ngk[NK], bmk[NK];
    float4 ndk[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const size_t off = (size_t)TG_RY(k) * P.W + TG_QX(k);
        ngk[k] = *reinterpret_cast<const uchar4*>(gray_u8 + off);
        bmk[k] = *reinterpret_cast<const uchar4*>(border + off);
        ndk[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if ((touched >> k) & 1u) ndk[k] = *reinterpret_cast<const float4*>(nodef_dep + off);
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const size_t off = (size_t)TG_RY(k) * P.W + TG_QX(k);
        const uint8_t ngv[4] = {ngk[k].x, ngk[k].y, ngk[k].z, ngk[k].w}, bmv[4] = {bmk[k].x, bmk[k].y, bmk[k].z, bmk[k].w};
        const float ndv[4] = {ndk[k].x, ndk[k].y, ndk[k].z, ndk[k].w};
        uint8_t o[4] = {0, 0, 0, 0};
        if ((touched >> k) & 1u) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float diff = z[k][p] - ndv[p];
                if (diff >= -eps && diff <= eps) diff = 0.0f;
                const float pen = fabsf(diff);
                const float cl = pen < 0.0f ? 0.0f : (pen > max_pen ? max_pen : pen);
                o[p] = (uint8_t)(div_mid_range(cl, max_pen) * 255.0f);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (!P.turn_off_border && bmv[p] == 1) o[p] = ngv[p];
        if (prev) *reinterpret_cast<uchar4*>(prev + off) = *reinterpret_cast<const uchar4*>(dst + off);
        *reinterpret_cast<uchar4*>(dst + off) = make_uchar4(o[0], o[1], o[2], o[3]);
    }
}
#undef TG_QX
#undef TG_RY

// Shared meshes of many small triangles (the 960-triangle marble): triangle-parallel.  A lane takes a triangle through the same
// transform / back-face cull / near clip / projection as above and min-reduces its depths straight into an LDS z-buffer (ds_min_u32 on
// the order-preserving integer image of the float) - the pixel-parallel kernels above test every pixel against every record, which for
// ~500 front faces of ~20 pixels each is two orders of magnitude more work (marble render 1.19 ms -> see DESIGN.md).  A depth image is a
// min over covering triangles, so the order of the atomics does not matter and the result is the one the oracle's sequential loop gives,
// bit for bit.  Triangles whose box spans more than kScatterBig pixels of the tile are queued and filled by a whole wavefront each.
constexpr int kScatterBig = 96, kScatterCap = 256;
__device__ __forceinline__ unsigned depth_key(float d) { const unsigned b = __float_as_uint(d); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key_depth(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__device__ __forceinline__ void scatter_pixel(const TriRec& r, int px, int py, unsigned* zb, int tile_x, int tile_y, int TW) {
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float a0 = r.y2 - fy, a1 = r.y1 - fy, a2 = r.y0 - fy;
    const float e0 = (r.x1 - fx) * a0 - (r.x2 - fx) * a1;
    const float e1 = (r.x2 - fx) * a2 - (r.x0 - fx) * a0;
    const float e2 = (r.x0 - fx) * a1 - (r.x1 - fx) * a2;
    const bool box = (fx >= r.xmin) & (fx <= r.xmax) & (fy >= r.ymin) & (fy <= r.ymax);
    const bool pos = (e0 >= 0.0f) & (e1 >= 0.0f) & (e2 >= 0.0f), neg = (e0 <= 0.0f) & (e1 <= 0.0f) & (e2 <= 0.0f);
    const float s = (e0 + e1) + e2;
    if (!(box & (pos | neg) & (s != 0.0f))) return;
    const float d = div_mid_range((e0 * r.d0 + e1 * r.d1) + e2 * r.d2, s);
    if (d == d) atomicMin(&zb[(py - tile_y) * TW + (px - tile_x)], depth_key(d));   // `d < z` is false for a NaN
}
// pixel-centre range of a record inside the tile (centres fx with xmin <= fx <= xmax; a one-pixel margin, the predicate decides)
__device__ __forceinline__ void scatter_box(const TriRec& r, int tile_x, int tile_y, int TW, int TH, int& x0, int& x1, int& y0, int& y1) {
    const float lo_x = fminf(fmaxf(r.xmin, (float)tile_x - 1.0f), (float)(tile_x + TW) + 1.0f), hi_x = fminf(fmaxf(r.xmax, (float)tile_x - 1.0f), (float)(tile_x + TW) + 1.0f);
    const float lo_y = fminf(fmaxf(r.ymin, (float)tile_y - 1.0f), (float)(tile_y + TH) + 1.0f), hi_y = fminf(fmaxf(r.ymax, (float)tile_y - 1.0f), (float)(tile_y + TH) + 1.0f);
    x0 = max(tile_x, (int)floorf(lo_x - 0.5f)); x1 = min(tile_x + TW - 1, (int)ceilf(hi_x - 0.5f));
    y0 = max(tile_y, (int)floorf(lo_y - 0.5f)); y1 = min(tile_y + TH - 1, (int)ceilf(hi_y - 0.5f));
}

template <int TW, int TH>
__global__ __launch_bounds__(kThreads) void k_render_scatter(RasterParams P, Stimulus S, const float* __restrict__ xform, int xform_soa, int n_envs,
                                                             const uint8_t* __restrict__ mask, const float* __restrict__ nodef_dep,
                                                             const uint8_t* __restrict__ gray_u8, const uint8_t* __restrict__ border,
                                                             uint8_t* __restrict__ out, uint8_t* __restrict__ save_prev,
                                                             const float* __restrict__ term_xform, const uint8_t* __restrict__ term_mask,
                                                             uint8_t* __restrict__ term_out) {
    KtScope kt_scope_(P.kt);
    __shared__ unsigned zb[TW * TH];
    __shared__ TriRec big[kScatterCap];
    __shared__ int big_n;
    const int env = blockIdx.y;
    if (mask != nullptr && mask[env] == 0) return;
    const int pass = (term_xform != nullptr && (int)blockIdx.z == P.term_layer) ? 0 : 1;
    if (pass == 0 && term_mask[env] == 0) return;
    const float* __restrict__ xf = pass == 0 ? term_xform : xform;
    uint8_t* __restrict__ img = pass == 0 ? term_out : out;
    const int tid = threadIdx.x;
    const int tiles_x = P.W / TW;
    const int tile_x = (blockIdx.x % tiles_x) * TW, tile_y = (blockIdx.x / tiles_x) * TH;
    float M[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = xform_soa ? xf[(size_t)k * n_envs + env] : xf[(size_t)env * 12 + k];
    for (int p = 4 * tid; p < TW * TH; p += 4 * kThreads) {
        const float4 nd = *reinterpret_cast<const float4*>(nodef_dep + (size_t)(tile_y + p / TW) * P.W + tile_x + p % TW);
        zb[p] = depth_key(nd.x); zb[p + 1] = depth_key(nd.y); zb[p + 2] = depth_key(nd.z); zb[p + 3] = depth_key(nd.w);
    }
    if (tid == 0) big_n = 0;
    bool beyond = true;
    if (S.closed_outward)
        for (int t = tid; t < S.n_tris; t += kThreads) beyond = beyond && tri_beyond_near(S.soup, t, M, P.near_);
    const bool cull = __syncthreads_and(beyond ? 1 : 0) != 0 && S.closed_outward != 0;   // (also the barrier after the z-buffer fill)
    const float tx0 = (float)tile_x, ty0 = (float)tile_y, tx1 = (float)(tile_x + TW), ty1 = (float)(tile_y + TH);
    for (int t = tid; t < S.n_tris; t += kThreads) {
        float cx[3], cy[3], cw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* v = S.soup + 9 * t + 3 * k;
            cx[k] = ((M[0] * v[0] + M[1] * v[1]) + M[2] * v[2]) + M[9];
            cy[k] = ((M[3] * v[0] + M[4] * v[1]) + M[5] * v[2]) + M[10];
            cw[k] = -(((M[6] * v[0] + M[7] * v[1]) + M[8] * v[2]) + M[11]);
        }
        if (cull && back_facing(cx, cy, cw)) continue;
        float ox[4], oy[4], ow[4];   // near-plane clip (Sutherland-Hodgman on w >= near), vertex order 0,1,2
        int no = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int k1 = (k + 1) % 3;
            const bool ain = cw[k] >= P.near_, bin = cw[k1] >= P.near_;
            if (ain) { ox[no] = cx[k]; oy[no] = cy[k]; ow[no] = cw[k]; ++no; }
            if (ain != bin) {
                const float tt = (P.near_ - cw[k]) / (cw[k1] - cw[k]);
                ox[no] = cx[k] + tt * (cx[k1] - cx[k]);
                oy[no] = cy[k] + tt * (cy[k1] - cy[k]);
                ow[no] = P.near_;
                ++no;
            }
        }
        for (int part = 0; part < 2; ++part) {
            if (part == 0 ? no < 3 : no != 4) continue;
            const int b = part == 0 ? 1 : 2, c = part == 0 ? 2 : 3;
            TriRec r;
            project_vertex(ox[0], oy[0], ow[0], P, r.x0, r.y0, r.d0);
            project_vertex(ox[b], oy[b], ow[b], P, r.x1, r.y1, r.d1);
            project_vertex(ox[c], oy[c], ow[c], P, r.x2, r.y2, r.d2);
            r.xmin = fminf(r.x0, fminf(r.x1, r.x2)); r.xmax = fmaxf(r.x0, fmaxf(r.x1, r.x2));
            r.ymin = fminf(r.y0, fminf(r.y1, r.y2)); r.ymax = fmaxf(r.y0, fmaxf(r.y1, r.y2));
            r.dmin = fminf(r.d0, fminf(r.d1, r.d2)) - kDepthSlack;
            if (r.xmax < tx0 || r.xmin > tx1 || r.ymax < ty0 || r.ymin > ty1) continue;   // the culls of emit()
            if (r.dmin >= P.zcull) continue;
            int x0, x1, y0, y1;
            scatter_box(r, tile_x, tile_y, TW, TH, x0, x1, y0, y1);
            if (x0 > x1 || y0 > y1) continue;
            if ((x1 - x0 + 1) * (y1 - y0 + 1) > kScatterBig) {
                const int slot = atomicAdd(&big_n, 1);
                if (slot < kScatterCap) { big[slot] = r; continue; }
            }
            for (int py = y0; py <= y1; ++py)
                for (int px = x0; px <= x1; ++px) scatter_pixel(r, px, py, zb, tile_x, tile_y, TW);
        }
    }
    __syncthreads();
    {
        const int nb = min(big_n, kScatterCap), wave = tid >> 6, lane = tid & 63;
        for (int i = wave; i < nb; i += kThreads / 64) {
            const TriRec r = big[i];
            int x0, x1, y0, y1;
            scatter_box(r, tile_x, tile_y, TW, TH, x0, x1, y0, y1);
            const int bw = x1 - x0 + 1, np = bw * (y1 - y0 + 1);
            for (int p = lane; p < np; p += 64) scatter_pixel(r, x0 + p % bw, y0 + p / bw, zb, tile_x, tile_y, TW);
        }
    }
    __syncthreads();
    // t_s_camera (tactile_sensor.py:271-292), as in k_render_tactile: a pixel no triangle lowered holds nodef_dep bit for bit -> 0
    const float eps = 1e-4f, max_pen = 0.05f;
    uint8_t* dst = img + (size_t)env * P.W * P.H;
    uint8_t* prev = (save_prev && pass == 1) ? save_prev + (size_t)env * P.W * P.H : nullptr;
    for (int p = 4 * tid; p < TW * TH; p += 4 * kThreads) {
        const size_t off = (size_t)(tile_y + p / TW) * P.W + tile_x + p % TW;
        const float4 nd = *reinterpret_cast<const float4*>(nodef_dep + off);
        const uchar4 ng = *reinterpret_cast<const uchar4*>(gray_u8 + off), bm = *reinterpret_cast<const uchar4*>(border + off);
        const float ndv[4] = {nd.x, nd.y, nd.z, nd.w};
        const uint8_t ngv[4] = {ng.x, ng.y, ng.z, ng.w}, bmv[4] = {bm.x, bm.y, bm.z, bm.w};
        uint8_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float diff = key_depth(zb[p + q]) - ndv[q];
            if (diff >= -eps && diff <= eps) diff = 0.0f;
            const float pen = fabsf(diff);
            const float cl = pen < 0.0f ? 0.0f : (pen > max_pen ? max_pen : pen);
            o[q] = (uint8_t)(div_mid_range(cl, max_pen) * 255.0f);
            if (!P.turn_off_border && bmv[q] == 1) o[q] = ngv[q];
        }
        if (prev) *reinterpret_cast<uchar4*>(prev + off) = *reinterpret_cast<const uchar4*>(dst + off);
        *reinterpret_cast<uchar4*>(dst + off) = make_uchar4(o[0], o[1], o[2], o[3]);
    }
}

// Small shared meshes (the 12-triangle edge): every triangle fits the record buffer in one round (rec_cap = 2 n_tris), so the pixel
// phase can run in HALVES passes over disjoint row groups, each pass carrying only its own slice of the z-buffer from the reference
// load to the store: 16 instead of 32 live depth registers, which lifts the kernel over the next occupancy step (VGPR budget).
// __launch_bounds__(., 6): six wavefronts per SIMD (80 VGPRs, 64 B of scratch) instead of the five its 96 VGPRs allow - a workgroup spends most
// of its 8 us waiting on memory, so resident workgroups are what count: edge_follow render 31.6 -> 28.4 us, 16 384 envs 0.290 -> 0.240 ms,
// object_balance 256 x 256 0.173 -> 0.153 ms; seven (72 VGPRs) the same, eight (64 VGPRs, 112 B of scratch) slower (32.9 us).
template <int TW, int TH, int HALVES, bool QREJ>
__global__ __launch_bounds__(kThreads, 6) void k_render_small(RasterParams P, Stimulus S, const float* __restrict__ xform, int xform_soa, int n_envs,
                                                           const uint8_t* __restrict__ mask, const float* __restrict__ nodef_dep,
                                                           const uint8_t* __restrict__ gray_u8, const uint8_t* __restrict__ border,
                                                           uint8_t* __restrict__ out, uint8_t* __restrict__ save_prev, int rec_cap,
                                                           const float* __restrict__ term_xform, const uint8_t* __restrict__ term_mask,
                                                           uint8_t* __restrict__ term_out) {
    KtScope kt_scope_(P.kt);
    constexpr int QPR = TW / 4, RPP = kThreads / QPR, NK = TH / RPP, NKH = NK / HALVES;
    static_assert(NK % HALVES == 0, "row groups");
    extern __shared__ TriRec recs[];
    __shared__ int count;
    __shared__ unsigned rblocks[512];    // per record: the 16 x 16 pass blocks of this tile (bit 8 by + bx) it can cover a pixel of (rec_cap <= 2 x 256)
    __shared__ unsigned rinside[512];    // ... and those it covers every pixel centre of (edges_cover_rect): no coverage test per pixel there
    const int env = blockIdx.y;
    if (mask != nullptr && mask[env] == 0) return;
    const int n_tris = S.n_tris;
    const int tiles_x = P.W / TW;
    // The workgroups of one image take interleaved 16-row groups (group g of workgroup ty = rows 16 (g tiles_y + ty) ...), not contiguous
    // halves: they then meet the stimulus about equally, instead of one drawing all of it while the other finds its tile empty.
    const int tiles_y = P.H / TH;
    const int tile_x = (blockIdx.x % tiles_x) * TW, ty = blockIdx.x / tiles_x;
    const int tid = threadIdx.x;
    const int pass = (term_xform != nullptr && (int)blockIdx.z == P.term_layer) ? 0 : 1;
    if (pass == 0 && term_mask[env] == 0) return;
    const float* __restrict__ xf = pass == 0 ? term_xform : xform;
    uint8_t* __restrict__ img = pass == 0 ? term_out : out;
    float M[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = xform_soa ? xf[(size_t)k * n_envs + env] : xf[(size_t)env * 12 + k];
    // Lane -> pixel mapping: a wavefront visits its share of the tile as 16 x 16 pixel blocks (4 quad columns x 16 rows per pass, pass kk =
    // its (kk & 1)-th column block, (kk >> 1)-th row group).  A pass is skipped by the whole wavefront when no lane's
    // quad row meets the record, so compact blocks matter: with the earlier full-width mapping (a pass = a 128 x 2 strip) nearly every pass
    // met the stimulus somewhere and ran with most lanes idle (render 55.1 -> 43.5 us for the edge).
    static_assert(TW == 128 && TH == 64 && NK == 8, "16 x 16 blocks of a 128 x 64 tile");
    const bool interleave = tiles_y > 2;   // two workgroups per image (128 x 128) gain nothing: contiguous halves keep the tile's record cull
    // ... and wavefront w the 16-pixel column blocks w and w + 4 of the tile, not the adjacent pair 2w, 2w + 1: the workgroup lasts as long
    // as its busiest wavefront, and a stimulus that covers one side of the tile now lands on all four (edge render 43.3 -> 39.7 us; dealing
    // the blocks out diagonally as well measured slower: 43.6 us, the per-pass column no longer folds into the address).
    const int qx0 = tile_x + 16 * (tid / 64) + 4 * (tid % 4);
#define TG_QX(kk) (qx0 + 64 * ((kk) & 1))
    const int ry0 = (interleave ? 16 : TH) * ty + ((tid % 64) / 4), ry_step = interleave ? 16 * tiles_y : 16;
#define TG_RY(kk) (ry0 + ry_step * ((kk) >> 1))
    const float tx0 = (float)tile_x, tx1 = (float)(tile_x + TW);
    const float ty0 = interleave ? 0.0f : (float)(TH * ty), ty1 = interleave ? (float)P.H : (float)(TH * ty + TH);
    if (tid == 0) count = 0;
    bool beyond = true;
    if (S.closed_outward)
        for (int t = tid; t < n_tris; t += kThreads) beyond = beyond && tri_beyond_near(S.soup, t, M, P.near_);
    const bool cull = __syncthreads_and(beyond ? 1 : 0) != 0 && S.closed_outward != 0;   // (also the barrier after count = 0)
    for (int t = tid; t < n_tris; t += kThreads) {
        float cx[3], cy[3], cw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* v = S.soup + 9 * t + 3 * k;
            const float vx = v[0], vy = v[1], vz = v[2];
            cx[k] = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
            cy[k] = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
            cw[k] = -(((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11]);
        }
        if (cull && back_facing(cx, cy, cw)) continue;
        float ox[4], oy[4], ow[4];
        int no = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int k1 = (k + 1) % 3;
            const bool ain = cw[k] >= P.near_, bin = cw[k1] >= P.near_;
            if (ain) { ox[no] = cx[k]; oy[no] = cy[k]; ow[no] = cw[k]; ++no; }
            if (ain != bin) {
                const float tt = (P.near_ - cw[k]) / (cw[k1] - cw[k]);
                ox[no] = cx[k] + tt * (cx[k1] - cx[k]);
                oy[no] = cy[k] + tt * (cy[k1] - cy[k]);
                ow[no] = P.near_;
                ++no;
            }
        }
        if (no >= 3) emit(recs, &count, rec_cap, ox, oy, ow, 0, 1, 2, P, tx0, ty0, tx1, ty1);
        if (no == 4) emit(recs, &count, rec_cap, ox, oy, ow, 0, 2, 3, P, tx0, ty0, tx1, ty1);
    }
    __syncthreads();
    const int n = min(count, rec_cap);
    // Which pass blocks can a record cover a pixel of?  Lane l < 32 of each wavefront stands for block (bx = l & 7, by = l >> 3); bounding box
    // and the edge functions over the block's pixel centres (edges_exclude_rect: rigorous, so skipping changes no pixel).  The two coplanar
    // triangles of the plate's face (object_balance: they fill the view) each cover half of it, and so do a box face's.
    {
        const int wv = tid >> 6, l = tid & 63, bx = l & 7, by = (l >> 3) & 3;
        const float X0 = (float)(tile_x + 16 * bx) + 0.5f, X1 = X0 + 15.0f;
        const float Y0 = (float)((interleave ? 16 : TH) * ty + ry_step * by) + 0.5f, Y1 = Y0 + 15.0f;
        // ... and a block whose pixels are all pasted from the reference image (the TacTip's border ring: ~30 % of the 16 x 16 blocks of a
        // 256 x 256 image) shows nothing of its depths: no record needs to visit it (blockmax: the host's table of k_render_blocks, -1 there)
        float bmax_b = 3.0e38f;
        if (P.blockmax != nullptr) {
            const int gx = tile_x + 16 * bx, gy = (interleave ? 16 : TH) * ty + ry_step * by;
            bmax_b = P.blockmax[((gy / 128) * (P.W / 128) + gx / 128) * 64 + ((gy % 128) / 16) * 8 + (gx % 128) / 16];
        }
        for (int t = wv; t < n; t += kThreads / 64) {
            const TriRec r = recs[t];
            bool miss = (r.ymax < Y0) | (r.ymin > Y1) | (r.xmax < X0) | (r.xmin > X1) | (r.dmin >= bmax_b);
            miss = miss | edges_exclude_rect(r.x0, r.y0, r.x1, r.y1, r.x2, r.y2, X0, X1, Y0, Y1);
            const unsigned long long m = __ballot(!miss);
            const unsigned long long mi = QREJ ? 0ull : __ballot(!miss && edges_cover_rect(r.x0, r.y0, r.x1, r.y1, r.x2, r.y2, X0, X1, Y0, Y1));
            if (l == 0) { rblocks[t] = (unsigned)m; rinside[t] = (unsigned)mi; }
        }
        __syncthreads();
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float eps = 1e-4f, max_pen = 0.05f;
    uint8_t* dst = img + (size_t)env * P.W * P.H;
    uint8_t* prev = (save_prev && pass == 1) ? save_prev + (size_t)env * P.W * P.H : nullptr;
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
        float z[NKH][4];
        unsigned touched = 0;
#pragma unroll
        for (int k = 0; k < NKH; ++k) {
            const float4 nd = *reinterpret_cast<const float4*>(nodef_dep + (size_t)TG_RY(h * NKH + k) * P.W + TG_QX(h * NKH + k));
            z[k][0] = nd.x; z[k][1] = nd.y; z[k][2] = nd.z; z[k][3] = nd.w;
        }
        for (int t = 0; t < n; ++t) {
            const unsigned rb = __builtin_amdgcn_readfirstlane(rblocks[t]);
            if (!((rb >> wave_u) & 0x11111111u)) continue;      // none of this wavefront's eight blocks (columns wave, wave + 4 of the four row groups)
            const unsigned ri = __builtin_amdgcn_readfirstlane(rinside[t]);
            const TriRec r = recs[t];
#pragma unroll
            for (int k = 0; k < NKH; ++k) {
                const int bit = 8 * ((h * NKH + k) >> 1) + wave_u + 4 * ((h * NKH + k) & 1);
                if (!((rb >> bit) & 1u)) continue;   // scalar: this pass's block
                const int qx = TG_QX(h * NKH + k);
                const float fy = (float)TG_RY(h * NKH + k) + 0.5f;
                if (!QREJ && ((ri >> bit) & 1u)) {     // (the few-large-triangles instantiation only: elsewhere the second loop body costs registers)
                    // the record covers every pixel centre of this block (round 5; the plate's face under the TacTip: nearly every block of
                    // config 5): `box & (pos | neg) & (s != 0)` is true at every pixel, only the interpolated depth and its test remain
                    if (r.dmin >= fmaxf(fmaxf(z[k][0], z[k][1]), fmaxf(z[k][2], z[k][3]))) continue;
                    const float a0 = r.y2 - fy, a1 = r.y1 - fy, a2 = r.y0 - fy;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float fx = (float)(qx + p) + 0.5f;
                        const float e0 = (r.x1 - fx) * a0 - (r.x2 - fx) * a1;
                        const float e1 = (r.x2 - fx) * a2 - (r.x0 - fx) * a0;
                        const float e2 = (r.x0 - fx) * a1 - (r.x1 - fx) * a2;
                        const float s = (e0 + e1) + e2;
                        const float d = div_mid_range((e0 * r.d0 + e1 * r.d1) + e2 * r.d2, s);
                        const bool hit = d < z[k][p];
                        z[k][p] = hit ? d : z[k][p];
                        touched |= hit ? (1u << k) : 0u;
                    }
                    continue;
                }
                if (fy < r.ymin || fy > r.ymax) continue;
                if ((float)qx + 3.5f < r.xmin || (float)qx + 0.5f > r.xmax) continue;
                if (r.dmin >= fmaxf(fmaxf(z[k][0], z[k][1]), fmaxf(z[k][2], z[k][3]))) continue;
                const float a0 = r.y2 - fy, a1 = r.y1 - fy, a2 = r.y0 - fy;
                if (QREJ) {   // conservative reject of the 4-pixel quad (see k_render_tactile)
                    const float fc = (float)qx + 2.0f;
                    const float b0 = r.x0 - fc, b1 = r.x1 - fc, b2 = r.x2 - fc;
                    const float c0 = b1 * a0 - b2 * a1, c1 = b2 * a2 - b0 * a0, c2 = b0 * a1 - b1 * a2;
                    const float A0 = fabsf(a0), A1 = fabsf(a1), A2 = fabsf(a2);
                    const float B0 = fabsf(b0) + 1.5f, B1 = fabsf(b1) + 1.5f, B2 = fabsf(b2) + 1.5f;
                    const float m0 = 1e-5f * (B1 * A0 + B2 * A1), m1 = 1e-5f * (B2 * A2 + B0 * A0), m2 = 1e-5f * (B0 * A1 + B1 * A2);
                    const float h0 = 1.5f * fabsf(r.y1 - r.y2) + m0, h1 = 1.5f * fabsf(r.y2 - r.y0) + m1, h2 = 1.5f * fabsf(r.y0 - r.y1) + m2;
                    const float sc = (c0 + c1) + c2, ms = 4.0f * ((m0 + m1) + m2);
                    const bool out_pos = (sc > ms) & ((c0 < -h0) | (c1 < -h1) | (c2 < -h2));
                    const bool out_neg = (sc < -ms) & ((c0 > h0) | (c1 > h1) | (c2 > h2));
                    if (out_pos | out_neg) continue;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float fx = (float)(qx + p) + 0.5f;
                    const float e0 = (r.x1 - fx) * a0 - (r.x2 - fx) * a1;
                    const float e1 = (r.x2 - fx) * a2 - (r.x0 - fx) * a0;
                    const float e2 = (r.x0 - fx) * a1 - (r.x1 - fx) * a2;
                    const bool box = (fx >= r.xmin) & (fx <= r.xmax);
                    const bool pos = (e0 >= 0.0f) & (e1 >= 0.0f) & (e2 >= 0.0f), neg = (e0 <= 0.0f) & (e1 <= 0.0f) & (e2 <= 0.0f);
                    const float s = (e0 + e1) + e2;
                    const float d = div_mid_range((e0 * r.d0 + e1 * r.d1) + e2 * r.d2, s);
                    const bool hit = box & (pos | neg) & (s != 0.0f) & (d < z[k][p]);
                    z[k][p] = hit ? d : z[k][p];
                    touched |= hit ? (1u << k) : 0u;
                }
            }
        }
        uchar4 ngk[NKH], bmk[NKH];
        float4 ndk[NKH];
#pragma unroll
        for (int k = 0; k < NKH; ++k) {
            const size_t off = (size_t)TG_RY(h * NKH + k) * P.W + TG_QX(h * NKH + k);
            ngk[k] = *reinterpret_cast<const uchar4*>(gray_u8 + off);
            bmk[k] = *reinterpret_cast<const uchar4*>(border + off);
            ndk[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if ((touched >> k) & 1u) ndk[k] = *reinterpret_cast<const float4*>(nodef_dep + off);
        }
#pragma unroll
        for (int k = 0; k < NKH; ++k) {
            const size_t off = (size_t)TG_RY(h * NKH + k) * P.W + TG_QX(h * NKH + k);
            const uint8_t ngv[4] = {ngk[k].x, ngk[k].y, ngk[k].z, ngk[k].w}, bmv[4] = {bmk[k].x, bmk[k].y, bmk[k].z, bmk[k].w};
            const float ndv[4] = {ndk[k].x, ndk[k].y, ndk[k].z, ndk[k].w};
            uint8_t o[4] = {0, 0, 0, 0};
            if ((touched >> k) & 1u) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float diff = z[k][p] - ndv[p];
                    if (diff >= -eps && diff <= eps) diff = 0.0f;
                    const float pen = fabsf(diff);
                    const float cl = pen < 0.0f ? 0.0f : (pen > max_pen ? max_pen : pen);
                    o[p] = (uint8_t)(div_mid_range(cl, max_pen) * 255.0f);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (!P.turn_off_border && bmv[p] == 1) o[p] = ngv[p];
            if (prev) *reinterpret_cast<uchar4*>(prev + off) = *reinterpret_cast<const uchar4*>(dst + off);
            *reinterpret_cast<uchar4*>(dst + off) = make_uchar4(o[0], o[1], o[2], o[3]);
        }
    }
}
#undef TG_QX
#undef TG_RY

#ifdef TG_BLK_STAMPS
// development only: per-phase clock stamps of k_render_blocks' last launch (first 1024 workgroups x 4 wavefronts x 8 stamps, absolute wall
// clock, 100 MHz), printed by raster_debug_stats()
__device__ unsigned long long g_blk_log[1024 * 4 * 8];
__device__ int g_blk_nt[1024 * 2];
#define TG_STAMP(i) do { if (lane == 0 && blockIdx.y < 1024 && blockIdx.x == 0 && blockIdx.z == 0) g_blk_log[(blockIdx.y * 4 + wave) * 8 + (i)] = wall_clock64(); } while (0)
void raster_debug_stats() {
    static unsigned long long h[1024 * 4 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_blk_log), sizeof h) != hipSuccess) return;
    unsigned long long t0 = ~0ull, t1 = 0;
    int nw = 0;
    for (int w = 0; w < 4096; ++w) if (h[w * 8 + 7]) { ++nw; if (h[w * 8 + 7] < t0) t0 = h[w * 8 + 7]; for (int i = 0; i < 7; ++i) if (h[w * 8 + i] > t1) t1 = h[w * 8 + i]; }
    if (!nw) return;
    fprintf(stderr, "k_render_blocks last launch: %d wavefronts, span %llu0 ns; per phase (x10 ns) mean since own start / mean since launch start / max since launch start:\n", nw, t1 - t0);
    for (int i = 0; i < 7; ++i) {
        double a = 0, b2 = 0; unsigned long long m = 0;
        for (int w = 0; w < 4096; ++w) if (h[w * 8 + 7] && h[w * 8 + i]) { a += (double)(h[w * 8 + i] - h[w * 8 + 7]); b2 += (double)(h[w * 8 + i] - t0); if (h[w * 8 + i] - t0 > m) m = h[w * 8 + i] - t0; }
        fprintf(stderr, "  [%d] %.1f / %.1f / %llu\n", i, a / nw, b2 / nw, m);
    }
    double st = 0; unsigned long long sm = 0;
    for (int w = 0; w < 4096; ++w) if (h[w * 8 + 7]) { st += (double)(h[w * 8 + 7] - t0); if (h[w * 8 + 7] - t0 > sm) sm = h[w * 8 + 7] - t0; }
    fprintf(stderr, "  start: mean %.1f max %llu\n", st / nw, sm);
    static int nt[2048];
    if (hipMemcpyFromSymbol(nt, HIP_SYMBOL(g_blk_nt), sizeof nt) != hipSuccess) return;
    // per workgroup: end time (max over its wavefronts) against records and reached blocks
    fprintf(stderr, "  wg: end(x10ns) n T   (sorted by end, every 64th)\n");
    static int order[1024]; static unsigned long long endt[1024];
    for (int g = 0; g < 1024; ++g) { order[g] = g; endt[g] = 0; for (int w = 0; w < 4; ++w) if (h[(g * 4 + w) * 8 + 6] > endt[g]) endt[g] = h[(g * 4 + w) * 8 + 6]; endt[g] -= t0; }
    for (int i = 0; i < 1024; ++i) for (int j = i + 1; j < 1024; ++j) if (endt[order[j]] < endt[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int i = 0; i < 1024; i += 64) fprintf(stderr, "   %llu %d %d |", endt[order[i]], nt[2 * order[i]] & 255, nt[2 * order[i] + 1]);
    fprintf(stderr, "   %llu %d %d\n", endt[order[1023]], nt[2 * order[1023]] & 255, nt[2 * order[1023] + 1]);
    fprintf(stderr, "  slowest 40 (end n T xcc se sh cu):");
    for (int i = 1023; i > 983; --i) { const int v = nt[2 * order[i]]; fprintf(stderr, " %llu/%d/%d/x%d.s%d.h%d.c%d", endt[order[i]], v & 255, nt[2 * order[i] + 1], (v >> 8) & 15, (v >> 16) & 7, (v >> 20) & 1, (v >> 12) & 15); }
    fprintf(stderr, "\n  mean end per xcc:");
    for (int x = 0; x < 8; ++x) { double a = 0; int c = 0; for (int g = 0; g < 1024; ++g) if (((nt[2 * g] >> 8) & 15) == x) { a += (double)endt[g]; ++c; } fprintf(stderr, " x%d %.0f (%d)", x, c ? a / c : 0.0, c); }
    fprintf(stderr, "\n  workgroups per (xcc, se, sh, cu): ");
    { static int cnt[8 * 8 * 2 * 16]; for (int g = 0; g < 1024; ++g) { const int v = nt[2 * g]; ++cnt[((((v >> 8) & 15) * 8 + ((v >> 16) & 7)) * 2 + ((v >> 20) & 1)) * 16 + ((v >> 12) & 15)]; }
      int hist[16] = {0}; for (int i = 0; i < 8 * 8 * 2 * 16; ++i) if (cnt[i] < 16) ++hist[cnt[i]]; for (int i = 0; i < 12; ++i) fprintf(stderr, " %d:%d", i, hist[i]); }
    fprintf(stderr, "\n");
    double an = 0, aT = 0; for (int g = 0; g < 1024; ++g) { an += nt[2 * g] & 255; aT += nt[2 * g + 1]; }
    fprintf(stderr, "  mean n %.2f mean T %.2f\n", an / 1024, aT / 1024);
}
#else
#define TG_STAMP(i) do { } while (0)
void raster_debug_stats() {}
#endif

// Small shared meshes, block form (round 3): ONE workgroup per (env, 128 x 128 region) sets the triangles up once (at most 64 records:
// meshes of up to 32 triangles - edge, cube).  The region is 64 blocks of BW x (256 / BW) pixels.
//   1. lane-as-record: every lane keeps one record's bounding box, dmin and DEPTH PLANE.  Window depth is affine over a triangle,
//      D(x, y) = d0 + A (x - x0) + B (y - y0), so its minimum over (block rectangle) n (record bounding box) sits at a corner.
//   2. lane-as-block: wavefront w broadcasts the records w, w + 4, ... (v_readlane) and each lane decides whether the record can change a
//      pixel of "its" block: bounding box, dmin against the block's largest undeformed depth (a host-made table; pasted ring pixels do not
//      count), and the plane minimum less a margin (rounding of this estimate against the pixel formula, scaled by the triangle's
//      conditioning = bounding-box area / twice its area; ill-conditioned records have no plane) against that depth less kGrey: t_s_camera
//      maps a depth less than 0.05 / 255 = 1.96e-4 below the undeformed one to grey level 0, which is what the untouched-sensor image
//      holds.  The ballots (per record, in LDS) and their union are the region's reach masks.
//   3. only CHANGED blocks are written: RasterParams::drawn records per image which blocks do not hold the untouched-sensor image; blocks
//      reached now are drawn, blocks drawn by the launch before and not reached now are restored (16-byte words, whole 128-byte lines),
//      everything else is left alone - the observation buffer is read-only for the caller (TG_RASTER_REWRITE_ALL=1: no record).
//   4. reached blocks are drawn by ALL four wavefronts together, 4 NQ = 12 blocks per round: wavefront w owns the w-th quarter of the rows
//      of each, a lane NQ quads (one in each of NQ block groups) - whatever the contact patch looks like, the four wavefronts finish
//      together, and the lanes of a visited record are in compact patches that mostly hit.  Records that reach none of a group's blocks
//      are skipped as scalar branches; a quad whose depth plane is nowhere in front of what it holds is skipped per lane.
// Against k_render_small (two workgroups per image, each lane carrying 8 quads spread over its half): half the set-ups, 4096 instead of
// 8192 wavefronts at 1024 envs, no wavefront whose share of the image is the whole contact patch, a third of the HBM traffic.  The pixel
// arithmetic is that of the other kernels, expression by expression; the depth test keeps the smallest d, so neither the record order
// nor the conservative skips can change the image.  DESIGN.md 4.2 has the measurements and the per-phase timeline.
#ifndef TG_BLK_NQ
#define TG_BLK_NQ 3
#endif
template <int BW>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_render_blocks(RasterParams P, Stimulus S, const float* __restrict__ xform, int xform_soa, int n_envs,
                                                           const uint8_t* __restrict__ mask, const float* __restrict__ nodef_dep,
                                                           const uint8_t* __restrict__ gray_u8, const uint8_t* __restrict__ border,
                                                           uint8_t* __restrict__ out, uint8_t* __restrict__ save_prev, int rec_cap,
                                                           const float* __restrict__ term_xform, const uint8_t* __restrict__ term_mask,
                                                           uint8_t* __restrict__ term_out) {
    KtScope kt_scope_(P.kt);
    constexpr int BH = 256 / BW, LPR = BW / 4, NBX = 128 / BW, NPW = 16, RQ = BH / 4;
    constexpr int NQ = TG_BLK_NQ;   // quads per lane in a drawing round: 4 NQ blocks per round   // NPW blocks per wavefront; RQ rows per block quarter
    static_assert(BW == kBlockW, "P.blockmax is laid out for kBlockW");
    extern __shared__ TriRec recs[];
    __shared__ int count;
    __shared__ unsigned long long reach_rec[64], reach_all;    // per record: the blocks it can change; their union
    const int env = blockIdx.y;
    if (mask != nullptr && mask[env] == 0) return;
    const int n_tris = S.n_tris;
    const int regions_x = P.W / 128;
    const int reg = blockIdx.x, rx = (reg % regions_x) * 128, ry = (reg / regions_x) * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TG_STAMP(7);
#ifdef TG_TL_STAMPS
    if (P.tl != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { const unsigned long long i_ = atomicAdd(P.tl + 4 * 8192 + 3, 1ull); P.tl[3 * 8192 + (i_ & 8191)] = wall_clock64(); }
#endif
    const int pass = (term_xform != nullptr && (int)blockIdx.z == P.term_layer) ? 0 : 1;
    if (pass == 0 && term_mask[env] == 0) return;
    const float* __restrict__ xf = pass == 0 ? term_xform : xform;
    uint8_t* __restrict__ img = pass == 0 ? term_out : out;
    const float bmax_l = P.blockmax[reg * 64 + lane];          // lane l <-> block l: its largest undeformed depth
    // the blocks of this image that do NOT hold the untouched-sensor image now (drawn by the previous launch on it); the terminal image's
    // buffer has no such record: everything is rewritten there
    unsigned long long* drawn_p = (P.drawn != nullptr && pass == 1) ? P.drawn + ((size_t)env * gridDim.x + reg) : nullptr;
    unsigned long long stale = ~0ull;
    if (drawn_p) stale = *drawn_p;
    float M[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = xform_soa ? xf[(size_t)k * n_envs + env] : xf[(size_t)env * 12 + k];
    const float tx0 = (float)rx, tx1 = (float)(rx + 128), ty0 = (float)ry, ty1 = (float)(ry + 128);
    // set-up: at most one triangle per lane, all in wavefront 0 (n_tris <= 32), so the vote that licenses the back-face cull (every vertex
    // of the mesh beyond the near plane) is a ballot, and the record counter is cleared by the wavefront that then counts (LDS operations
    // of one wavefront execute in order) - one workgroup barrier instead of four
    if (tid == 0) { count = 0; reach_all = 0ull; }
    const bool has = tid < n_tris;                                 // (lanes of wavefront 0)
    float cx[3] = {0.0f, 0.0f, 0.0f}, cy[3] = {0.0f, 0.0f, 0.0f}, cw[3] = {0.0f, 0.0f, 0.0f};
    bool beyond = true;
    if (has) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* v = S.soup + 9 * tid + 3 * k;
            const float vx = v[0], vy = v[1], vz = v[2];
            cx[k] = ((M[0] * vx + M[1] * vy) + M[2] * vz) + M[9];
            cy[k] = ((M[3] * vx + M[4] * vy) + M[5] * vz) + M[10];
            cw[k] = -(((M[6] * vx + M[7] * vy) + M[8] * vz) + M[11]);
            beyond = beyond && (cw[k] >= P.near_);
        }
    }
    TG_STAMP(0);
    const bool cull = __ballot(!beyond) == 0ull && S.closed_outward != 0;      // wavefront 0: the vote; the others have nothing to cull
    TG_STAMP(1);
    if (has && !(cull && back_facing(cx, cy, cw))) {
        if ((cw[0] >= P.near_) & (cw[1] >= P.near_) & (cw[2] >= P.near_)) {       // the usual case: nothing to clip (the loop below would
            emit(recs, &count, rec_cap, cx, cy, cw, 0, 1, 2, P, tx0, ty0, tx1, ty1);   // hand over the same three vertices in the same order)
        } else {
            // near-plane clip, written out per case so that no array is indexed by a run-time count (that costs a scratch allocation): the
            // vertex sequences are those of the loop in k_render_tactile - for edge k -> k + 1: vertex k if inside, then the crossing
            const bool i0 = cw[0] >= P.near_, i1 = cw[1] >= P.near_, i2 = cw[2] >= P.near_;
            float ix[3], iy[3];                                   // crossing of edge k -> (k + 1) % 3
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
            if (no >= 3) emit(recs, &count, rec_cap, ox, oy, ow, 0, 1, 2, P, tx0, ty0, tx1, ty1);
            if (no == 4) emit(recs, &count, rec_cap, ox, oy, ow, 0, 2, 3, P, tx0, ty0, tx1, ty1);
        }
    }
    __syncthreads();
    TG_STAMP(2);
    const int n = __builtin_amdgcn_readfirstlane(min(count, rec_cap));     // <= 64 (launch_render)
    const float eps = 1e-4f, max_pen = 0.05f;
    constexpr float kGrey = 1.5e-4f;
    uint8_t* dst = img + (size_t)env * P.W * P.H;
    uint8_t* prev = (save_prev && pass == 1) ? save_prev + (size_t)env * P.W * P.H : nullptr;

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
        if (cond < 64.0f) {                             // (NaN and inf compare false: no plane for degenerate records)
            q_A = (ud * vy - vd * uy) / ar; q_B = (ux * vd - vx * ud) / ar;
            q_mg = 2e-5f * cond;
        }
    }
    // 2. lane-as-block: which of the 64 blocks can record t change?  Wavefront w takes the records w, w + 4, ... (all four doing all of them
    //    was a third of this kernel's vector instructions: the four wavefronts of a SIMD share its issue slots)
    unsigned long long reached;
    {
        const float X0 = (float)(rx + (lane % NBX) * BW) + 0.5f, Y0 = (float)(ry + (lane / NBX) * BH) + 0.5f;   // first / last pixel centres
        const float X1 = X0 + (float)(BW - 1), Y1 = Y0 + (float)(BH - 1);
        unsigned long long part = 0ull;
#define TG_RL(v) __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), t))
        for (int t = wave; t < n; t += 4) {
            const float xl = TG_RL(q_xl), xh = TG_RL(q_xh), yl = TG_RL(q_yl), yh = TG_RL(q_yh), dm = TG_RL(q_dm);
            const float x0 = TG_RL(q_x0), y0 = TG_RL(q_y0), d0 = TG_RL(q_d0), A = TG_RL(q_A), B = TG_RL(q_B), mg = TG_RL(q_mg);
            bool miss = (yh < Y0) | (yl > Y1) | (xh < X0) | (xl > X1) | (dm >= bmax_l);
            const float xa = fmaxf(X0, xl), xb = fminf(X1, xh), ya = fmaxf(Y0, yl), yb = fminf(Y1, yh);
            const float low = (d0 + A * ((A >= 0.0f ? xa : xb) - x0)) + B * ((B >= 0.0f ? ya : yb) - y0);
            miss = miss | (low - mg >= bmax_l - kGrey);
            // ... and whether the triangle itself (not just its bounding box and plane) can cover a pixel centre of the block
            if (kEdgeReach) miss = miss | edges_exclude_rect(x0, y0, TG_RL(q_x1), TG_RL(q_y1), TG_RL(q_x2), TG_RL(q_y2), X0, X1, Y0, Y1);
            const unsigned long long m = __ballot(!miss);
            part |= m;
            if (lane == 0) reach_rec[t] = m;
        }
        if (lane == 0 && part) atomicOr(&reach_all, part);
        __syncthreads();
        reached = reach_all;
    }
    reached = __builtin_amdgcn_readfirstlane((unsigned)reached) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(reached >> 32)) << 32);
    stale = __builtin_amdgcn_readfirstlane((unsigned)stale) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(stale >> 32)) << 32);
    if (drawn_p && tid == 0) *drawn_p = reached;
    TG_STAMP(3);
#ifdef TG_BLK_STAMPS
    if (tid == 0 && blockIdx.y < 1024 && blockIdx.x == 0 && blockIdx.z == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g_blk_nt[2 * blockIdx.y] = n | ((xcc & 15) << 8) | (((hw >> 8) & 15) << 12) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 20);
        g_blk_nt[2 * blockIdx.y + 1] = __builtin_popcountll(reached);
    }
#endif
    const unsigned long long restore = stale & ~reached;      // blocks to bring back to the untouched-sensor image
    const int lq = lane & 15, lrow = lq / LPR, lcol = 4 * (lq % LPR);     // this lane's quad inside a block quarter
    // 3. blocks nothing reaches show the untouched sensor's image.  A lane moves one 16-pixel row of one block as a 16-byte word; the lanes of
    //    an instruction are the NBX blocks across the region x (64 / NBX) consecutive rows, so an instruction writes whole 128-byte lines
    //    wherever a line's blocks are all unreached (4-byte words in 16 x 16 patterns measured 10.6 us for the 16.8 MB of 1024 envs).
    //    Wavefront w owns the row groups w, w + 4, ...
    static_assert(BW == 16, "one 16-byte word per block row");
    constexpr int RPI = 64 / NBX, NCP = 128 / RPI / 4;          // rows per instruction; copy instructions per wavefront
    const int cbx = lane % NBX, crow0 = lane / NBX;
#define TG_CROW(i) (RPI * (4 * (i) + wave) + crow0)             /* row of the region this lane moves in its i-th instruction */
#define TG_COFF(i) ((size_t)(ry + TG_CROW(i)) * P.W + (rx + cbx * BW))
#define TG_CLEAN(i) (!((reached >> ((TG_CROW(i) / BH) * NBX + cbx)) & 1ull))
#define TG_RESTORE(i) ((restore >> ((TG_CROW(i) / BH) * NBX + cbx)) & 1ull)
    if (prev) {
        uint4 pv[NCP];
#pragma unroll
        for (int i = 0; i < NCP; ++i)
            pv[i] = *reinterpret_cast<const uint4*>(dst + TG_COFF(i));      // (every address is valid: only the stores are predicated)
#pragma unroll
        for (int i = 0; i < NCP; ++i)
            if (TG_CLEAN(i)) *reinterpret_cast<uint4*>(prev + TG_COFF(i)) = pv[i];
    }
    if (restore) {   // (rare once a contact is established: blocks the contact patch has left)
        uint4 tv[NCP];
#pragma unroll
        for (int i = 0; i < NCP; ++i)
            tv[i] = *reinterpret_cast<const uint4*>(P.tmpl + TG_COFF(i));
#pragma unroll
        for (int i = 0; i < NCP; ++i)
            if (TG_RESTORE(i)) *reinterpret_cast<uint4*>(dst + TG_COFF(i)) = tv[i];
    }
    // 4. reached blocks, 4 NQ per round, every wavefront a quarter of each
    TG_STAMP(4);
    unsigned long long left = reached;
    while (left) {
        int bq[NQ];
        unsigned long long grp[NQ], grp_any = 0ull;   // the blocks of this round's groups (four blocks each: a lane has one quad in each group)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            int ids[4];
            const unsigned long long before = left;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ids[k] = left ? (int)__builtin_ctzll(left) : -1;
                left = left ? (left & (left - 1ull)) : 0ull;
            }
            grp[j] = before & ~left;
            grp_any |= grp[j];
            const int sub = lane >> 4;
            bq[j] = sub == 0 ? ids[0] : (sub == 1 ? ids[1] : (sub == 2 ? ids[2] : ids[3]));
        }
        size_t off[NQ];
        int qx[NQ];
        float fy[NQ], z[NQ][4];
        uchar4 ng[NQ], bmk[NQ], old[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int b = bq[j] < 0 ? 0 : bq[j];
            const int y = ry + (b / NBX) * BH + RQ * wave + lrow;
            qx[j] = rx + (b % NBX) * BW + lcol;
            fy[j] = (float)y + 0.5f;
            off[j] = (size_t)y * P.W + qx[j];
            float4 nd = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);  // no block: nothing passes the depth cull
            ng[j] = make_uchar4(0, 0, 0, 0); bmk[j] = ng[j]; old[j] = ng[j];
            if (bq[j] >= 0) {
                nd = *reinterpret_cast<const float4*>(nodef_dep + off[j]);
                ng[j] = *reinterpret_cast<const uchar4*>(gray_u8 + off[j]);
                bmk[j] = *reinterpret_cast<const uchar4*>(border + off[j]);
                if (prev) old[j] = *reinterpret_cast<const uchar4*>(dst + off[j]);
            }
            z[j][0] = nd.x; z[j][1] = nd.y; z[j][2] = nd.z; z[j][3] = nd.w;
        }
        for (int t = 0; t < n; ++t) {
            unsigned long long mt = reach_rec[t];
            mt = __builtin_amdgcn_readfirstlane((unsigned)mt) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mt >> 32)) << 32);
            if (!(mt & grp_any)) continue;            // the record reaches none of this round's blocks
            const TriRec r = recs[t];
            const float rA = TG_RL(q_A), rB = TG_RL(q_B), rmg = TG_RL(q_mg);
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (!(mt & grp[j])) continue;
                if (fy[j] < r.ymin || fy[j] > r.ymax) continue;
                if ((float)qx[j] + 3.5f < r.xmin || (float)qx[j] + 0.5f > r.xmax) continue;
                const float zmax = fmaxf(fmaxf(z[j][0], z[j][1]), fmaxf(z[j][2], z[j][3]));
                if (r.dmin >= zmax) continue;
                // the record's depth plane over the quad's four pixel centres (see 1.): nowhere in front of what the quad holds -> no pixel can pass d < z
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
        // the undeformed depths again (L2) rather than four more registers per quad held through the record loop (112 -> 100 VGPRs at NQ = 3;
        // NQ = 4 - sixteen blocks per round - fits four wavefronts per SIMD this way and measured the same: a workgroup whose contact patch is
        // twice the average has twice the arithmetic however its rounds are cut)
        float4 nd2[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            nd2[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (bq[j] >= 0) nd2[j] = *reinterpret_cast<const float4*>(nodef_dep + off[j]);
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
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
            if (prev) *reinterpret_cast<uchar4*>(prev + off[j]) = old[j];
            *reinterpret_cast<uchar4*>(dst + off[j]) = make_uchar4(o[0], o[1], o[2], o[3]);
        }
    }
    TG_STAMP(5);
    TG_STAMP(6);
#undef TG_RESTORE
#undef TG_RL
#undef TG_COFF
#undef TG_CROW
#undef TG_CLEAN
}

void make_gray_u8(const float* nodef_gray_host, int npix, uint8_t* out_host) {
    for (int i = 0; i < npix; ++i) out_host[i] = (uint8_t)nodef_gray_host[i];   // the truncating cast of tactile_sensor.py:291-292
}

void launch_render(const RasterParams& P, const Stimulus& S_in, const float* xform, int xform_soa, int n_envs,
                   const uint8_t* mask, const float* nodef_dep, const uint8_t* gray_u8, const uint8_t* border, uint8_t* out,
                   uint8_t* save_prev, const float* term_xform, const uint8_t* term_mask, uint8_t* term_out, hipStream_t stream) {
    Stimulus S = S_in;
    S.win_side = 0;
    if (S.kind == 1) {
        // side of the frustum window (k_render_tactile) for any camera orientation: the truncated pyramid lies within the sphere of radius
        // |apex - far corner| about the camera, so its xy box is at most 2 R wide; + 2 x 2 cells widening, + floor / ceil, + 1 (vertices)
        const double w_cull = 1.01 * ((double)P.C1 / (((double)P.zcull + 2.0 * (double)kDepthSlack) - (double)P.C0));
        const double ex = w_cull * ((double)P.hw / (double)P.kx), ey = w_cull * ((double)P.hh / (double)P.ky);
        const double R = std::sqrt(ex * ex + ey * ey + w_cull * w_cull);
        int side = (int)std::ceil(2.0 * R / (double)S.scale) + 8;
        const int full = S.rows > S.cols ? S.rows : S.cols;
        S.win_side = (side < full && side > 0) ? side : full;
    }
    int rec_cap = 2 * S.n_tris;
    static const bool scatter_off = getenv("TG_NO_SCATTER_RASTER") != nullptr;   // A/B switch for the parity test
    rec_cap = rec_cap > kBatch ? kBatch : (rec_cap < 2 ? 2 : rec_cap);
    if (S.kind == 1 && rec_cap > 256) rec_cap = 256;   // a dozen heightfield triangles survive the depth cull; more just take another round
    const size_t wcap = S.kind == 1 ? (((size_t)S.win_side * S.win_side + 3) & ~(size_t)3) : 0;
    const size_t lds = (size_t)rec_cap * sizeof(TriRec) + (S.kind == 1 ? wcap * (2 * sizeof(float) + 2 * sizeof(unsigned short) + 1) + (size_t)rec_cap * sizeof(unsigned) + 16 : 0);
    if (P.W % 128 == 0 && P.H % 128 == 0) {
        if (S.kind == 0 && S.n_tris <= 256 && rec_cap >= 2 * S.n_tris) {
            // a small shared mesh (edge, cube, pole): every triangle fits the record buffer in one round -> the two-pass kernel with
            // 16 x 16 pass blocks on 128 x 64 tiles, whatever the launch size (16 384 envs: 0.75 -> 0.42 ms against 128 x 128 tiles)
            // up to 32 triangles whose image is mostly the untouched sensor's (edge, cube): the block kernel; a stimulus that fills the view
            // (the pole's plate: every block is drawn, nothing to skip) stays with the two-pass kernel below (256 x 256: 0.151 against 0.157 ms)
            static const bool blocks_off = getenv("TG_NO_BLOCK_RASTER") != nullptr;   // A/B switch (parity test, measurements)
            if (P.blockmax != nullptr && P.tmpl != nullptr && S.n_tris <= 32 && !S.fills_view && !blocks_off) {
                dim3 gb((P.W / 128) * (P.H / 128), n_envs, term_xform ? 2 : 1);
                hipLaunchKernelGGL((k_render_blocks<kBlockW>), gb, dim3(kThreads), lds, stream, P, S, xform, xform_soa, n_envs, mask,
                                   nodef_dep, gray_u8, border, out, save_prev, rec_cap, term_xform, term_mask, term_out);
                return;
            }
            // (the terminal image as a second pass of the env's own workgroups instead of a grid layer of its own - 8 192 workgroups per launch
            //  that start, look and leave - was measured in round 5: the pass loop costs the kernel more registers than the layer costs time,
            //  render 108 -> 125 us at 256 x 256; not kept)
            dim3 grid((P.W / 128) * (P.H / 64), n_envs, term_xform ? 2 : 1);
            if (S.skip_quad_reject)
                hipLaunchKernelGGL((k_render_small<128, 64, 2, false>), grid, dim3(kThreads), lds, stream, P, S, xform, xform_soa, n_envs, mask,
                                   nodef_dep, gray_u8, border, out, save_prev, rec_cap, term_xform, term_mask, term_out);
            else
                hipLaunchKernelGGL((k_render_small<128, 64, 2, true>), grid, dim3(kThreads), lds, stream, P, S, xform, xform_soa, n_envs, mask,
                                   nodef_dep, gray_u8, border, out, save_prev, rec_cap, term_xform, term_mask, term_out);
        } else {
            dim3 grid((P.W / 128) * (P.H / 128), n_envs, term_xform ? 2 : 1);
            if (S.kind == 1) {   // heightfield: 128 x 64 tiles (the per-workgroup staging is cheap since it is windowed: 0.108 -> 0.098 ms)
                dim3 g2((P.W / 128) * (P.H / 64), n_envs, term_xform ? 2 : 1);
                static const bool dbg_win = getenv("TG_DEBUG_WIN") != nullptr;
                if (dbg_win) fprintf(stderr, "heightfield window side %d cells\n", S.win_side);
                // a narrow view (DIGIT over the horizontal surface: a window of a few cells) leaves few records per tile: there the per-record
                // cell masks pay (render 63 -> 57 us); a wide one (TacTip over the vertical surface) has hundreds of records and keeps the
                // bounding-box bands - as its own instantiation, the cell code costs that kernel registers even when it is switched off
                if (kEdgeReach && S.win_side <= kCellsWinSide)
                    hipLaunchKernelGGL((k_render_tactile<128, 64, true, true>), g2, dim3(kThreads), lds, stream, P, S, xform, xform_soa, n_envs, mask,
                                       nodef_dep, gray_u8, border, out, save_prev, rec_cap, term_xform, term_mask, term_out);
                else
                hipLaunchKernelGGL((k_render_tactile<128, 64, true>), g2, dim3(kThreads), lds, stream, P, S, xform, xform_soa, n_envs, mask,
                                   nodef_dep, gray_u8, border, out, save_prev, rec_cap, term_xform, term_mask, term_out);
            } else if (!scatter_off)   // a shared mesh of many small triangles (the marble): triangle-parallel, LDS z-buffer
                hipLaunchKernelGGL((k_render_scatter<128, 128>), grid, dim3(kThreads), 0, stream, P, S, xform, xform_soa, n_envs, mask,
                                   nodef_dep, gray_u8, border, out, save_prev, term_xform, term_mask, term_out);
            else
                hipLaunchKernelGGL((k_render_tactile<128, 128, false>), grid, dim3(kThreads), lds, stream, P, S, xform, xform_soa, n_envs, mask,
                                   nodef_dep, gray_u8, border, out, save_prev, rec_cap, term_xform, term_mask, term_out);
        }
    } else {  // 64x64 images
        dim3 grid((P.W / 64) * (P.H / 64), n_envs, term_xform ? 2 : 1);
        if (S.kind == 0 && S.n_tris > 256 && !scatter_off) {
            hipLaunchKernelGGL((k_render_scatter<64, 64>), grid, dim3(kThreads), 0, stream, P, S, xform, xform_soa, n_envs, mask,
                               nodef_dep, gray_u8, border, out, save_prev, term_xform, term_mask, term_out);
            return;
        }
        hipLaunchKernelGGL((k_render_tactile<64, 64, false>), grid, dim3(kThreads), lds, stream, P, S, xform, xform_soa, n_envs, mask,
                           nodef_dep, gray_u8, border, out, save_prev, rec_cap, term_xform, term_mask, term_out);
    }
}

}  // namespace tg
