// tg_selftest.hip - libtactile_gym_hip_test.so: device self-tests of pieces of the product's kernels against brute force / the compiler's own
// arithmetic.  TEST INFRASTRUCTURE: built next to the product library from the same headers (tg_raster_dev.hpp, tg_narrowphase.hpp), linked
// into nothing that ships; include/tactile_gym_hip_test.h declares its three entry points, tests/ are its only callers
// (tests/test_gpu_parity.py, tests/test_gpu_narrowphase.py).  Until round 4 these lived in libtactile_gym_hip.so and its ABI.
// Compiled with -ffp-contract=off like tg_raster.hip (the raster's expressions are the subject).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tactile_gym_hip_test.h"
#include "tg_raster_dev.hpp"

namespace tg {

// tg_selftest_division: the refinement above against the compiler's correctly rounded `/` on pseudo-random operand pairs with exponents in
// 2^-40 .. 2^24 (a superset of what a covered pixel produces); counts the pairs whose quotient bits differ.
__global__ void k_selftest_division(long long n, unsigned long long seed, unsigned long long* mismatches) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (long long i = i0; i < n; i += stride) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const unsigned ea = 127u - 40u + (unsigned)((z >> 46) & 0xff) % 65u, eb = 127u - 40u + (unsigned)((z >> 54) & 0xff) % 65u;
        const float a = __uint_as_float(((unsigned)(z >> 63) << 31) | (ea << 23) | ((unsigned)z & 0x7fffffu));
        const float b = __uint_as_float(((unsigned)((z >> 62) & 1) << 31) | (eb << 23) | ((unsigned)(z >> 23) & 0x7fffffu));
        bad += __float_as_uint(div_mid_range(a, b)) != __float_as_uint(a / b);
    }
    if (bad) atomicAdd(mismatches, bad);
}
int selftest_division(long long n, unsigned long long seed, long long* mismatches_host) {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) return -1;
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k_selftest_division, dim3(2048), dim3(256), 0, 0, n, seed, d);
    const hipError_t e = hipMemcpy(mismatches_host, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? 0 : -1;
}

// tg_selftest_penetration_division: t_s_camera's (penetration / 0.05) - the division of tactile_sensor.py:284-289 as the kernels evaluate it since
// round 5, div_mid_range(cl, 0.05f) - against the correctly rounded `/` for EVERY float a penetration can be: 0 and all of [1e-4, 0.05].
__global__ void k_selftest_penetration_division(unsigned long long* mismatches) {
    const unsigned lo = __float_as_uint(1e-4f), hi = __float_as_uint(0.05f);
    const unsigned long long n = (unsigned long long)(hi - lo) + 2ull, stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float a = i == n - 1 ? 0.0f : __uint_as_float(lo + (unsigned)i);
        const float max_pen = 0.05f;
        bad += __float_as_uint(div_mid_range(a, max_pen)) != __float_as_uint(a / max_pen);
    }
    if (bad) atomicAdd(mismatches, bad);
}
int selftest_penetration_division(long long* mismatches_host) {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) return -1;
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k_selftest_penetration_division, dim3(4096), dim3(256), 0, 0, d);
    const hipError_t e = hipMemcpy(mismatches_host, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? 0 : -1;
}

// tg_selftest_edge_exclusion: edges_exclude_rect against brute force.  Pseudo-random triangles in window coordinates - image-sized, slivers
// (third vertex a hair off the line of the other two), huge (coordinates up to 1e4), vertices snapped onto pixel centres - and rectangles
// as the kernels pass them (16 x 16 blocks, 32 x 8 cells, at pixel-centre coordinates within 256 x 256); every pixel centre of the rectangle
// is put through the pixel loops' own edge expressions.  out[0] = cases where the rule excluded a rectangle that holds a pixel with all three
// edge functions >= 0 or all <= 0 (must be 0: a superset of `hit`), out[1] = rectangles excluded, out[2] = rectangles that held no such pixel.
// Round 5, edges_cover_rect (the renders' "this block is wholly inside the triangle: no coverage test per pixel"): out[3] = rectangles it called
// covered in which some pixel fails the pixel loops' coverage predicate (must be 0), out[4] = rectangles called covered, out[5] = rectangles
// whose every pixel passes the predicate.
__global__ void k_selftest_edge_exclusion(long long n, unsigned long long seed, unsigned long long* out) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, excl = 0, empty = 0, bad_cov = 0, n_cov = 0, n_full = 0;
    for (long long i = i0; i < n; i += stride) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        auto next = [&]() { z += 0x9E3779B97F4A7C15ull; unsigned long long r = z; r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull; r = (r ^ (r >> 27)) * 0x94D049BB133111EBull; return r ^ (r >> 31); };
        auto unif = [&](float lo, float hi) { return lo + (hi - lo) * (float)((next() >> 40) * (1.0 / 16777216.0)); };
        const int kind = (int)(next() % 5);
        const float span = kind == 2 ? 1.0e4f : 356.0f, off = kind == 2 ? -5.0e3f : -50.0f;
        float x0 = off + unif(0.0f, span), y0 = off + unif(0.0f, span), x1 = off + unif(0.0f, span), y1 = off + unif(0.0f, span);
        float x2 = off + unif(0.0f, span), y2 = off + unif(0.0f, span);
        if (kind == 1) { const float t = unif(-0.5f, 1.5f), eps = unif(-1e-3f, 1e-3f); x2 = x0 + t * (x1 - x0) + eps; y2 = y0 + t * (y1 - y0) - eps; }   // sliver
        if (kind == 3) { x0 = floorf(x0) + 0.5f; y0 = floorf(y0) + 0.5f; x1 = floorf(x1) + 0.5f; y2 = floorf(y2) + 0.5f; }                          // on pixel centres
        if (kind == 4) { x1 = x0 + unif(-24.0f, 24.0f); y1 = y0 + unif(-24.0f, 24.0f); x2 = x0 + unif(-24.0f, 24.0f); y2 = y0 + unif(-24.0f, 24.0f); }   // heightfield-sized
        const bool cell = (next() & 1ull) != 0;
        const int w = cell ? 32 : 16, h = cell ? 8 : 16;
        const float X0 = (float)((int)(next() % (256 / w)) * w) + 0.5f, Y0 = (float)((int)(next() % (256 / h)) * h) + 0.5f;
        const float X1 = X0 + (float)(w - 1), Y1 = Y0 + (float)(h - 1);
        const bool ex = edges_exclude_rect(x0, y0, x1, y1, x2, y2, X0, X1, Y0, Y1);
        const bool cov = edges_cover_rect(x0, y0, x1, y1, x2, y2, X0, X1, Y0, Y1);
        const float bxl = fminf(x0, fminf(x1, x2)), bxh = fmaxf(x0, fmaxf(x1, x2)), byl = fminf(y0, fminf(y1, y2)), byh = fmaxf(y0, fmaxf(y1, y2));
        bool covered = false, all_hit = true;
        for (int py = 0; py < h; ++py) {
            const float fy = Y0 + (float)py;
            const float a0 = y2 - fy, a1 = y1 - fy, a2 = y0 - fy;
            for (int px = 0; px < w; ++px) {
                const float fx = X0 + (float)px;
                const float e0 = (x1 - fx) * a0 - (x2 - fx) * a1;
                const float e1 = (x2 - fx) * a2 - (x0 - fx) * a0;
                const float e2 = (x0 - fx) * a1 - (x1 - fx) * a2;
                const bool pos = (e0 >= 0.0f) & (e1 >= 0.0f) & (e2 >= 0.0f), neg = (e0 <= 0.0f) & (e1 <= 0.0f) & (e2 <= 0.0f);
                covered = covered | pos | neg;
                // the pixel loops' whole coverage predicate (bounding box, same-signed edge functions, non-degenerate)
                all_hit = all_hit & ((fx >= bxl) & (fx <= bxh) & (fy >= byl) & (fy <= byh) & (pos | neg) & (((e0 + e1) + e2) != 0.0f));
            }
        }
        bad += (ex && covered) ? 1 : 0;
        excl += ex ? 1 : 0;
        empty += covered ? 0 : 1;
        bad_cov += (cov && !all_hit) ? 1 : 0;
        n_cov += cov ? 1 : 0;
        n_full += all_hit ? 1 : 0;
    }
    if (bad) atomicAdd(out, bad);
    atomicAdd(out + 1, excl);
    atomicAdd(out + 2, empty);
    if (bad_cov) atomicAdd(out + 3, bad_cov);
    atomicAdd(out + 4, n_cov);
    atomicAdd(out + 5, n_full);
}
int selftest_edge_exclusion(long long n, unsigned long long seed, long long* out_host /*[6]*/) {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 48) != hipSuccess) return -1;
    (void)hipMemset(d, 0, 48);
    hipLaunchKernelGGL(k_selftest_edge_exclusion, dim3(2048), dim3(256), 0, 0, n, seed, d);
    const hipError_t e = hipMemcpy(out_host, d, 48, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? 0 : -1;
}


}  // namespace tg

extern "C" int tg_selftest_division(int64_t n, uint64_t seed, int64_t* mismatches) {
    if (!mismatches || n < 0) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return -2;
    long long m = 0;
    if (tg::selftest_division((long long)n, (unsigned long long)seed, &m) != 0) return -3;
    *mismatches = (int64_t)m;
    return 0;
}

extern "C" int tg_selftest_edge_exclusion(int64_t n, uint64_t seed, int64_t* out) {
    if (!out || n < 0) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return -2;
    long long m[6] = {-1, -1, -1, -1, -1, -1};
    if (tg::selftest_edge_exclusion((long long)n, (unsigned long long)seed, m) != 0) return -3;
    for (int k = 0; k < 6; ++k) out[k] = m[k];
    return 0;
}

extern "C" int tg_selftest_penetration_division(int64_t* mismatches) {
    if (!mismatches) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return -2;
    long long m = 0;
    if (tg::selftest_penetration_division(&m) != 0) return -3;
    *mismatches = (int64_t)m;
    return 0;
}
