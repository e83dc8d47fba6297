// tg_narrow_test.hip - tg_selftest_narrowphase: the wave-mapped GJK / EPA of tg_narrowphase.hpp on caller-supplied hull placements, one
// wavefront per case, for the parity test against oracle/narrowphase.c (tests/test_gpu_narrowphase.py).  Part of libtactile_gym_hip_test.so
// (test infrastructure, include/tactile_gym_hip_test.h), not of the product library.
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/tactile_gym_hip_test.h"
#include "tg_narrowphase.hpp"

namespace tg {
namespace {
__global__ __launch_bounds__(64) void k_narrow_test(int n_hull, const double* __restrict__ hulls, const double* __restrict__ half, double* __restrict__ out) {
    __shared__ double scratch[narrow::kScratchWords];
    const int cs = blockIdx.x, lane = threadIdx.x;
    narrow::Hull H;
    H.n = n_hull;
    const double* h = hulls + (size_t)cs * n_hull * 3;
#pragma unroll
    for (int k = 0; k < narrow::kSlots; ++k) {
        const int i = 64 * k + lane;
        const bool in = i < n_hull;
        H.x[k] = in ? h[3 * i] : 0.0; H.y[k] = in ? h[3 * i + 1] : 0.0; H.z[k] = in ? h[3 * i + 2] : 0.0;
    }
    const double e[3] = {half[0], half[1], half[2]};
    double sd = 0.0, n[3] = {0, 0, 0}, pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
    const bool ok = narrow::gjk_epa_hull_box(H, e, (narrow::lptr<double>)scratch, sd, n, pa, pb, lane);
    if (lane == 0) {
        double* o = out + (size_t)cs * 11;
        o[0] = ok ? 1.0 : 0.0; o[1] = sd;
        for (int c = 0; c < 3; ++c) { o[2 + c] = n[c]; o[5 + c] = pa[c]; o[8 + c] = pb[c]; }
    }
}
// the same against a second hull (tg_spin.hip's pair): hull A per case in B's frame, hull B shared
__global__ __launch_bounds__(64) void k_narrow_test_hulls(int n_hull, const double* __restrict__ hulls, int n_b, const double* __restrict__ hull_b, double* __restrict__ out) {
    __shared__ double scratch[narrow::kScratchWords];
    const int cs = blockIdx.x, lane = threadIdx.x;
    narrow::Hull H;
    H.n = n_hull;
    const double* h = hulls + (size_t)cs * n_hull * 3;
#pragma unroll
    for (int k = 0; k < narrow::kSlots; ++k) {
        const int i = 64 * k + lane;
        const bool in = i < n_hull;
        H.x[k] = in ? h[3 * i] : 0.0; H.y[k] = in ? h[3 * i + 1] : 0.0; H.z[k] = in ? h[3 * i + 2] : 0.0;
    }
    narrow::HullB B;
    B.n = n_b;
#pragma unroll
    for (int k = 0; k < narrow::kSlotsB; ++k) {
        const int i = 64 * k + lane;
        const bool in = i < n_b;
        B.x[k] = in ? hull_b[3 * i] : 0.0; B.y[k] = in ? hull_b[3 * i + 1] : 0.0; B.z[k] = in ? hull_b[3 * i + 2] : 0.0;
    }
    double sd = 0.0, n[3] = {0, 0, 0}, pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
    const bool ok = narrow::gjk_epa_hull_hull(H, B, (narrow::lptr<double>)scratch, sd, n, pa, pb, lane);
    if (lane == 0) {
        double* o = out + (size_t)cs * 11;
        o[0] = ok ? 1.0 : 0.0; o[1] = sd;
        for (int c = 0; c < 3; ++c) { o[2 + c] = n[c]; o[5 + c] = pa[c]; o[8 + c] = pb[c]; }
    }
}
}  // namespace
}  // namespace tg

extern "C" int tg_selftest_narrowphase(int32_t n_cases, int32_t n_hull, const double* hulls, const double* half, double* out) {
    if (!hulls || !half || !out || n_cases <= 0 || n_hull <= 0 || n_hull > 64 * tg::narrow::kSlots) return -1;
    double *dh = nullptr, *de = nullptr, *dout = nullptr;
    const size_t hb = (size_t)n_cases * n_hull * 3 * 8;
    int rc = -2;
    if (hipMalloc(&dh, hb) == hipSuccess && hipMalloc(&de, 24) == hipSuccess && hipMalloc(&dout, (size_t)n_cases * 11 * 8) == hipSuccess &&
        hipMemcpy(dh, hulls, hb, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(de, half, 24, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(tg::k_narrow_test, dim3(n_cases), dim3(64), 0, 0, n_hull, dh, de, dout);
        if (hipMemcpy(out, dout, (size_t)n_cases * 11 * 8, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    if (dh) (void)hipFree(dh);
    if (de) (void)hipFree(de);
    if (dout) (void)hipFree(dout);
    return rc;
}

extern "C" int tg_selftest_narrowphase_hulls(int32_t n_cases, int32_t n_hull, const double* hulls, int32_t n_b, const double* hull_b, double* out) {
    if (!hulls || !hull_b || !out || n_cases <= 0 || n_hull <= 0 || n_hull > 64 * tg::narrow::kSlots || n_b <= 0 || n_b > 64 * tg::narrow::kSlotsB) return -1;
    double *dh = nullptr, *db = nullptr, *dout = nullptr;
    const size_t hb = (size_t)n_cases * n_hull * 3 * 8, bb = (size_t)n_b * 3 * 8;
    int rc = -2;
    if (hipMalloc(&dh, hb) == hipSuccess && hipMalloc(&db, bb) == hipSuccess && hipMalloc(&dout, (size_t)n_cases * 11 * 8) == hipSuccess &&
        hipMemcpy(dh, hulls, hb, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(db, hull_b, bb, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(tg::k_narrow_test_hulls, dim3(n_cases), dim3(64), 0, 0, n_hull, dh, n_b, db, dout);
        if (hipMemcpy(out, dout, (size_t)n_cases * 11 * 8, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    if (dh) (void)hipFree(dh);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return rc;
}
