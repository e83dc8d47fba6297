// Latency microbenchmark (gfx950): cycles per dependent step of the cross-lane broadcast patterns a wave-per-env Gauss-Seidel can use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
__device__ __forceinline__ double bcast_sgpr(double v, int l) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l); u.i[1] = __builtin_amdgcn_readlane(u.i[1], l); return u.d;
}
template <int CTRL> __device__ __forceinline__ double bcast_dpp(double v) {   // row_newbcast: lane (CTRL & 15) of each 16-lane row to the row
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, false);
    u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, false);
    return u.d;
}
template <int MODE> __global__ void k(double* out, long long* cyc, int iters, double g0) {
    const int lane = threadIdx.x;
    double s = 1.0 + 1e-3 * lane, g = g0 + 1e-9 * lane, g2 = g0 * 0.5, lam = 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { s = __builtin_fma(-g, s, s); }                                  // plain dependent fma chain
            else if (MODE == 1) { double d = bcast_sgpr(s - lam, i); s = __builtin_fma(-g, d, s); }   // sub -> readlane x2 -> fma
            else if (MODE == 2) { double d = (i & 1) ? bcast_dpp<0x150 + 3>(s - lam) : bcast_dpp<0x150 + 5>(s - lam); s = __builtin_fma(-g, d, s); }
            else if (MODE == 3) { double c = __builtin_fmin(__builtin_fmax(s, -1e3), 1e3); double d = bcast_sgpr(c - lam, i); lam = lane == i ? c : lam; s = __builtin_fma(-g, d, s); }
            else if (MODE == 4) { double c = __builtin_fmin(__builtin_fmax(s, -1e3), 1e3); double d = (i & 1) ? bcast_dpp<0x150 + 3>(c - lam) : bcast_dpp<0x150 + 5>(c - lam); lam = lane == i ? c : lam; s = __builtin_fma(-g, d, s); }
            else if (MODE == 6) { double d = bcast_sgpr(s, i); s = __builtin_fma(-g, d, s); asm volatile("s_mov_b64 exec, %2\n\tv_add_f64 %0, %0, %1\n\ts_mov_b64 exec, -1" : "+v"(lam) : "s"(d), "n"(1u << 3)); }
            else if (MODE == 7) { double d = bcast_sgpr(s, i); s = __builtin_fma(-g, d, s); lam = lane == i ? lam + d : lam; }
            else if (MODE == 8) { double d = bcast_sgpr(s, i); s = __builtin_fma(-g, d, s); }
            else if (MODE == 9) { double c; asm("v_max_f64 %0, %1, -%2" : "=v"(c) : "v"(s), "v"(lam)); double d = bcast_sgpr(c, i); s = __builtin_fma(-g, d, s); asm volatile("s_mov_b64 exec, %2\n\tv_add_f64 %0, %0, %1\n\ts_mov_b64 exec, -1" : "+v"(lam) : "s"(d), "n"(1u << 3)); }
            else if (MODE == 10 || MODE == 11) {
                const double s1 = bcast_sgpr(s, 9), s2 = bcast_sgpr(s, 10), limit = 0.065 * bcast_sgpr(lam, 8);
                const double tot2 = __builtin_fma(s2, s2, s1 * s1);
                double mx; asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(tot2), "v"(1e-280));
                const double y = __builtin_amdgcn_rsq(mx);
                const double t = tot2 * y, lh = (0.5 * limit) * y, ly = limit * y;
                const double e = __builtin_fma(-t, y, 1.0);
                double f = __builtin_fma(lh, e, ly);
                f = tot2 > limit * limit ? f : 1.0;
                const double dl = __builtin_fma(s, f, -lam);
                const double d1 = bcast_sgpr(dl, 9), d2 = bcast_sgpr(dl, 10);
                s = __builtin_fma(-g, d1, s); s = __builtin_fma(-g2, d2, s);
                if (MODE == 10) asm volatile("s_mov_b64 exec, %2\n\tv_add_f64 %0, %0, %1\n\ts_mov_b64 exec, -1" : "+v"(lam) : "v"(dl), "n"(0x600));
            }
            else if (MODE >= 20 && MODE <= 23) {   // the kernel's lane-local friction step: 20 as shipped (v_min), 21 clamp modifier, 22 f32 seed + clamp, 23 f32 seed two Newton
                union { double d; int i[2]; } u; u.d = s;
                u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0xD8, 0xF, 0xF, true); u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0xD8, 0xF, 0xF, true);
                const double xp = u.d, limit = g2;
                const double tot2 = __builtin_fma(xp, xp, s * s);
                double f;
                if (MODE == 20 || MODE == 21) {
                    const double y = __builtin_amdgcn_rsq(tot2);
                    const double t = tot2 * y, ly = limit * y, lh = 0.5 * ly, e = __builtin_fma(-t, y, 1.0);
                    if (MODE == 20) { const double f0 = __builtin_fma(lh, e, ly); asm("v_min_f64 %0, %1, %2" : "=v"(f) : "v"(f0), "v"(1.0)); }
                    else asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(f) : "v"(lh), "v"(e), "v"(ly));
                } else {
                    const double y = (double)__builtin_amdgcn_rsqf((float)tot2);
                    const double t = tot2 * y, ly = limit * y, lh = 0.5 * ly, e = __builtin_fma(-t, y, 1.0);
                    asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(f) : "v"(lh), "v"(e), "v"(ly));
                }
                const double dl = __builtin_fma(s, f, -lam);
                lam = __builtin_fma(g0, dl, lam);
                const double d1 = bcast_sgpr(dl, 9), d2 = bcast_sgpr(dl, 10);
                s = __builtin_fma(-g, d1, s); s = __builtin_fma(-g2, d2, s);
            }
            else if (MODE == 5) { float sf = (float)s; float d = __shfl(sf, i); s = __builtin_fma(-g, (double)d, s); }   // ds_bpermute path
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = s + lam;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks) {
    double* out; long long* cyc; hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1e-7);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1e-7); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    printf("%-44s blocks %5d: %.2f us, %.1f ns per step, s_memtime ticks per step %.2f\n", name, blocks, ms * 1e3, ms * 1e6 / (iters * 8.0), (double)h[0] / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}
__global__ void k_rsq(const double* x, double* y0, double* y1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double v = x[i], y = __builtin_amdgcn_rsq(v);
    y0[i] = y;
    double t = v * y, h = 0.5 * y, e = __builtin_fma(-t, y, 1.0);
    y1[i] = __builtin_fma(h, e, y);
}
void rsq_accuracy() {
    const int n = 1 << 20; std::vector<double> x(n), a(n), b(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; double u = (st >> 11) * (1.0 / 9007199254740992.0); x[i] = ldexp(1.0 + u, (int)(st % 81) - 60); }
    double *dx, *d0, *d1; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rsq, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0;
    for (int i = 0; i < n; ++i) { long double ex = 1.0L / sqrtl((long double)x[i]); e0 = fmax(e0, fabs((double)((a[i] - ex) / ex))); e1 = fmax(e1, fabs((double)((b[i] - ex) / ex))); }
    printf("v_rsq_f64: max rel err of the seed %.3e, after one Newton step %.3e (ulp = 1.1e-16)\n", e0, e1);
}
int main() {
    rsq_accuracy();
    for (int blocks : {1024}) {
        run<0>("fma chain", blocks);
        run<1>("sub + readlane x2 + fma(sgpr)", blocks);
        run<2>("sub + dpp row_newbcast x2 + fma", blocks);
        run<3>("clamp + sub + readlane x2 + sel + fma", blocks);
        run<4>("clamp + sub + dpp x2 + sel + fma", blocks);
        run<5>("cvt + ds_bpermute + cvt + fma", blocks);
        run<8>("readlane x2 + fma", blocks);
        run<6>("readlane x2 + fma + exec-masked add (asm)", blocks);
        run<7>("readlane x2 + fma + cndmask add", blocks);
        run<9>("vmax_neg + readlane x2 + fma + exec add", blocks);
        run<11>("friction step, no lambda update", blocks);
        run<10>("friction step + exec-masked add", blocks);
        run<20>("lane-local friction step as shipped (v_min)", blocks);
        run<21>("lane-local friction step, clamp modifier", blocks);
        run<22>("lane-local friction step, f32 rsq seed + clamp", blocks);
    }
    return 0;
}
