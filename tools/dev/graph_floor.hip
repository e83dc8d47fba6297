// Dispatch-floor microbenchmark (gfx950): time per iteration of a replayed graph of K dependent kernels that do (almost) nothing, against
// the same kernels launched on the stream, so that the step's "gaps" (ms_per_step - sum of kernel durations) can be priced.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/graph_floor.hip -o /tmp/graph_floor && /tmp/graph_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tiny(unsigned* p, int grid_work) { if (threadIdx.x == 0 && grid_work) p[blockIdx.x] += 1; }
__global__ void k_spin(unsigned* p, long long cycles) {   // one wavefront per workgroup busy for `cycles`, then one store
    const long long t0 = wall_clock64();              // 100 MHz
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) p[blockIdx.x] += 1;
}
int main() {
    unsigned* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t s, cap; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&cap));
    const int iters = 3000;
    auto run = [&](const char* name, auto&& body) {
        for (int i = 0; i < 200; ++i) body();
        (void)hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) body();
        (void)hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("%-58s %7.2f us per iteration\n", name, us);
    };
    for (int K = 1; K <= 3; ++K) {
        for (int mode = 0; mode < 3; ++mode) {   // 0 tiny (16 workgroups), 1 spin 15 us on 16 workgroups, 2 spin 15 us on 2048 workgroups
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < K; ++k) {
                if (mode == 0) hipLaunchKernelGGL(k_tiny, dim3(16), dim3(64), 0, cap, d, 1);
                else hipLaunchKernelGGL(k_spin, dim3(mode == 1 ? 16 : 2048), dim3(mode == 1 ? 64 : 256), 0, cap, d, 15LL * 100);   // s_memtime: 100 MHz
            }
            CK(hipStreamEndCapture(cap, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            char name[128];
            snprintf(name, sizeof name, "graph of %d %s kernel(s)", K, mode == 0 ? "tiny" : mode == 1 ? "15-us 16-wg" : "15-us 2048-wg");
            run(name, [&]() { (void)hipGraphLaunch(ge, s); });
            snprintf(name, sizeof name, "stream launches, %d %s kernel(s)", K, mode == 0 ? "tiny" : mode == 1 ? "15-us 16-wg" : "15-us 2048-wg");
            run(name, [&]() { for (int k = 0; k < K; ++k) { if (mode == 0) hipLaunchKernelGGL(k_tiny, dim3(16), dim3(64), 0, s, d, 1);
                                                          else hipLaunchKernelGGL(k_spin, dim3(mode == 1 ? 16 : 2048), dim3(mode == 1 ? 64 : 256), 0, s, d, 15LL * 100); } });
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    }
    return 0;
}
