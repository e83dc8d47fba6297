// tg_kt.hpp - a kernel's duration by its own clock (profiling mode only: tg_profile_enable).
//
// HIP events around a launch carry 3 - 5 us of their own (an EMPTY event pair measures 4.7 us on an MI355X), which is a fifth of the
// headline's kernels: the per-kernel figures of bench.py did not fit inside the step they add up to.  With `base` non-null
// every wavefront's first lane stamps wall_clock64 (the constant 100 MHz counter) when it starts and when it leaves the kernel, into a slot
// of its own (no atomics: 4096 same-line atomics would cost more than the kernels measured); k_kt_reduce (tg_api.hip) then adds
// max(end) - min(start) over the slots - first wavefront's start to last wavefront's end, what rocprofv3 --kernel-trace reports per dispatch
// less the dispatch itself - to the accumulator of the launch's class.  Plain stores: a wavefront ends only when its memory operations are
// acknowledged (s_endpgm waits), and an atomic's round trip per wavefront showed as +25 us on a kernel whose wavefronts come in sixteen rounds
// (16 384 envs), where a store rides under the image stores issued just before it - and only the workgroups that can hold the extremes stamp at
// all: workgroups are dispatched in index order, so the first start is among the first kEdge of them and the last end (but for a straggler that
// outlives 8192 later wavefronts) among the last kEdge of a z-layer (see ~KtScope).  A scope of several instrumented launches moves the base
// on between them (reset_sequence), so that a later launch does not overwrite the first one's start stamps.  base == nullptr (every launch outside
// profiling mode, every graph): one uniform branch.
#pragma once
#include <hip/hip_runtime.h>

namespace tg {

#ifdef TG_NO_KT        // A/B build switch: what the instrumentation costs a kernel in registers (see the kernels that do without it)
struct KtScope { __device__ __forceinline__ explicit KtScope(unsigned long long*) {} };
#else
struct KtScope {
    unsigned long long* base;     // (uniform: stays in scalar registers; the slot address is formed again at the end rather than held in VGPRs)
    static constexpr size_t kEdge = 2048;   // workgroups at either end of the grid that stamp (8192 wavefronts: more than the chip holds at once)
    __device__ __forceinline__ static size_t workgroup() {
        return (size_t)blockIdx.x + (size_t)gridDim.x * ((size_t)blockIdx.y + (size_t)gridDim.y * (size_t)blockIdx.z);
    }
    __device__ __forceinline__ static unsigned long long* slot(unsigned long long* b, size_t wg) {
        return b + 2 * (wg * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6));
    }
    __device__ __forceinline__ explicit KtScope(unsigned long long* b) : base(b) {
        if (base != nullptr && (threadIdx.x & 63) == 0) {
            const size_t wg = workgroup();
            if (wg < kEdge) slot(base, wg)[0] = wall_clock64();
        }
    }
    __device__ __forceinline__ ~KtScope() {
        if (base != nullptr && (threadIdx.x & 63) == 0) {
            // the last kEdge workgroups of EVERY z-layer, not of the grid: the renders' second layer (the terminal images, blockIdx.z = 1) is
            // workgroups that look at their env's flag and leave in nearly every launch - the grid's last workgroups by index, and long gone when the
            // first layer's last wavefronts end (until this was found, late in round 6, the heightfield render read 44.7 us by this clock and 61 by
            // rocprofv3, and surface_follow's kernels did not add up to its step by 20 us; k_render_blocks 16.4 against 17.3)
            const size_t wg = workgroup(), layer = (size_t)gridDim.x * gridDim.y;
            const size_t in_layer = (size_t)blockIdx.x + (size_t)gridDim.x * (size_t)blockIdx.y;
            if (in_layer + kEdge >= layer) slot(base, wg)[1] = wall_clock64();
        }
    }
    KtScope(const KtScope&) = delete;
    KtScope& operator=(const KtScope&) = delete;
};
#endif

}  // namespace tg
