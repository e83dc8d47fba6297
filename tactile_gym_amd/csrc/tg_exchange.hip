// Per-step exchange of observations to rank 0 (replaces SubprocVecEnv's pickled pipes, sb3_helpers/rl_utils.py:17-30).
//
//  * Tile-sparse tactile payload.  A tactile image is zero away from the contact patch and a constant paste on the sensor's border ring
//    (tactile_sensor.py:261-294): cut into 16 x 16 tiles, 3-9 % of the tiles of an edge_follow image differ from that constant
//    template.  tg_pack_tiles writes only those tiles (lossless), tg_unpack_tiles restores the images.
//  * Direct stores over xGMI.  Rank 0 allocates the receive slots (tg_ipc_alloc) and hands every peer the IPC handle; a peer's pack
//    kernel stores straight into its slot of rank 0's HBM over its own xGMI link - no staging copy, no collective, and the message
//    length (count of live tiles) never has to be known on a host.  Ordering is carried by monotone 32-bit flags in the same allocation:
//    a one-lane kernel releases a flag at system scope after the producing kernel has finished (its stores are written back at the end
//    of that kernel), the other side's one-wave kernel spins on it with acquire loads - bounded by a wall-clock timeout, so that a
//    missing partner becomes an error word instead of a hung queue.
//
// Everything here is context free: raw device pointers and a HIP stream.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/tactile_gym_hip.h"
#include "tg_exchange.h"

namespace tg {
namespace {

constexpr uint32_t kTileMagic = 0x54475431u;   // "TGT1"
constexpr int kRecWords = 17;                  // a record: one 16-byte header (tile id) + 16 rows of 16 pixels

#define TGX_HIP(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return report_error(-2, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

// One wavefront per 64 tiles of one image; lane = tile.  A lane reads its tile (16 rows x 16 B: for one row the lanes of a tile row read
// consecutive 16-byte pieces, whole cache lines are used) and the same piece of the template; live = any pixel differs.  The wave takes
// `count` consecutive record slots with ONE atomic on a counter in local memory, compacts its live tiles through LDS and streams the
// records out as consecutive 16-byte stores (so a remote destination sees full-width writes).  The last wave to finish publishes the
// header and clears the counters for the next launch.
__global__ __launch_bounds__(64) void k_pack_tiles(const uint8_t* __restrict__ obs, const uint8_t* __restrict__ tmpl, int n_img, int H, int W, int T,
                                                   int TW, int groups, uint8_t* __restrict__ dst, uint32_t* __restrict__ counters,
                                                   const uint8_t* __restrict__ tail_src, size_t tail_bytes, size_t tail_offset) {
    __shared__ uint4 rec[64 * kRecWords];
    const int img = blockIdx.x / groups, g = blockIdx.x - img * groups;
    const int lane = threadIdx.x;
    const int tile = g * 64 + lane;
    uint4 rows[16];
    bool live = false;
    if (tile < T) {
        const int tr = tile / TW, tc = tile - tr * TW;
        const size_t off = (size_t)(tr * 16) * W + (size_t)tc * 16;
        const uint8_t* __restrict__ o = obs + (size_t)img * H * W + off;
        const uint8_t* __restrict__ t = tmpl + off;
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            rows[r] = *reinterpret_cast<const uint4*>(o + (size_t)r * W);
            const uint4 tv = *reinterpret_cast<const uint4*>(t + (size_t)r * W);
            acc |= (rows[r].x ^ tv.x) | (rows[r].y ^ tv.y) | (rows[r].z ^ tv.z) | (rows[r].w ^ tv.w);
        }
        live = acc != 0;
    }
    const unsigned long long mask = __ballot(live);
    const int cnt = __popcll(mask);
    uint32_t base = 0;
    if (lane == 0 && cnt) base = atomicAdd(&counters[0], (uint32_t)cnt);
    base = __builtin_amdgcn_readfirstlane(base);
    if (live) {
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        rec[rank * kRecWords] = make_uint4((uint32_t)(img * T + tile), 0u, 0u, 0u);
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[rank * kRecWords + 1 + r] = rows[r];
    }
    __syncthreads();
    uint4* __restrict__ out = reinterpret_cast<uint4*>(dst + 16) + (size_t)base * kRecWords;
    for (int i = lane; i < cnt * kRecWords; i += 64) out[i] = rec[i];
    // No fence here (round 5).  The header's count comes from atomics (the ticket is taken after this wave's slot atomic has RETURNED: `base`
    // is consumed above), and every reader of records / header runs behind this kernel's end - a later launch or, on a peer, a flag raised by
    // one.  An agent-scope fence per workgroup is an L2 write-back + invalidate on a multi-XCD part: with 1024 workgroups it was ~20 us of this
    // kernel (profiles/r5_exp_fused_step.txt measured the same fence in k_step_render).
    __shared__ uint32_t last;
    if (lane == 0) {
        const uint32_t ticket = atomicAdd(&counters[1], 1u);
        last = ticket == gridDim.x - 1 ? 1u : 0u;
        if (last) {                                       // every other wave has taken its slots and written its records
            const uint32_t total = atomicExch(&counters[0], 0u);
            counters[1] = 0u;
            *reinterpret_cast<uint4*>(dst) = make_uint4(total, (uint32_t)n_img, (uint32_t)T, kTileMagic);
        }
    }
    __syncthreads();
    if (last && tail_src != nullptr) {                    // the small block that rides behind the images (reward | done | feature): same message, same launch
        const size_t n16 = tail_bytes / 16;
        for (size_t i = lane; i < n16; i += 64) reinterpret_cast<uint4*>(dst + tail_offset)[i] = reinterpret_cast<const uint4*>(tail_src)[i];
        if ((size_t)lane < tail_bytes - 16 * n16) dst[tail_offset + 16 * n16 + lane] = tail_src[16 * n16 + lane];
    }
}

__global__ __launch_bounds__(256) void k_fill_template(const uint4* __restrict__ tmpl, int hw16, size_t total16, uint4* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total16; i += (size_t)gridDim.x * blockDim.x) dst[i] = tmpl[i % hw16];
}

// 16 lanes per record (one row each); records beyond the header's count do nothing.
__global__ __launch_bounds__(256) void k_scatter_tiles(const uint8_t* __restrict__ src, int n_img, int H, int W, int T, int TW, uint8_t* __restrict__ dst) {
    const uint4 hdr = *reinterpret_cast<const uint4*>(src);
    if (hdr.w != kTileMagic || (int)hdr.z != T) return;                     // not a tile message for this image geometry: leave the template
    const uint32_t cap = (uint32_t)n_img * (uint32_t)T;
    const uint32_t count = hdr.x < cap ? hdr.x : cap;
    const uint32_t slot = blockIdx.x * 16u + (threadIdx.x >> 4);
    const int row = threadIdx.x & 15;
    if (slot >= count) return;
    const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(src + 16) + (size_t)slot * kRecWords;
    const uint32_t id = rec[0].x;
    const int img = (int)(id / (uint32_t)T), tile = (int)(id - (uint32_t)img * (uint32_t)T);
    if (img >= n_img) return;
    const int tr = tile / TW, tc = tile - tr * TW;
    *reinterpret_cast<uint4*>(dst + (size_t)img * H * W + (size_t)(tr * 16 + row) * W + (size_t)tc * 16) = rec[1 + row];
}

// 16-byte pieces (grid stride), then the last < 16 bytes one by one: a device copy whose destination may be another GPU's memory.
// The same two steps for the messages of several ranks in one launch each (rank 0 unpacks world - 1 messages per step; one launch per
// message costs more than the messages' bytes): blockIdx.y = rank, messages `stride` bytes apart, rank `skip` left alone (rank 0's own
// block is a plain copy of its images).
__global__ __launch_bounds__(256) void k_fill_template_multi(const uint4* __restrict__ tmpl, int hw16, size_t per_rank16, int skip, uint4* __restrict__ dst) {
    const int r = blockIdx.y;
    if (r == skip) return;
    uint4* __restrict__ d = dst + (size_t)r * per_rank16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_rank16; i += (size_t)gridDim.x * blockDim.x) d[i] = tmpl[i % hw16];
}
__global__ __launch_bounds__(256) void k_scatter_tiles_multi(const uint8_t* __restrict__ src, size_t stride, int skip, int n_img, int H, int W, int T, int TW,
                                                            uint8_t* __restrict__ dst) {
    const int r = blockIdx.y;
    if (r == skip) return;
    const uint8_t* __restrict__ s = src + (size_t)r * stride;
    uint8_t* __restrict__ d = dst + (size_t)r * n_img * H * W;
    const uint4 hdr = *reinterpret_cast<const uint4*>(s);
    if (hdr.w != kTileMagic || (int)hdr.z != T) return;
    const uint32_t cap = (uint32_t)n_img * (uint32_t)T;
    const uint32_t count = hdr.x < cap ? hdr.x : cap;
    const int row = threadIdx.x & 15;
    for (uint32_t slot = blockIdx.x * 16u + (threadIdx.x >> 4); slot < count; slot += gridDim.x * 16u) {
        const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(s + 16) + (size_t)slot * kRecWords;
        const uint32_t id = rec[0].x;
        const int img = (int)(id / (uint32_t)T), tile = (int)(id - (uint32_t)img * (uint32_t)T);
        if (img >= n_img) continue;
        const int tr = tile / TW, tc = tile - tr * TW;
        *reinterpret_cast<uint4*>(d + (size_t)img * H * W + (size_t)(tr * 16 + row) * W + (size_t)tc * 16) = rec[1 + row];
    }
}

// With a list of the tiles the previous message of this slot had live (`prev`: per rank {count, ids...}, stride `prev_stride` words), the
// template is restored on exactly those tiles instead of on the whole batch (a full fill is 17 MB per rank and step; the live tiles are a
// few % of that), and the scatter records the new list.  The batch buffer must hold the template to begin with.
__global__ __launch_bounds__(256) void k_restore_tiles_multi(const uint8_t* __restrict__ tmpl, const uint32_t* __restrict__ prev, size_t prev_stride, int skip,
                                                            int n_img, int H, int W, int T, int TW, uint8_t* __restrict__ dst) {
    const int r = blockIdx.y;
    if (r == skip) return;
    const uint32_t* __restrict__ pl = prev + (size_t)r * prev_stride;
    const uint32_t cap = (uint32_t)n_img * (uint32_t)T;
    const uint32_t count = pl[0] < cap ? pl[0] : cap;
    uint8_t* __restrict__ d = dst + (size_t)r * n_img * H * W;
    const int row = threadIdx.x & 15;
    for (uint32_t slot = blockIdx.x * 16u + (threadIdx.x >> 4); slot < count; slot += gridDim.x * 16u) {
        const uint32_t id = pl[1 + slot];
        const int img = (int)(id / (uint32_t)T), tile = (int)(id - (uint32_t)img * (uint32_t)T);
        if (img >= n_img) continue;
        const int tr = tile / TW, tc = tile - tr * TW;
        const size_t off = (size_t)(tr * 16 + row) * W + (size_t)tc * 16;
        *reinterpret_cast<uint4*>(d + (size_t)img * H * W + off) = *reinterpret_cast<const uint4*>(tmpl + off);
    }
}
__global__ __launch_bounds__(256) void k_scatter_tiles_multi_keep(const uint8_t* __restrict__ src, size_t stride, int skip, int n_img, int H, int W, int T, int TW,
                                                                 uint8_t* __restrict__ dst, uint32_t* __restrict__ prev, size_t prev_stride) {
    const int r = blockIdx.y;
    if (r == skip) return;
    const uint8_t* __restrict__ s = src + (size_t)r * stride;
    uint8_t* __restrict__ d = dst + (size_t)r * n_img * H * W;
    uint32_t* __restrict__ pl = prev + (size_t)r * prev_stride;
    const uint4 hdr = *reinterpret_cast<const uint4*>(s);
    const bool ok = hdr.w == kTileMagic && (int)hdr.z == T;
    const uint32_t cap = (uint32_t)n_img * (uint32_t)T;
    const uint32_t count = ok ? (hdr.x < cap ? hdr.x : cap) : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) pl[0] = count;
    const int row = threadIdx.x & 15;
    for (uint32_t slot = blockIdx.x * 16u + (threadIdx.x >> 4); slot < count; slot += gridDim.x * 16u) {
        const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(s + 16) + (size_t)slot * kRecWords;
        const uint32_t id = rec[0].x;
        if (row == 0) pl[1 + slot] = id;
        const int img = (int)(id / (uint32_t)T), tile = (int)(id - (uint32_t)img * (uint32_t)T);
        if (img >= n_img) continue;
        const int tr = tile / TW, tc = tile - tr * TW;
        *reinterpret_cast<uint4*>(d + (size_t)img * H * W + (size_t)(tr * 16 + row) * W + (size_t)tc * 16) = rec[1 + row];
    }
}

// two ranges in one launch (rank 0 copies its own images and the block behind them per step)
// flags (nullable): the first n lanes of workgroup 0 also raise flags[i * stride] to `value` (system-scope release) - rank 0's per-step
// "slot consumed" signal rides in the launch that copies its own images instead of a launch of its own
__global__ __launch_bounds__(256) void k_copy_bytes2(const uint8_t* __restrict__ s1, size_t b1, uint8_t* __restrict__ d1, const uint8_t* __restrict__ s2, size_t b2,
                                                    uint8_t* __restrict__ d2, uint32_t* flags, int n_flags, int stride, uint32_t value) {
    if (flags != nullptr && blockIdx.x == 0 && (int)threadIdx.x < n_flags)
        __hip_atomic_store(flags + (size_t)threadIdx.x * stride, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const size_t n1 = b1 / 16, n2 = b2 / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n1) reinterpret_cast<uint4*>(d1)[i] = reinterpret_cast<const uint4*>(s1)[i];
        else reinterpret_cast<uint4*>(d2)[i - n1] = reinterpret_cast<const uint4*>(s2)[i - n1];
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < b1 - 16 * n1) d1[16 * n1 + threadIdx.x] = s1[16 * n1 + threadIdx.x];
        if (threadIdx.x < b2 - 16 * n2) d2[16 * n2 + threadIdx.x] = s2[16 * n2 + threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void k_copy_bytes(const uint8_t* __restrict__ src, size_t bytes, uint8_t* __restrict__ dst) {
    const size_t n16 = bytes / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    if (blockIdx.x == 0 && threadIdx.x < bytes - 16 * n16) dst[16 * n16 + threadIdx.x] = src[16 * n16 + threadIdx.x];
}

__global__ __launch_bounds__(64) void k_flag_set(uint32_t* flags, int n, int stride, uint32_t value) {
    const int i = threadIdx.x;
    if (i < n) __hip_atomic_store(flags + (size_t)i * stride, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Lane i waits until flags[i * stride] has reached `value` (monotone counters, compared modulo 2^32).  100 MHz wall clock.
__global__ __launch_bounds__(64) void k_flag_wait(const uint32_t* flags, int n, int stride, uint32_t value, uint32_t* err, long long timeout_ticks) {
    const int i = threadIdx.x;
    if (i >= n) return;
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // a wait of this rank has timed out already: the
    const uint32_t* p = flags + (size_t)i * stride;                                                  // exchange is broken, do not sit out every later one
    const long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
        if (wall_clock64() - t0 > timeout_ticks) {
            if (err) atomicOr(err, 1u << (i & 31));
            break;
        }
        __builtin_amdgcn_s_sleep(16);
    }
}

bool tile_geometry(int h, int w) { return h > 0 && w > 0 && h % 16 == 0 && w % 16 == 0; }

}  // namespace
}  // namespace tg

using tg::report_error;

int tg_tiles_capacity(int32_t n_images, int32_t h, int32_t w, int64_t* bytes) {
    if (!bytes || n_images <= 0 || !tg::tile_geometry(h, w)) return report_error(-1, "tg_tiles_capacity: image sides must be positive multiples of 16");
    *bytes = 16 + (int64_t)n_images * (h / 16) * (w / 16) * (16 * tg::kRecWords);
    return 0;
}

int tg_pack_tiles(void* stream, const void* obs_dev, const void* template_dev, int32_t n_images, int32_t h, int32_t w, void* dst_dev, void* counters_dev,
                  const void* tail_src_dev, int64_t tail_bytes, int64_t tail_offset) {
    if (!obs_dev || !template_dev || !dst_dev || !counters_dev || n_images <= 0 || !tg::tile_geometry(h, w))
        return report_error(-1, "tg_pack_tiles: bad argument (image sides must be multiples of 16)");
    if (tail_src_dev && (tail_bytes < 0 || tail_bytes > (1 << 20) || tail_offset < 0 || (tail_offset & 15) || ((uintptr_t)tail_src_dev & 15)))
        return report_error(-1, "tg_pack_tiles: the tail must be at most 1 MiB, 16-byte aligned at both ends");
    const int TW = w / 16, T = TW * (h / 16), groups = (T + 63) / 64;
    if ((int64_t)n_images * groups > 0x7fffffffLL || (int64_t)n_images * T > 0x7fffffffLL) return report_error(-1, "tg_pack_tiles: too many tiles for one launch");
    hipLaunchKernelGGL(tg::k_pack_tiles, dim3((unsigned)(n_images * groups)), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)obs_dev,
                       (const uint8_t*)template_dev, n_images, h, w, T, TW, groups, (uint8_t*)dst_dev, (uint32_t*)counters_dev,
                       (const uint8_t*)(tail_bytes > 0 ? tail_src_dev : nullptr), (size_t)(tail_bytes > 0 ? tail_bytes : 0), (size_t)tail_offset);
    TGX_HIP(hipGetLastError());
    return 0;
}

int tg_unpack_tiles(void* stream, const void* src_dev, const void* template_dev, int32_t n_images, int32_t h, int32_t w, void* dst_dev) {
    if (!src_dev || !template_dev || !dst_dev || n_images <= 0 || !tg::tile_geometry(h, w))
        return report_error(-1, "tg_unpack_tiles: bad argument (image sides must be multiples of 16)");
    const int TW = w / 16, T = TW * (h / 16);
    const int hw16 = h * w / 16;
    const size_t total16 = (size_t)n_images * hw16;
    const unsigned fill_blocks = (unsigned)((total16 + 255) / 256 < 8192 ? (total16 + 255) / 256 : 8192);
    hipLaunchKernelGGL(tg::k_fill_template, dim3(fill_blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)template_dev, hw16, total16, (uint4*)dst_dev);
    const int64_t cap = (int64_t)n_images * T;
    if ((cap + 15) / 16 > 0x7fffffffLL) return report_error(-1, "tg_unpack_tiles: too many tiles for one launch");
    hipLaunchKernelGGL(tg::k_scatter_tiles, dim3((unsigned)((cap + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src_dev, n_images, h, w,
                       T, TW, (uint8_t*)dst_dev);
    TGX_HIP(hipGetLastError());
    return 0;
}

int tg_unpack_tiles_multi(void* stream, const void* src_dev, int64_t src_stride, int32_t n_ranks, int32_t skip_rank, const void* template_dev,
                          int32_t n_images, int32_t h, int32_t w, void* dst_dev, void* prev_ids_dev) {
    if (!src_dev || !template_dev || !dst_dev || n_images <= 0 || n_ranks <= 0 || n_ranks > 65535 || src_stride < 16 || (src_stride & 15) || !tg::tile_geometry(h, w))
        return report_error(-1, "tg_unpack_tiles_multi: bad argument (stride a multiple of 16, image sides multiples of 16)");
    const int TW = w / 16, T = TW * (h / 16);
    const int hw16 = h * w / 16;
    const size_t per_rank16 = (size_t)n_images * hw16;
    if (prev_ids_dev != nullptr) {          // restore only what the slot's previous message had live, then scatter and remember the new list
        const int64_t cap_ = (int64_t)n_images * T;
        const unsigned blocks = (unsigned)((cap_ + 15) / 16 < 1024 ? (cap_ + 15) / 16 : 1024);
        hipLaunchKernelGGL(tg::k_restore_tiles_multi, dim3(blocks, (unsigned)n_ranks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)template_dev,
                           (const uint32_t*)prev_ids_dev, (size_t)(cap_ + 1), skip_rank, n_images, h, w, T, TW, (uint8_t*)dst_dev);
        hipLaunchKernelGGL(tg::k_scatter_tiles_multi_keep, dim3(blocks, (unsigned)n_ranks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src_dev,
                           (size_t)src_stride, skip_rank, n_images, h, w, T, TW, (uint8_t*)dst_dev, (uint32_t*)prev_ids_dev, (size_t)(cap_ + 1));
        TGX_HIP(hipGetLastError());
        return 0;
    }
    const unsigned fill_blocks = (unsigned)((per_rank16 + 255) / 256 < 2048 ? (per_rank16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(tg::k_fill_template_multi, dim3(fill_blocks, (unsigned)n_ranks), dim3(256), 0, (hipStream_t)stream, (const uint4*)template_dev, hw16,
                       per_rank16, skip_rank, (uint4*)dst_dev);
    const int64_t cap = (int64_t)n_images * T;
    const unsigned sc_blocks = (unsigned)((cap + 15) / 16 < 4096 ? (cap + 15) / 16 : 4096);     // grid stride beyond that: most messages hold a few % of the tiles
    hipLaunchKernelGGL(tg::k_scatter_tiles_multi, dim3(sc_blocks, (unsigned)n_ranks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src_dev,
                       (size_t)src_stride, skip_rank, n_images, h, w, T, TW, (uint8_t*)dst_dev);
    TGX_HIP(hipGetLastError());
    return 0;
}

static int g_ipc_uncached = 0;
int tg_ipc_alloc_was_uncached(void) { return g_ipc_uncached; }

int tg_ipc_alloc(int64_t bytes, void** dev_ptr, uint8_t* handle) {
    if (bytes <= 0 || !dev_ptr || !handle) return report_error(-1, "tg_ipc_alloc: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI hands IPC handles out as 64 bytes");
    void* p = nullptr;
    // uncached device memory: remote stores land in HBM and no stale line of it can sit in a local L2 (what RCCL allocates for its own
    // peer-written buffers); plain hipMalloc if the runtime refuses the flag
    g_ipc_uncached = 1;
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        g_ipc_uncached = 0;              // reported (tg_ipc_alloc_was_uncached): the caller decides whether cached receive slots are acceptable
        TGX_HIP(hipMalloc(&p, (size_t)bytes));
    }
    if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return report_error(-2, "tg_ipc_alloc: clearing the allocation failed");
    }
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return report_error(-2, (std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e) + " (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set)").c_str());
    }
    memcpy(handle, &h, 64);
    *dev_ptr = p;
    return 0;
}

int tg_ipc_free(void* dev_ptr) {
    if (dev_ptr) TGX_HIP(hipFree(dev_ptr));
    return 0;
}

int tg_ipc_open(const uint8_t* handle, void** dev_ptr) {
    if (!handle || !dev_ptr) return report_error(-1, "tg_ipc_open: NULL argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    void* p = nullptr;
    TGX_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *dev_ptr = p;
    return 0;
}

int tg_ipc_close(void* dev_ptr) {
    if (dev_ptr) TGX_HIP(hipIpcCloseMemHandle(dev_ptr));
    return 0;
}

int tg_copy_bytes(void* stream, void* dst_dev, const void* src_dev, int64_t bytes) {
    if (!dst_dev || !src_dev || bytes < 0) return report_error(-1, "tg_copy_bytes: bad argument");
    if (((uintptr_t)dst_dev | (uintptr_t)src_dev) & 15) return report_error(-1, "tg_copy_bytes: pointers must be 16-byte aligned");
    if (bytes == 0) return 0;
    const size_t n16 = (size_t)bytes / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096);
    hipLaunchKernelGGL(tg::k_copy_bytes, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src_dev, (size_t)bytes, (uint8_t*)dst_dev);
    TGX_HIP(hipGetLastError());
    return 0;
}

int tg_copy_bytes2(void* stream, void* dst1_dev, const void* src1_dev, int64_t bytes1, void* dst2_dev, const void* src2_dev, int64_t bytes2) {
    return tg_copy_bytes2_flag(stream, dst1_dev, src1_dev, bytes1, dst2_dev, src2_dev, bytes2, nullptr, 0, 1, 0u);
}

int tg_copy_bytes2_flag(void* stream, void* dst1_dev, const void* src1_dev, int64_t bytes1, void* dst2_dev, const void* src2_dev, int64_t bytes2,
                        void* flags_dev, int32_t n_flags, int32_t stride_words, uint32_t value) {
    if (!dst1_dev || !src1_dev || !dst2_dev || !src2_dev || bytes1 < 0 || bytes2 < 0) return report_error(-1, "tg_copy_bytes2: bad argument");
    if (flags_dev && (n_flags <= 0 || n_flags > 64 || stride_words <= 0)) return report_error(-1, "tg_copy_bytes2_flag: bad flag argument (1 <= n <= 64)");
    if (((uintptr_t)dst1_dev | (uintptr_t)src1_dev | (uintptr_t)dst2_dev | (uintptr_t)src2_dev) & 15) return report_error(-1, "tg_copy_bytes2: pointers must be 16-byte aligned");
    const size_t n16 = (size_t)(bytes1 + bytes2) / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096);
    hipLaunchKernelGGL(tg::k_copy_bytes2, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src1_dev, (size_t)bytes1, (uint8_t*)dst1_dev,
                       (const uint8_t*)src2_dev, (size_t)bytes2, (uint8_t*)dst2_dev, (uint32_t*)flags_dev, (int)n_flags, (int)stride_words, value);
    TGX_HIP(hipGetLastError());
    return 0;
}

int tg_flag_set(void* stream, void* flags_dev, int32_t n, int32_t stride_words, uint32_t value) {
    if (!flags_dev || n <= 0 || n > 64 || stride_words <= 0) return report_error(-1, "tg_flag_set: bad argument (1 <= n <= 64)");
    hipLaunchKernelGGL(tg::k_flag_set, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t*)flags_dev, n, stride_words, value);
    TGX_HIP(hipGetLastError());
    return 0;
}

int tg_flag_wait(void* stream, const void* flags_dev, int32_t n, int32_t stride_words, uint32_t value, void* err_dev, int32_t timeout_ms) {
    if (!flags_dev || n <= 0 || n > 64 || stride_words <= 0 || timeout_ms <= 0) return report_error(-1, "tg_flag_wait: bad argument (1 <= n <= 64, timeout > 0)");
    hipLaunchKernelGGL(tg::k_flag_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)flags_dev, n, stride_words, value, (uint32_t*)err_dev,
                       (long long)timeout_ms * 100000LL);
    TGX_HIP(hipGetLastError());
    return 0;
}
