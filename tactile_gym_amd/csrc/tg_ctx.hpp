// tg_ctx.hpp - what the translation units of the C ABI share: the context behind a tg_ctx*, error reporting, the device-selection / dispatch macros,
// the host -> device translation of a robot description.  tg_api.hip (configuration, creation, the step / reset launch sequences and their graphs,
// the reset bank), tg_api_state.hip (state read-back, inspection, profiling, the broadphase guard's entry points) and tg_api_ops.hip (the
// context-free function-level entry points) include it; round 6 split them out of one 2 400-line file.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/tactile_gym_hip.h"
#include "tg_kernels.hpp"
#include "tg_contact_wave.h"
#include "tg_spin.h"
#include "tg_fused.h"
#include "tg_scene.h"
#include "tg_noise.h"
#include "tg_raster.h"
#include "tg_exchange.h"
#include "tg_broadphase.h"


namespace tg {

inline thread_local std::string g_err;                       // tg_last_error(): one per thread, shared by every translation unit
static inline int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define TG_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

// Is every connected surface of the mesh closed and consistently wound with outward normals?  Vertices are matched by coordinates (OBJ
// files repeat them per face); closed + consistent = every directed edge a->b is met exactly once by b->a; outward = positive signed
// volume per connected component.  (What licenses the raster's back-face cull, tg_raster.hip:back_facing.)
static bool mesh_closed_outward(const tg_mesh* mesh) {
    const int nt = mesh->n_tris, nv = mesh->n_verts;
    if (nt < 4 || nv < 4) return false;
    std::vector<int> canon(nv);
    {
        std::vector<int> order(nv);
        for (int i = 0; i < nv; ++i) order[i] = i;
        auto key = [&](int i) { return std::make_tuple(mesh->verts[3 * i], mesh->verts[3 * i + 1], mesh->verts[3 * i + 2]); };
        std::sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
        for (int k = 0; k < nv; ++k) canon[order[k]] = (k > 0 && key(order[k]) == key(order[k - 1])) ? canon[order[k - 1]] : order[k];
    }
    std::map<std::pair<int, int>, int> edge;       // directed edge -> triangle
    std::vector<int> parent(nt);
    for (int t = 0; t < nt; ++t) parent[t] = t;
    auto find = [&](int x) { while (parent[x] != x) x = parent[x] = parent[parent[x]]; return x; };
    for (int t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) {
            const int a = canon[mesh->tris[3 * t + k]], b = canon[mesh->tris[3 * t + (k + 1) % 3]];
            if (a == b) return false;                                   // degenerate triangle
            if (!edge.emplace(std::make_pair(a, b), t).second) return false;   // the same directed edge twice: inconsistent winding
        }
    for (const auto& e : edge) {
        const auto opp = edge.find(std::make_pair(e.first.second, e.first.first));
        if (opp == edge.end()) return false;                            // open boundary
        parent[find(e.second)] = find(opp->second);
    }
    std::map<int, double> vol;
    for (int t = 0; t < nt; ++t) {
        const float* a = mesh->verts + 3 * mesh->tris[3 * t]; const float* b = mesh->verts + 3 * mesh->tris[3 * t + 1]; const float* c = mesh->verts + 3 * mesh->tris[3 * t + 2];
        vol[find(t)] += (double)a[0] * ((double)b[1] * c[2] - (double)b[2] * c[1]) - (double)a[1] * ((double)b[0] * c[2] - (double)b[2] * c[0]) +
                        (double)a[2] * ((double)b[0] * c[1] - (double)b[1] * c[0]);
    }
    for (const auto& v : vol) if (!(v.second > 0.0)) return false;      // a component wound inside out
    return true;
}

template <typename T> static int build_dev_robot(const tg_robot& r, DevRobot<T>& d) {
    memset(&d, 0, sizeof d);
    const int N = r.ndof;
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) { d.jpos[i][k] = (T)r.joint_pos[i][k]; d.jaxis[i][k] = (T)r.joint_axis[i][k]; }
        for (int k = 0; k < 9; ++k) d.jrot[i][k] = (T)r.joint_rot[i][k];
        // merge the bodies welded to link i: m, com, inertia about com in link coordinates
        double m = 0, com[3] = {0, 0, 0};
        for (int b = 0; b < TG_MAX_BODIES_PER_LINK; ++b) {
            m += r.body_mass[i][b];
            for (int k = 0; k < 3; ++k) com[k] += r.body_mass[i][b] * r.body_com[i][b][k];
        }
        if (m > 0) for (int k = 0; k < 3; ++k) com[k] /= m;
        double I[3][3] = {{0}};
        for (int b = 0; b < TG_MAX_BODIES_PER_LINK; ++b) {
            const double mb = r.body_mass[i][b];
            if (mb <= 0) continue;
            const double* R = r.body_rot[i][b];
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < 3; ++k) I[a][c] += R[3 * a + k] * r.body_inertia[i][b][k] * R[3 * c + k];
            double dv[3];
            for (int k = 0; k < 3; ++k) dv[k] = r.body_com[i][b][k] - com[k];
            const double dd = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) I[a][c] += mb * ((a == c ? dd : 0.0) - dv[a] * dv[c]);
        }
        // FK constants and the merged angular-damping inertia
        {
            const double* Rj = r.joint_rot[i];
            const double* a = r.joint_axis[i];
            const double aaT[9] = {a[0] * a[0], a[0] * a[1], a[0] * a[2], a[1] * a[0], a[1] * a[1], a[1] * a[2], a[2] * a[0], a[2] * a[1], a[2] * a[2]};
            const double ax[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
            for (int rr = 0; rr < 3; ++rr)
                for (int cc = 0; cc < 3; ++cc) {
                    double sa = 0, sb = 0, sc = 0;
                    for (int k = 0; k < 3; ++k) {
                        sa += Rj[3 * rr + k] * aaT[3 * k + cc];
                        sb += Rj[3 * rr + k] * ((k == cc ? 1.0 : 0.0) - aaT[3 * k + cc]);
                        sc += Rj[3 * rr + k] * ax[3 * k + cc];
                    }
                    d.fkA[i][3 * rr + cc] = (T)sa; d.fkB[i][3 * rr + cc] = (T)sb; d.fkC[i][3 * rr + cc] = (T)sc;
                }
            double Ia[3][3] = {{0}};
            for (int b = 0; b < TG_MAX_BODIES_PER_LINK; ++b) {
                if (r.body_mass[i][b] <= 0) continue;
                const double* R = r.body_rot[i][b];
                for (int aa = 0; aa < 3; ++aa)
                    for (int c = 0; c < 3; ++c)
                        for (int k = 0; k < 3; ++k) Ia[aa][c] += R[3 * aa + k] * r.body_inertia[i][b][k] * R[3 * c + k];
            }
            d.lang[i][0] = (T)Ia[0][0]; d.lang[i][1] = (T)Ia[0][1]; d.lang[i][2] = (T)Ia[0][2];
            d.lang[i][3] = (T)Ia[1][1]; d.lang[i][4] = (T)Ia[1][2]; d.lang[i][5] = (T)Ia[2][2];
        }
        d.lmass[i] = (T)m;
        for (int k = 0; k < 3; ++k) d.lcom[i][k] = (T)com[k];
        d.linert[i][0] = (T)I[0][0]; d.linert[i][1] = (T)I[0][1]; d.linert[i][2] = (T)I[0][2];
        d.linert[i][3] = (T)I[1][1]; d.linert[i][4] = (T)I[1][2]; d.linert[i][5] = (T)I[2][2];
        for (int b = 0; b < TG_MAX_BODIES_PER_LINK; ++b) {
            d.bmass[i][b] = (T)r.body_mass[i][b];
            for (int k = 0; k < 3; ++k) { d.bcom[i][b][k] = (T)r.body_com[i][b][k]; d.binert[i][b][k] = (T)r.body_inertia[i][b][k]; }
            for (int k = 0; k < 9; ++k) d.brot[i][b][k] = (T)r.body_rot[i][b][k];
        }
        d.rest_q[i] = (T)r.rest_q[i];
    }
    d.tcp_link = r.tcp_link; d.sensor_link = r.sensor_link;
    for (int k = 0; k < 3; ++k) { d.tcp_pos[k] = (T)r.tcp_pos[k]; d.sensor_pos[k] = (T)r.sensor_pos[k]; d.gravity[k] = (T)r.gravity[k]; }
    for (int k = 0; k < 9; ++k) { d.tcp_rot[k] = (T)r.tcp_rot[k]; d.sensor_rot[k] = (T)r.sensor_rot[k]; }
    d.lin_damp = (T)r.linear_damping; d.ang_damp = (T)r.angular_damping; d.joint_damp = (T)r.joint_damping;
    d.max_force = (T)r.max_force; d.pos_gain = (T)r.pos_gain; d.vel_gain = (T)r.vel_gain;
    // Upper bound of trace(M(q)) over all joint angles (sim_tick's a-priori no-clamp test, tg_physics.hpp): M_ii is the inertia of
    // the subtree of joint i about its axis <= sum over the subtree's links of trace(I_l) + m_l D^2, with D <= the summed lengths of the
    // joint offsets on the way plus the link's own centre-of-mass offset.
    {
        auto parent = [&](int i) { return r.topology == 0 ? Topo<0>::parent(i) : Topo<1>::parent(i); };
        auto len3 = [](const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
        double tb = 0.0, dmax = 0.0;
        for (int i = 0; i < kMaxDof; ++i) d.diag_sqrt[i] = (T)0;
        for (int i = 0; i < N; ++i) {
            double di = 0.0;
            for (int l = i; l < N; ++l) {
                double D = 0.0; int k = l; bool under = false;
                while (k >= 0) { if (k == i) { under = true; break; } D += len3(r.joint_pos[k]); k = parent(k); }
                if (!under) continue;
                const double com[3] = {(double)d.lcom[l][0], (double)d.lcom[l][1], (double)d.lcom[l][2]};
                D += len3(com);
                di += ((double)d.linert[l][0] + (double)d.linert[l][3] + (double)d.linert[l][5]) + (double)d.lmass[l] * D * D;
            }
            tb += di;                                      // d_i >= M_ii(q): inertia of joint i's subtree about its axis
            d.diag_sqrt[i] = (T)(std::sqrt(di) * 1.0000001);
            dmax = di > dmax ? di : dmax;
        }
        d.trace_bound = (T)tb;
        d.diag_sqrt_max = (T)(std::sqrt(dmax) * 1.0000001);
    }
    d.res_thr = (T)0;   // tg_config.solver_residual_threshold: set by tg_create; the function-level entry points run the default solver
    return 0;
}

static int check_robot(const tg_robot* r) {
    if (!r) return fail(-1, "robot is NULL");
    if (r->topology == 0 && r->ndof != Topo<0>::N) return fail(-1, "topology 0 (serial chain) is built for ndof = 6");
    if (r->topology == 1 && r->ndof != Topo<1>::N) return fail(-1, "topology 1 (MG400 tree) needs ndof = 8");
    if (r->topology != 0 && r->topology != 1) return fail(-1, "unknown robot topology");
    if (r->tcp_link < 0 || r->tcp_link >= r->ndof || r->sensor_link < 0 || r->sensor_link >= r->ndof) return fail(-1, "frame link out of range");
    return 0;
}
}  // namespace tg

struct tg_ctx {
    tg_config cfg;
    tg_robot robot;
    int H, W, act_dim;
    hipStream_t own_stream = nullptr, stream = nullptr;
    bool tmpl_ready = false;               // object_balance: State.reset_tmpl has been (or will have been, in stream order) filled by a full reset
    hipStream_t capture_stream = nullptr;   // the step graph is captured here, never on the stream work runs on (see tg_step)
    void *d_robot = nullptr, *d_const = nullptr;   // DevRobot<T>, EnvConst<T>
    // broadphase guard (tg_set_broadphase; tg_broadphase.hip): device scene + hull vertices, per-env results [3][n], totals {env-checks, pairs, hits}
    tg::BpScene* d_bp = nullptr;
    double* d_bp_hull = nullptr;
    int32_t* d_bp_out = nullptr;
    unsigned long long* d_bp_tot = nullptr;
    bool bp_every_step = false;
    tg::State st{};
    tg::RasterParams rp{};
    float *d_nodef_dep = nullptr, *d_verts = nullptr, *d_soup = nullptr, *d_actions = nullptr;
    uint8_t* d_nodef_gray = nullptr;   // uint8(nodef_gray)
    uint8_t *d_border = nullptr, *d_obs = nullptr, *d_term = nullptr, *d_mask = nullptr;
    size_t packed_obs_bytes = 0, packed_bytes = 0, packed_feature_off = 0;   // d_obs = [obs | pad to 16 | reward f32[n] | done u8[n] | pad to 4 | feature f32[n][12]]
    int32_t* d_tris = nullptr;
    int n_tris = 0;
    tg::Stimulus stim{};
    // scene camera (tg_set_scene): shared triangle set, per-env eye<-frame transforms, rgb images
    tg::SceneParams scene{};
    tg::SceneView scene_view{};
    bool scene_on = false, scene_every_step = false;
    float *d_scene_verts = nullptr, *d_scene_xf = nullptr, *d_scene_spheres = nullptr;
    int32_t* d_scene_tris = nullptr;
    uint32_t *d_scene_attr = nullptr, *d_scene_local = nullptr;
    unsigned long long* d_scene_static = nullptr;
    tg::SceneChunk* d_scene_chunks = nullptr;
    uint8_t *d_vis = nullptr, *d_vis_term = nullptr;
    uint8_t* d_episode = nullptr;     // [ep_return f64[n] | ep_final_return f32[n] | ep_final_len i32[n]] (tg_get_episode_stats)
    void* d_block_tables = nullptr;   // k_render_blocks' tables (rp.blockmax, rp.tmpl point into it)
    uint8_t* d_tile_tmpl = nullptr;   // tile-sparse payload (tg_pack_tiles): the image every env shows without a contact - zero inside, the pasted ring outside
    int32_t *d_int_idx = nullptr, *d_int_rank = nullptr;   // interior-only payload: pixel of interior position k / interior position of pixel p (-1: ring)
    int n_interior = 0;
    float* d_oracle = nullptr;        // [n][34] observation_mode "oracle" vectors (tg_get_obs_oracle), allocated on first use
    float* d_oracle_term = nullptr;   // tg_enable_oracle_obs: the step's own vectors (before any reset): rows of finished envs = terminal observation
    bool oracle_every_step = false;
    bool cfg_turn_off_border = false;
    // hipGraph of one tg_step launch sequence, keyed by the device action pointer it was captured with (launch-bound inner loop:
    // 3-4 kernels per step, one graph launch instead)
    hipStream_t aux_stream = nullptr;                // object_balance: the reset of finished envs runs here, beside the render
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Render targets (tg_set_obs_targets, round 5): the tactile images of a step land in the context's own buffer (target 0) or in one of up to two
    // caller-owned buffers (targets 1, 2: rank 0's blocks of the two alternating gathered batches, parallel.py) - each with its own changed-block
    // record and its own captured graphs, since the destination is a kernel argument.
    uint8_t* obs_ext[2] = {nullptr, nullptr};
    unsigned long long* drawn_ext[2] = {nullptr, nullptr};
    int obs_sel = 0;                                             // 0 own buffer, 1 / 2 = obs_ext[0 / 1]
    hipGraphExec_t step_graph_t[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // [target][0 reads d_actions, 1 the pinned caller-owned device buffer]
    hipGraphExec_t* step_graph = step_graph_t[0];                // the selected target's pair
    const float* step_graph_actions[2] = {nullptr, nullptr};
    hipStream_t step_graph_stream[2] = {nullptr, nullptr};
    bool graph_broken = false;
    // tg_step_random: the policy of a random-action rollout (action_space.sample() for the whole batch) inside the step's graph
    unsigned long long* d_draw = nullptr;      // [0] draw counter, [1] seed, [2] ticket of the sampler's last-block election
    hipGraphExec_t random_graph_t[3] = {nullptr, nullptr, nullptr};
    uint64_t random_seed = 0;
    // reset bank (edge_follow / surface_follow, auto_reset; tg_kernels.hpp: BankAux)
    tg::State bk{};                    // the bank view: st's layout, the reset-written arrays in allocations of the bank's own
    tg::BankAux aux{};
    int bank_mode = 0;                 // 0 off, 1 refills on bank_stream every bank_every steps, 2 as 1 and waited for (tests)
    uint8_t* h_rows = nullptr;         // tg_copy_obs_rows: pinned staging block
    size_t h_rows_bytes = 0;
    int bank_every = 8;
    static constexpr int kBankRing = 16, kBankLag = 8;    // bank_refill: markers on the step stream, one per visit; the host stays <= kBankLag visits ahead
    hipEvent_t ev_bank_ring[kBankRing] = {};
    unsigned long long bank_visits = 0;
    long long bank_steps = 0;
    hipStream_t bank_stream = nullptr;
    hipEvent_t ev_bank = nullptr, ev_bank_done = nullptr;
    std::vector<void*> bank_allocs;
    void* d_bank = nullptr;            // BankDev {bk, aux} in device memory (k_reset's argument)
    // profiling
    bool profile = false;          // tg_profile_enable(1): HIP event pairs around every launch class, no graph
    bool profile_clock = false;    // tg_profile_enable(2): the kernels' own clock only (tg_kt.hpp): the step stays ONE graph, reduce nodes behind its scopes
    struct Ev { hipEvent_t a, b; int which; };
    std::vector<Ev> events;
    double prof_ms[6] = {0, 0, 0, 0, 0, 0};      // HIP events: step, render (k_step_render when fused), reset sequence, masked render, scene camera, an EMPTY
    int64_t prof_n[6] = {0, 0, 0, 0, 0, 0};      // event pair (what every figure before it carries on top of its kernels)
    // one launch per step (tg_fused.hip): -1 = TG_FUSED_STEP=0, 1 = TG_FUSED_STEP=1, 0 = where it measures faster (use_fused_step)
    int fused_pref = 0;
    // profiling by the kernels' own clock (tg_kt.hpp): per-wavefront {start, end} slots, reduced after every timed scope into {ticks, scopes}
    unsigned long long* d_kt = nullptr;          // [kt_slots][2]
    unsigned long long* d_kt_acc = nullptr;      // [8][2]
    size_t kt_slots = 0;
    double wall_clock_khz = 100000.0;
};

static inline bool env_has_feature(int env_kind) {   // envs with an extended_feature observation (push 12, roll 3, surface_follow -v1 / -v2 6 of the 12-wide rows)
    return env_kind == TG_ENV_OBJECT_PUSH || env_kind == TG_ENV_OBJECT_ROLL || env_kind == TG_ENV_SURFACE_FOLLOW_AUTO;
}

namespace tg {
static void drain_events(tg_ctx* c) {
    for (auto& e : c->events) {
        (void)hipEventSynchronize(e.b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e.a, e.b);
        c->prof_ms[e.which] += ms;
        c->prof_n[e.which] += 1;
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    c->events.clear();
}
#define TG_DISPATCH(ctx_dtype, ctx_topo, CALL)                                               \
    do {                                                                                     \
        if ((ctx_dtype) == TG_PHYSICS_F64) {                                                 \
            if ((ctx_topo) == 0) { CALL(double, 0); } else { CALL(double, 1); }              \
        } else {                                                                             \
            if ((ctx_topo) == 0) { CALL(float, 0); } else { CALL(float, 1); }                \
        }                                                                                    \
    } while (0)

static inline uint8_t* obs_buf(const tg_ctx* c) { return c->obs_sel == 0 ? c->d_obs : c->obs_ext[c->obs_sel - 1]; }   // where this step's images go
static inline RasterParams raster_params(const tg_ctx* c) {   // ... and the changed-block record that belongs to that buffer
    RasterParams P = c->rp;
    if (c->obs_sel != 0) P.drawn = c->drawn_ext[c->obs_sel - 1];
    return P;
}
// SoA [field][n] device -> AoS [n][field] host
template <typename T> static int fetch_soa(tg_ctx* c, const T* dev, int fields, T* host) {
    const int n = c->cfg.num_envs;
    std::vector<T> tmp((size_t)fields * n);
    hipError_t e = hipMemcpyAsync(tmp.data(), dev, tmp.size() * sizeof(T), hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return fail(-2, hipGetErrorString(e));
    e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(-2, hipGetErrorString(e));
    for (int f = 0; f < fields; ++f)
        for (int i = 0; i < n; ++i) host[(size_t)i * fields + f] = tmp[(size_t)f * n + i];
    return 0;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : -1; }
};

template <typename T> static int upload_robot(const tg_robot* robot, DevBuf& buf) {
    DevRobot<T> dr;
    build_dev_robot(*robot, dr);
    if (buf.alloc(sizeof dr)) return fail(-2, "hipMalloc failed");
    if (hipMemcpy(buf.p, &dr, sizeof dr, hipMemcpyHostToDevice) != hipSuccess) return fail(-2, "hipMemcpy failed");
    return 0;
}

static int need_device() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(-3, "no HIP device visible — no CPU fallback");
    return 0;
}
bool use_fused_step(const tg_ctx* c);                        // tg_api.hip
}  // namespace tg

void drop_step_graphs(tg_ctx* c);                            // tg_api.hip: every captured step graph of every render target (captured again on the next step)

// Every entry point that touches the device first makes the context's device current: the caller may have switched devices
// (torch.cuda.set_device, another thread) since tg_create.
#define TG_ENTER(ctx)                                                                                        \
    do {                                                                                                     \
        int dev_ = -1;                                                                                       \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ != (ctx)->cfg.device) TG_HIP(hipSetDevice((ctx)->cfg.device)); \
    } while (0)
