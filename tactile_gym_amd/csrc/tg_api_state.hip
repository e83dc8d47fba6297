// tg_api_state.hip - the C ABI's read side (include/tactile_gym_hip.h): state read-back and set-up of joint states, inspection (step mode, bank /
// episode statistics), profiling, and the broadphase guard's entry points.  Split out of tg_api.hip in round 6 (tg_ctx.hpp is what they share).
#include "tg_ctx.hpp"

namespace tg {
template <typename T, int TOPO> static void launch_refresh_t(tg_ctx* c) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_refresh<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st);
}

template <typename T, int TOPO> static void launch_refresh_rpy_t(tg_ctx* c) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_refresh_rpy<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st);
}
// tg_pack_done_rows: one 256-thread workgroup per env; a finished env's slot is the number of finished envs before it (a prefix count over the
// done flags: deterministic ascending order, no atomics), workgroup 0 also writes the header.  Everything goes straight into pinned host memory.
// tg_get_state, surface_follow: each env's live surface (State::hsel) out of the three thirds, [n][cells]; workgroup x = env
__global__ __launch_bounds__(256) void k_gather_live_surface(int n, int cells, const uint8_t* __restrict__ hsel, const double* __restrict__ heights,
                                                             const float* __restrict__ zoff, double* __restrict__ out_h, float* __restrict__ out_z) {
    const int env = blockIdx.x;
    const size_t idx = (size_t)(hsel[env] & 3) * n + env;
    if (out_h != nullptr) for (int k = threadIdx.x; k < cells; k += 256) out_h[(size_t)env * cells + k] = heights[idx * cells + k];
    if (out_z != nullptr && threadIdx.x == 0) out_z[env] = zoff[idx];
}
__global__ __launch_bounds__(256) void k_pack_done_rows(const uint8_t* __restrict__ done, const uint8_t* __restrict__ term, const float* __restrict__ ep_ret,
                                                        const int32_t* __restrict__ ep_len, int n, int cap, size_t img, uint8_t* __restrict__ dst) {
    __shared__ int part[256];
    const int env = blockIdx.x, t = threadIdx.x;
    const bool mine = done[env] != 0;
    if (!mine && env != 0) return;
    const int upto = env == 0 ? n : env;                       // workgroup 0 counts them all (the header's count); the others the ones before them
    int cnt = 0;
    for (int i = t; i < upto; i += 256) cnt += done[i] != 0 ? 1 : 0;
    part[t] = cnt;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (t < k) part[t] += part[t + k]; __syncthreads(); }
    const int total = part[0];
    uint32_t* hdr = reinterpret_cast<uint32_t*>(dst);
    if (env == 0 && t == 0) { hdr[0] = (uint32_t)total; hdr[1] = (uint32_t)cap; hdr[2] = (uint32_t)n; hdr[3] = 0x74674452u; }
    if (!mine) return;
    const int slot = env == 0 ? 0 : total;
    if (slot >= cap) return;
    int32_t* ids = reinterpret_cast<int32_t*>(dst + 16);
    float* ret = reinterpret_cast<float*>(dst + 16 + (size_t)cap * 4);
    int32_t* len = reinterpret_cast<int32_t*>(dst + 16 + (size_t)cap * 8);
    if (t == 0) { ids[slot] = env; ret[slot] = ep_ret[env]; len[slot] = ep_len[env]; }
    const size_t rows_off = (16 + (size_t)cap * 12 + 15) & ~(size_t)15;
    const uint4* src = reinterpret_cast<const uint4*>(term + (size_t)env * img);
    uint4* out = reinterpret_cast<uint4*>(dst + rows_off + (size_t)slot * img);
    for (size_t w = t; w < img / 16; w += 256) out[w] = src[w];
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_done_rows_bytes(tg_ctx* c, int32_t cap, int64_t* bytes) {
    if (!c || !bytes || cap < 1) return fail(-1, "tg_done_rows_bytes: bad argument");
    *bytes = (int64_t)(((16 + (size_t)cap * 12 + 15) & ~(size_t)15) + (size_t)cap * c->H * c->W);
    return 0;
}
int tg_pack_done_rows(tg_ctx* c, void* dst, int32_t cap) {
    if (!c || !dst || cap < 1 || ((uintptr_t)dst & 15)) return fail(-1, "tg_pack_done_rows: bad argument");
    if (((size_t)c->H * c->W) % 16 != 0) return fail(-1, "tg_pack_done_rows: image bytes not a multiple of 16");
    if (!c->cfg.auto_reset) return fail(-1, "tg_pack_done_rows: the context has no auto-reset (no terminal observations)");
    TG_ENTER(c);
    hipLaunchKernelGGL(k_pack_done_rows, dim3(c->cfg.num_envs), dim3(256), 0, c->stream, c->st.done, c->d_term, c->st.ep_final_return, c->st.ep_final_len,
                       c->cfg.num_envs, cap, (size_t)c->H * c->W, (uint8_t*)dst);
    TG_HIP(hipGetLastError());
    return 0;
}
int tg_set_broadphase(tg_ctx* c, const tg_broadphase* g) {
    if (!c) return fail(-1, "NULL argument");
    TG_ENTER(c);
    TG_HIP(hipStreamSynchronize(c->stream));
    drop_step_graphs(c);                         // the guard is (or stops being) a node of the step graphs
    if (c->d_bp) { (void)hipFree(c->d_bp); c->d_bp = nullptr; }
    if (c->d_bp_hull) { (void)hipFree(c->d_bp_hull); c->d_bp_hull = nullptr; }
    c->bp_every_step = false;
    if (!g) return 0;
    if (g->n_hull_verts < 0 || (g->n_hull_verts > 0 && !g->hull_verts) || !(g->margin >= 0.0) || !(g->sphere_half > 0.0))
        return fail(-1, "tg_set_broadphase: bad argument");
    const int N = c->robot.ndof;
    BpScene h{};
    for (int k = 0; k < TG_BP_SLOTS; ++k) {
        const tg_bp_box& b = g->box[k];
        if (b.src < TG_BP_NONE || b.src > TG_BP_BALL) return fail(-1, "tg_set_broadphase: unknown pose source");
        if (b.src == TG_BP_LINK && (b.link < -1 || b.link >= N)) return fail(-1, "tg_set_broadphase: link index out of range");
        if (b.src == TG_BP_LINK && (b.hull_off < 0 || b.hull_n < 0 || b.hull_off + b.hull_n > g->n_hull_verts)) return fail(-1, "tg_set_broadphase: hull range out of bounds");
        if ((b.src == TG_BP_BODY || b.src == TG_BP_SPHERE) && c->st.body_pos == nullptr) return fail(-1, "tg_set_broadphase: this env has no free body");
        if (b.src == TG_BP_BALL && c->st.ball == nullptr) return fail(-1, "tg_set_broadphase: this env has no ball");
        if (b.src == TG_BP_EDGE && c->cfg.env_kind != TG_ENV_EDGE_FOLLOW) return fail(-1, "tg_set_broadphase: TG_BP_EDGE outside edge_follow");
        if (b.conj < -1 || b.conj >= TG_BP_SLOTS || (b.conj >= 0 && g->box[b.conj].src == TG_BP_NONE)) return fail(-1, "tg_set_broadphase: bad conj slot");
        h.box[k] = b;
    }
    h.margin = g->margin; h.hull_margin = g->hull_margin; h.sphere_half = g->sphere_half; h.ball_radius = g->ball_radius;
    for (int k = 0; k < 3; ++k) h.stim_pos[k] = c->cfg.stim_pos[k];
    h.table_slot = 16; h.has_ball = c->st.ball != nullptr;
    const size_t hb = (size_t)std::max(g->n_hull_verts, 1) * 3 * 8;
    TG_HIP(hipMalloc(&c->d_bp_hull, hb));
    if (g->n_hull_verts > 0) TG_HIP(hipMemcpy(c->d_bp_hull, g->hull_verts, (size_t)g->n_hull_verts * 3 * 8, hipMemcpyHostToDevice));
    h.hull = c->d_bp_hull;
    TG_HIP(hipMalloc(&c->d_bp, sizeof h)); TG_HIP(hipMemcpy(c->d_bp, &h, sizeof h, hipMemcpyHostToDevice));
    const size_t ob = (size_t)3 * c->cfg.num_envs * 4;
    if (!c->d_bp_out) TG_HIP(hipMalloc(&c->d_bp_out, ob));
    if (!c->d_bp_tot) TG_HIP(hipMalloc(&c->d_bp_tot, 3 * 8));
    TG_HIP(hipMemset(c->d_bp_out, 0, ob)); TG_HIP(hipMemset(c->d_bp_tot, 0, 3 * 8));
    c->bp_every_step = g->every_step != 0;
    return 0;
}
int tg_check_broadphase(tg_ctx* c) {
    if (!c) return fail(-1, "NULL argument");
    if (!c->d_bp) return fail(-1, "tg_check_broadphase: no guard (tg_set_broadphase)");
    TG_ENTER(c);
    if (launch_broadphase(c->cfg.physics_dtype, c->robot.topology, c->cfg.num_envs, c->stream, c->d_robot, c->d_bp, c->st, c->d_bp_out, c->d_bp_tot))
        return fail(-3, "tg_check_broadphase: unsupported topology");
    TG_HIP(hipGetLastError());
    return 0;
}
int tg_get_broadphase_totals(tg_ctx* c, int64_t* checks, int64_t* pairs, int64_t* hits) {
    if (!c || !checks || !pairs || !hits) return fail(-1, "NULL argument");
    *checks = *pairs = *hits = 0;
    if (!c->d_bp_tot) return 0;
    TG_ENTER(c);
    unsigned long long t[3];
    TG_HIP(hipMemcpyAsync(t, c->d_bp_tot, sizeof t, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    *checks = (int64_t)t[0]; *pairs = (int64_t)t[1]; *hits = (int64_t)t[2];
    return 0;
}

int tg_get_step_mode(tg_ctx* c, int32_t* mode, int32_t* envs_per_wavefront) {
    if (!c || !mode) return fail(-1, "NULL argument");
    *mode = use_fused_step(c) ? 1 : 0;
    if (envs_per_wavefront) *envs_per_wavefront = *mode ? fused_envs_per_wave(c->cfg.num_envs) : 0;
    return 0;
}

int tg_get_actions(tg_ctx* c, void** dev_actions) {
    if (!c || !dev_actions) return fail(-1, "NULL argument");
    *dev_actions = c->d_actions;
    return 0;
}

int tg_get_interior_count(tg_ctx* c, int32_t* k) {
    if (!c || !k) return fail(-1, "NULL argument");
    *k = c->cfg_turn_off_border ? -1 : 4 * c->n_interior;   // bytes per image; -1: the ring carries rendered values (turn_off_border), nothing to drop
    return 0;
}
int tg_get_bank_stats(tg_ctx* c, int64_t* swapped, int64_t* late, int32_t* mode) {
    if (!c || !swapped || !late || !mode) return fail(-1, "NULL argument");
    TG_ENTER(c);
    *swapped = 0; *late = 0; *mode = c->bank_mode;
    if (c->st.tmpl_stats != nullptr) {           // object_push: the reset template's counters (mode 3; k_reset_contact_wave)
        unsigned long long t[2] = {0, 0};
        TG_HIP(hipMemcpyAsync(t, c->st.tmpl_stats, 16, hipMemcpyDeviceToHost, c->stream));
        TG_HIP(hipStreamSynchronize(c->stream));
        *swapped = (int64_t)t[0]; *late = (int64_t)t[1]; *mode = c->st.reset_tmpl != nullptr ? 3 : 0;
        return 0;
    }
    if (c->bank_mode == 0) return 0;
    unsigned long long h[2] = {0, 0};
    TG_HIP(hipMemcpyAsync(h, c->aux.stats, 16, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    *swapped = (int64_t)h[0]; *late = (int64_t)h[1];
    return 0;
}

int tg_get_episode_stats(tg_ctx* c, void** ret_f32, void** len_i32) {
    if (!c || !ret_f32 || !len_i32) return fail(-1, "NULL argument");
    *ret_f32 = c->st.ep_final_return;
    *len_i32 = c->st.ep_final_len;
    return 0;
}
int tg_copy_episode_stats(tg_ctx* c, float* ret, int32_t* len) {
    if (!c || !ret || !len) return fail(-1, "NULL argument");
    TG_ENTER(c);
    TG_HIP(hipMemcpyAsync(ret, c->st.ep_final_return, (size_t)c->cfg.num_envs * 4, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipMemcpyAsync(len, c->st.ep_final_len, (size_t)c->cfg.num_envs * 4, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int tg_get_state(tg_ctx* c, const tg_state_view* v) {
    if (!c || !v) return fail(-1, "NULL argument");
    TG_ENTER(c);
    const int nd = c->robot.ndof;
    int rc = 0;
    if (v->q && (rc = fetch_soa(c, c->st.q, nd, v->q))) return rc;
    if (v->qd && (rc = fetch_soa(c, c->st.qd, nd, v->qd))) return rc;
    if (v->qd_target && (rc = fetch_soa(c, c->st.qd_target, nd, v->qd_target))) return rc;
    if (v->tcp_rpy && (c->cfg.env_kind == TG_ENV_EDGE_FOLLOW || c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO)) {   // k_step leaves this read-back to be recomputed on demand
#define CALL(T, TOPO) launch_refresh_rpy_t<T, TOPO>(c)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    }
    if (v->tcp_pos && (rc = fetch_soa(c, c->st.tcp_pos, 3, v->tcp_pos))) return rc;
    if (v->tcp_rpy && (rc = fetch_soa(c, c->st.tcp_rpy, 3, v->tcp_rpy))) return rc;
    if (v->edge_ang && (rc = fetch_soa(c, c->st.edge_ang, 1, v->edge_ang))) return rc;
    if (v->embed_dist && (rc = fetch_soa(c, c->st.embed, 1, v->embed_dist))) return rc;
    if (v->stim_xform && (rc = fetch_soa(c, c->st.stim_xform, 12, v->stim_xform))) return rc;
    if (v->step_count && (rc = fetch_soa(c, c->st.step_count, 1, v->step_count))) return rc;
    if (v->reset_ticks && (rc = fetch_soa(c, c->st.reset_ticks, 1, v->reset_ticks))) return rc;
    if (v->rng_state && (rc = fetch_soa(c, c->st.rng, 1, v->rng_state))) return rc;
    if (v->solver_sweeps && (rc = fetch_soa(c, c->st.sweeps, 1, v->solver_sweeps))) return rc;
    {   // broadphase guard results of the last check (zeros without a guard)
        int32_t* dst[3] = {v->broadphase_pairs, v->broadphase_hits, v->broadphase_mask};
        for (int k = 0; k < 3; ++k) {
            if (!dst[k]) continue;
            if (c->d_bp_out) { if ((rc = fetch_soa(c, c->d_bp_out + (size_t)k * c->cfg.num_envs, 1, dst[k]))) return rc; }
            else memset(dst[k], 0, (size_t)c->cfg.num_envs * 4);
        }
    }
    if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE) {
        if (v->body_pos && (rc = fetch_soa(c, c->st.body_pos, 3, v->body_pos))) return rc;
        if (v->body_rot && (rc = fetch_soa(c, c->st.body_rot, 9, v->body_rot))) return rc;
        if (v->body_linvel && (rc = fetch_soa(c, c->st.body_v, 3, v->body_linvel))) return rc;
        if (v->body_angvel && (rc = fetch_soa(c, c->st.body_w, 3, v->body_angvel))) return rc;
        if (v->gravity_z && (rc = fetch_soa(c, c->st.gravity, 1, v->gravity_z))) return rc;
        if (c->st.dish && v->dish_state && (rc = fetch_soa(c, c->st.dish, 20, v->dish_state))) return rc;
        if (c->st.ball) {
            if (v->ball_pos && (rc = fetch_soa(c, c->st.ball, 3, v->ball_pos))) return rc;
            if (v->ball_linvel && (rc = fetch_soa(c, c->st.ball + (size_t)3 * c->cfg.num_envs, 3, v->ball_linvel))) return rc;
            if (v->ball_angvel && (rc = fetch_soa(c, c->st.ball + (size_t)6 * c->cfg.num_envs, 3, v->ball_angvel))) return rc;
            if (v->ball_impulse && (rc = fetch_soa(c, c->st.ball + (size_t)12 * c->cfg.num_envs, 1, v->ball_impulse))) return rc;
        }
    }
    if (c->cfg.env_kind == TG_ENV_OBJECT_ROLL) {
        if (v->body_pos && (rc = fetch_soa(c, c->st.body_pos, 3, v->body_pos))) return rc;
        if (v->body_rot && (rc = fetch_soa(c, c->st.body_rot, 9, v->body_rot))) return rc;
        if (v->body_linvel && (rc = fetch_soa(c, c->st.body_v, 3, v->body_linvel))) return rc;
        if (v->body_angvel && (rc = fetch_soa(c, c->st.body_w, 3, v->body_angvel))) return rc;
        if (v->goal_pos && (rc = fetch_soa(c, c->st.goal, 3, v->goal_pos))) return rc;          // goal_pos_tcp
        if (v->obj_mass && (rc = fetch_soa(c, c->st.obj_mass, 1, v->obj_mass))) return rc;      // the episode's radius
    }
    if (c->cfg.env_kind == TG_ENV_OBJECT_PUSH) {
        if (v->body_pos && (rc = fetch_soa(c, c->st.body_pos, 3, v->body_pos))) return rc;
        if (v->body_rot && (rc = fetch_soa(c, c->st.body_rot, 9, v->body_rot))) return rc;
        if (v->body_linvel && (rc = fetch_soa(c, c->st.body_v, 3, v->body_linvel))) return rc;
        if (v->body_angvel && (rc = fetch_soa(c, c->st.body_w, 3, v->body_angvel))) return rc;
        if (v->traj && (rc = fetch_soa(c, c->st.traj, 3 * TG_MAX_TRAJ_POINTS, v->traj))) return rc;
        if (v->goal_id && (rc = fetch_soa(c, c->st.goal_id, 1, v->goal_id))) return rc;
        if (v->obj_mass && (rc = fetch_soa(c, c->st.obj_mass, 1, v->obj_mass))) return rc;
    }
    if (v->contact_count || v->contact_ids) {
        // contact pairs of the last sim tick, in solver row order: the cube vertices on the table in vertex order (ids 0-7; the marble's
        // single table contact is id 0), then the tip contact (id 8 + index of the tip-core hull vertex that made it; 8 for the marble)
        const int n = c->cfg.num_envs;
        std::vector<int32_t> code(n);
        if ((rc = fetch_soa(c, c->st.contact_code, 1, code.data()))) return rc;
        for (int i = 0; i < n; ++i) {
            int cnt = 0;
            int32_t ids[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
            for (int b = 0; b < 8 && cnt < 4; ++b) if ((code[i] >> b) & 1) ids[cnt++] = b;
            if ((code[i] >> 30) & 1) {   // tg_config.narrowphase != 0: bits 8-11 = the live slots of the tip - cube manifold, ids 8 + slot
                for (int k = 0; k < 4; ++k) if ((code[i] >> (8 + k)) & 1) ids[cnt++] = 8 + k;
            } else if ((code[i] >> 8) & 1) ids[cnt++] = 8 + (code[i] >> 9);
            if (v->contact_count) v->contact_count[i] = cnt;
            if (v->contact_ids) for (int k = 0; k < 8; ++k) v->contact_ids[(size_t)i * 8 + k] = ids[k];
        }
    }
    if (c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        if (v->goal_pos && (rc = fetch_soa(c, c->st.goal, 3, v->goal_pos))) return rc;
        if (v->direction && (rc = fetch_soa(c, c->st.dir, 2, v->direction))) return rc;
        if (v->surf_zoff || v->heights) {
            const int n = c->cfg.num_envs, cells = c->cfg.surf_rows * c->cfg.surf_cols;
            DevBuf hh, zz;
            if ((v->heights && hh.alloc((size_t)n * cells * 8)) || (v->surf_zoff && zz.alloc((size_t)n * 4))) return fail(-2, "hipMalloc failed");
            hipLaunchKernelGGL(k_gather_live_surface, dim3(n), dim3(256), 0, c->stream, n, cells, c->st.hsel, c->st.heights, c->st.surf_zoff,
                               v->heights ? (double*)hh.p : nullptr, v->surf_zoff ? (float*)zz.p : nullptr);
            if (v->heights) TG_HIP(hipMemcpyAsync(v->heights, hh.p, (size_t)n * cells * 8, hipMemcpyDeviceToHost, c->stream));
            if (v->surf_zoff) TG_HIP(hipMemcpyAsync(v->surf_zoff, zz.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            TG_HIP(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

int tg_set_joint_state(tg_ctx* c, const double* q, const double* qd) {
    if (!c || !q || !qd) return fail(-1, "NULL argument");
    TG_ENTER(c);
    if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE || c->cfg.env_kind == TG_ENV_OBJECT_PUSH || c->cfg.env_kind == TG_ENV_OBJECT_ROLL)
        return fail(-1, "tg_set_joint_state: not supported for envs with a free object");
    const int n = c->cfg.num_envs, nd = c->robot.ndof;
    std::vector<double> a((size_t)TG_MAX_DOF * n, 0.0), b((size_t)TG_MAX_DOF * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int f = 0; f < nd; ++f) { a[(size_t)f * n + i] = q[(size_t)i * nd + f]; b[(size_t)f * n + i] = qd[(size_t)i * nd + f]; }
    TG_HIP(hipMemcpyAsync(c->st.q, a.data(), a.size() * 8, hipMemcpyHostToDevice, c->stream));
    TG_HIP(hipMemcpyAsync(c->st.qd, b.data(), b.size() * 8, hipMemcpyHostToDevice, c->stream));
    TG_HIP(hipMemsetAsync(c->st.licence, 0, (size_t)n * 4, c->stream));   // new configuration: full solve and exact sines / cosines next
#define CALL(T, TOPO) launch_refresh_t<T, TOPO>(c)
    TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    launch_render(raster_params(c), c->stim, c->st.stim_xform, 1, c->cfg.num_envs, nullptr, c->d_nodef_dep, c->d_nodef_gray, c->d_border, obs_buf(c),
                  nullptr, nullptr, nullptr, nullptr, c->stream);          // the observation of the new configuration
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_profile_enable(tg_ctx* c, int32_t enable) {
    if (!c) return fail(-1, "NULL ctx");
    (void)hipStreamSynchronize(c->stream);
    drain_events(c);
    c->profile = enable == 1;
    if (c->profile_clock != (enable == 2)) {
        // the step graphs carry the slot pointer (or its absence) in their kernel arguments: captured again on the next step
        drop_step_graphs(c);
        c->profile_clock = enable == 2;
    }
    for (int k = 0; k < 6; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    if (enable && !c->d_kt) {
        // slots for the largest launch of this context: a render of every env's image in 64-row tiles, two passes, four wavefronts each
        const size_t n = (size_t)c->cfg.num_envs;
        c->kt_slots = std::max<size_t>(n + 64, (size_t)((c->rp.W + 63) / 64) * (size_t)((c->rp.H + 63) / 64) * n * 8);
        TG_HIP(hipMalloc(&c->d_kt, c->kt_slots * 16));
        TG_HIP(hipMalloc(&c->d_kt_acc, 8 * 16));
        std::vector<unsigned long long> init(c->kt_slots * 2);
        for (size_t i = 0; i < c->kt_slots; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
        TG_HIP(hipMemcpy(c->d_kt, init.data(), c->kt_slots * 16, hipMemcpyHostToDevice));
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->cfg.device) == hipSuccess && khz > 0) c->wall_clock_khz = (double)khz;
    }
    if (c->d_kt_acc) TG_HIP(hipMemset(c->d_kt_acc, 0, 8 * 16));
    return 0;
}
// which: 0 step kernel, 1 render (the one launch of a fused step), 2 reset sequence, 3 masked render, 4 scene camera, 5 an empty event pair: by
// HIP events on the launch stream; 8 + k (k = 0 .. 3): class k by the kernels' own clock (tg_kt.hpp: first wavefront start -> last wavefront end).
int tg_profile_get(tg_ctx* c, int32_t which, double* total_ms, int64_t* launches) {
    if (!c || which < 0 || (which > 5 && which < 8) || which > 11) return fail(-1, "bad argument");
    (void)hipStreamSynchronize(c->stream);
    drain_events(c);
    if (which >= 8) {
        unsigned long long h[2] = {0, 0};
        if (c->d_kt_acc) TG_HIP(hipMemcpy(h, c->d_kt_acc + 2 * (which - 8), 16, hipMemcpyDeviceToHost));
        if (total_ms) *total_ms = (double)h[0] / c->wall_clock_khz;
        if (launches) *launches = (int64_t)h[1];
        return 0;
    }
    if (total_ms) *total_ms = c->prof_ms[which];
    if (launches) *launches = c->prof_n[which];
    return 0;
}

}  // extern "C"
