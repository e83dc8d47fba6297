// tg_kernels.hpp - device side of libtactile_gym_hip.so: task constants, per-env SoA state, the controller / reward / camera helpers
// and the lane-per-env step / reset kernels (one lane = one env).  Included by tg_api.hip (C ABI + launches) and by
// tg_contact_wave.hip (the wave-per-env contact solver of object_push / object_roll).
//
// Per-env state lives in HBM as struct-of-arrays with the env index minor ([field][num_envs], doubles), so a
// wavefront of 64 consecutive envs reads or writes one 512-byte contiguous run per field: the whole dynamic state
// crosses HBM exactly once per env step (in) and once (out); the 24 sim ticks in between run out of registers.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/tactile_gym_hip.h"
#include "tg_physics.hpp"
#include "tg_kt.hpp"

namespace tg {

// ------------------------------------------------------------------------------------------------ task constants
template <typename T> struct EnvConst {
    int num_envs, act_dim, movement_mode, noise_mode, reward_mode, max_steps, action_repeat, solver_iters;
    T dt, min_action, max_action, act_lo[6], act_hi[6], tcp_lims[6][2];
    T work_pos[3];
    Q4<T> work_q, work_qinv;
    M3<T> work_R, work_Rinv;
    T work_inv_pos[3];
    T stim_pos[3], edge_height, edge_len, term_dist, embed_default;
    double embed_lo, embed_hi;   // random draws are evaluated in double on every path
    M3<T> cam_rot;               // R(cam_rpy) in the sensor-body frame
    T cam_pos[3];
    // surface_follow
    int env_kind, surf_rows, surf_cols, surf_goal, surf_vertical;
    M3<T> stim_R;                // surface_follow-v2: rotation of the upright heightfield (identity otherwise)
    double surf_scale, surf_range, surf_interp, surf_extent, auto_scale;
    double xbin_lo, xbin_hi, ybin_lo, ybin_hi;   // np.linspace bounds of x_bins / y_bins (base_surface_env.py:258-282)
    // object_balance
    BodyConst<T> body;
    BallConst<T> ball;           // object_balance, object_mode ball_on_plate
    SpinConst<T> spin;           // object_balance, object_mode spinning_plate (spin.n_dish > 0): `body` is the spool, the dish is the env's object
    M3<T> obj_init_rot;
    T obj_init_rpy_deg[3], obj_base_width, obj_base_height, term_deg, term_pos, ext_force[3];
    int rand_gravity, rand_embed;
    double gravity_lo, gravity_hi, gravity_default;
    int control_mode, max_blocking;   // TG_CONTROL_*; blocking_move's step cap in position control
    int fused_reset;             // edge_follow with auto_reset: k_reset keeps the terminal camera transform, one render launch draws both images
    // object_push
    PushScene<T> push;
    int traj_type, traj_n, rand_init_orn, rand_obj_mass, reset_goal_id;
    // object_roll
    int roll_rand_init_pos, roll_rand_size, roll_rand_embed;
    double roll_radius, roll_init_range, roll_goal_lo, roll_goal_hi;
    double traj_spacing, traj_max_perturb, traj_init_offset, mass_lo, mass_hi, init_orn_range, traj_ang_range, obj_mass0;
    T obj_init_pos[3];
    double obj_init_rpy[3];
};

struct State {   // device pointers, SoA [field][num_envs]
    double *q, *qd, *qd_target, *tcp_pos, *tcp_rpy, *edge_ang, *embed;
    float *stim_xform, *term_xform, *reward;   // term_xform: camera<-stimulus transform of the terminal observation (fused reset)
    int32_t *step_count, *reset_ticks;
    int32_t* licence;               // [n] env steps for which the analytic fixed point stays licensed without a new full solve (k_step)
    double* trig_sc;                // [16][n] sin / cos of the joint angles at the end of the last k_step (valid while the licence holds)
    double* edge_sc;                // [2][n] sin / cos of the episode's edge angle
    uint64_t* rng;
    uint8_t* done;
    // surface_follow
    double *dir, *goal, *heights;   // [2][n], [3][n], [3][n][rows*cols]: three surfaces per env - the live one, the last episode's, and a spare the reset bank fills
    uint8_t* hsel;                  // [n] surface_follow: bits 0-1 the live third of heights / surf_zoff, bits 2-3 the last episode's (round 6).  The finished episode's
                                    // surface stays where it is for the terminal image of that step (one fused render launch draws both images); the bank's precomputed
                                    // surface is already in the spare third (no 32 KB copy: the env moves there); a reset on the spot overwrites the last episode's
    double* accum;                  // [n] sparse reward: the episode's accumulated dense reward
    float* surf_zoff;               // [3][n], indexed like heights
    int64_t* noise_seed;            // [n]
    // object_balance
    double *body_pos, *body_rot, *body_v, *body_w, *ext_pos, *gravity;   // [3][n], [9][n], [3][n], [3][n], [3][n], [n]
    uint8_t* ext_pending;           // [n]
    unsigned long long* tmpl_stats; // [2] object_push: resets that took the reset template / that ran their blocking move (k_reset_contact_wave); null otherwise
    double* reset_tmpl;             // [2 N + 2] object_balance: the arm's state after Robot.reset (q, qd, ticks used, valid flag) - see k_reset_body
    double* ball;                   // [13][n] ball_on_plate: position, linear velocity, angular velocity, one-shot torque, last normal impulse
    double* dish;                   // [20][n] spinning_plate: the dish's base position (0-2), orientation (3-11, row major), linear (12-14) and angular (15-17)
                                    // velocity, the last tick's summed normal impulse (18) and number of contact points (19); null otherwise
    const double* spin_hulls;       // spinning_plate: [n_dish][3] the dish's hull in its base frame, then [n_spool][3] the spool's
    // object_push
    double *traj, *obj_mass;        // [3][TG_MAX_TRAJ_POINTS][n] work-frame x, y, yaw; [n]
    int32_t* goal_id;               // [n]
    int32_t* contact_code;          // [n] contact pairs of the last sim tick (sim_tick_push's contact_code)
    float *feature, *term_feature;  // [n][12] extended_feature observation (object_push_env.py:611-629), AoS
    // per-env episode statistics, what the reference's callers get from the Monitor wrapper (sb3_helpers/rl_utils.py:17-30, 59):
    double* ep_return;              // [n] sum of the float32 rewards handed out since the episode began
    float* ep_final_return;         // [n] the finished episode's return (valid where done)
    int32_t* ep_final_len;          // [n] its length in env steps
    const void* tip_verts;          // [n_tip][3] in the physics dtype
    unsigned long long* draw;       // tg_step_random on the lane-mapped k_step: {draw counter, seed, ticket}; the kernel draws its own actions (nullptr: reads `actions`)
    float* act_out;                 // ... and leaves them here ([n][act_dim])
    unsigned long long* kt;         // profiling mode: per-wavefront {start, end} wall-clock slots (tg_kt.hpp); null otherwise
    int32_t* sweeps;                // [n] threshold mode (tg_config.solver_residual_threshold > 0): PGS sweeps the ticks of the env's last step ran (tg_state_view.solver_sweeps)
    double* mani;                   // [37][n] object_push with tg_config.narrowphase != 0: the tip - cube contact manifold (la, lb, normal of 4 points; count)
#ifdef TG_TL_STAMPS
    unsigned long long* tl;         // development: [4][8192] launch-start stamps (wall clock) + [4] counters behind them
#endif
};
#ifdef TG_TL_STAMPS
#define TG_TL(ptr, k) do { if ((ptr) != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { \
    const unsigned long long i_ = atomicAdd((ptr) + 4 * 8192 + (k), 1ull); (ptr)[(k) * 8192 + (i_ & 8191)] = wall_clock64(); } } while (0)
#else
#define TG_TL(ptr, k)
#endif

// Reset bank (edge_follow / surface_follow with auto_reset).  These envs' reset is a pure function of the env's RNG stream
// (task draws, surface noise, rest pose -> IK -> blocking move: nothing of the finished episode enters), so the NEXT post-reset state of
// every env is computed ahead of time, on a second low-priority stream, into a second set of SoA arrays (the "bank view": a State whose
// pointers address the bank's own allocations).  tag[env] = the value of the env's main RNG state the bank entry was computed from; the
// entry is usable iff it equals the env's current RNG state (tg_seed, tg_reset(mask) and a late reset all move the stream on and thereby
// invalidate it without any bookkeeping).  The refill kernels publish the tag with an agent-scope release after the data; k_reset reads it
// with an acquire before the data, and releases the new RNG state after its reads - the only ordering between the two streams.
struct BankAux {
    unsigned long long* tag;     // [n]
    unsigned long long* rng_in;  // [n] refill: the RNG state phase 1 started from (published as the tag by the last phase)
    uint8_t* need;               // [n] refill: envs whose entry is being recomputed by this sequence of launches
    uint8_t* late;               // [n] step, surface_follow: finished envs whose bank entry was not ready: reset on the spot by the launches that follow (phase 2 clears it)
    uint8_t* swapped;            // [n] step, surface_follow: finished envs that took their bank entry: k_gen_surface copies the bank's heights and clears it
    unsigned long long* stats;   // [2] auto-resets that took a bank entry / that were done on the spot (tg_get_bank_stats)
    int enabled;
};
struct BankDev { State bk; BankAux aux; };   // in device memory: k_reset reads it only in lanes whose env has finished (no second State among the kernel arguments of every step)

// SplitMix64 (identical integer stream in oracle/ref_env.py: Rng)
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
__device__ inline double rng_uniform(uint64_t& s, double lo, double hi) {
#pragma clang fp contract(off)   // lo + (hi - lo) * u must round like the host's two-step evaluation (no FMA)
    s += kGolden;
    const double u = (double)(mix64(s) >> 11) * (1.0 / 9007199254740992.0);
    const double span = (hi - lo) * u;
    return lo + span;
}

// World pose of the TCP frame -> work-frame position / rpy, following the reference's chain of PyBullet helpers
// (base_robot_arm.py:62-75, 153-172): matrix -> quaternion -> euler -> quaternion -> multiply -> euler.
template <typename T, bool NEED_WORLD_RPY = true>
__device__ __forceinline__ void world_to_work(const EnvConst<T>& c, V3<T> pos, const M3<T>& R, V3<T>& wpos, T (&wrpy)[3], T (&rpy_world)[3]) {
    Q4<T> q = quat_from_mat(R);
    Q4<T> q2 = q;
    // euler -> quaternion of the euler angles just taken from q gives q back (to rounding, up to the sign, which the product and the second
    // euler conversion do not see) except in getEulerFromQuaternion's gimbal branches, which project: a caller that does not need the world
    // angles themselves skips the six transcendentals of the round trip unless some lane of the wavefront is in such a branch.
    const T sarg = T(-2) * (q.x * q.z - q.w * q.y);
    if (NEED_WORLD_RPY || __any(!(tabs(sarg) < T(0.99999)))) {
        euler_from_quat(q, rpy_world[0], rpy_world[1], rpy_world[2]);
        q2 = quat_from_euler(rpy_world[0], rpy_world[1], rpy_world[2]);
    }
    wpos = load_v3(c.work_inv_pos) + mul(c.work_Rinv, pos);
    const Q4<T> qw = quat_mul(c.work_qinv, q2);
    euler_from_quat(qw, wrpy[0], wrpy[1], wrpy[2]);
}

// np.digitize(v, np.linspace(lo, hi, n)) for increasing bins: the number of bin edges <= v.  Edges are formed exactly like
// numpy's linspace (k * step + lo in two roundings, last edge = hi), hence no FMA contraction here.
__device__ inline int digitize_linspace(double v, double lo, double hi, int n) {
#pragma clang fp contract(off)
    const double step = (hi - lo) / (double)(n - 1);
    // The edges are non-decreasing in k (rounding is monotone), so the count is (largest k with edge_k <= v) + 1.  k0 = floor((v - lo) / step)
    // is that k to within one; the edges below k0 - 1 are then certainly <= v and those above k0 + 2 certainly > v, and the four in between
    // are compared exactly as before (64 comparisons per call otherwise: 2.5 us of surface_follow's k_step).
    double t = floor((v - lo) / step);
    t = t < -1.0 ? -1.0 : (t > (double)n ? (double)n : t);
    int base = (int)t - 1;
    base = base < 0 ? 0 : (base > n ? n : base);
    int count = base;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = base + j;
        const double prod = (double)k * step;
        const double edge = (k == n - 1) ? hi : prod + lo;
        count += (k < n && edge <= v) ? 1 : 0;
    }
    return count;
}
// np.linspace(lo, hi, n)[k], formed like digitize_linspace's edges
__device__ inline double linspace_value(double lo, double hi, int n, int k) {
#pragma clang fp contract(off)
    const double step = (hi - lo) / (double)(n - 1);
    const double prod = (double)k * step;
    return (k == n - 1) ? hi : prod + lo;
}
// np.gradient(f, h) along one axis of a rows x cols array (central differences inside, one-sided at the ends)
__device__ inline double grad_axis(const double* f, int idx, int n, int stride, double h) {
    if (idx == 0) return (f[stride] - f[0]) / h;
    if (idx == n - 1) return (f[(size_t)(n - 1) * stride] - f[(size_t)(n - 2) * stride]) / h;
    return (f[(size_t)(idx + 1) * stride] - f[(size_t)(idx - 1) * stride]) / (2.0 * h);
}

#ifdef TG_KSTEP_STAMPS
__device__ unsigned long long g_kstep_stamps[16];
#define TG_KSTAMP(i) { if (blockIdx.x == 0 && threadIdx.x == 0) g_kstep_stamps[i] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); }
#else
#define TG_KSTAMP(i)
#endif
// Everything that follows the physics of a step or a reset: TCP pose read-back, reward / termination
// (edge_follow_env.py:371-452) and the camera<-stimulus transform handed to the raster (tactile_sensor.py:150-229).
// Episode statistics (the Monitor wrapper the reference's callers always apply: sb3_helpers/rl_utils.py:17-30, 59): the return is the sum, in
// double, of the float32 rewards as handed out - what a Monitor around this env would add up.  done: the finished episode's figures are
// kept for info["episode"] and the running sum starts again.
__device__ __forceinline__ void episode_step(const State& st, int env, float reward, bool done, int step_count) {
    const double acc = st.ep_return[env] + (double)reward;
    if (done) { st.ep_final_return[env] = (float)acc; st.ep_final_len[env] = step_count; }
    st.ep_return[env] = done ? 0.0 : acc;
}

template <typename T, int TOPO>
__device__ __forceinline__ bool finish_env(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const T (&q)[Topo<TOPO>::N],
                                           T edge_ang, int step_count, bool write_reward_done, const JointTrig<T, Topo<TOPO>::N>* trig = nullptr,
                                           bool lazy_rpy = false /* k_step: tcp_rpy is a read-back only, tg_get_state recomputes it (k_refresh_rpy) */) {
    const int n = c.num_envs;
    bool done_flag = false;   // what went into st.done[env], for a caller that resets the env in the same launch (k_step<.., true>)
    Kin<T, TOPO> k;
    if (trig != nullptr) forward_kinematics<T, TOPO, true>(m, q, k, trig);
    else forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    st.tcp_pos[0 * n + env] = (double)ptcp.x; st.tcp_pos[1 * n + env] = (double)ptcp.y; st.tcp_pos[2 * n + env] = (double)ptcp.z;
    if (!lazy_rpy) {
        T rpy[3];
        { Q4<T> qq = quat_from_mat(Rtcp); euler_from_quat(qq, rpy[0], rpy[1], rpy[2]); }
        st.tcp_rpy[0 * n + env] = (double)rpy[0]; st.tcp_rpy[1 * n + env] = (double)rpy[1]; st.tcp_rpy[2 * n + env] = (double)rpy[2];
    }
    T se = T(0), ce = T(1);   // stimulus yaw: edge angle for edge_follow (sin / cos cached by the reset), none for the surface
    if (c.env_kind == TG_ENV_EDGE_FOLLOW) {
        if (lazy_rpy) { se = (T)st.edge_sc[0 * n + env]; ce = (T)st.edge_sc[1 * n + env]; }
        else { tsincos(edge_ang, &se, &ce); st.edge_sc[0 * n + env] = (double)se; st.edge_sc[1 * n + env] = (double)ce; }
    }
    if (write_reward_done && c.env_kind == TG_ENV_EDGE_FOLLOW) {
        const T gx = c.stim_pos[0] + c.edge_len * ce, gy = c.stim_pos[1] + c.edge_len * se;
        const T dx = ptcp.x - gx, dy = ptcp.y - gy;
        const T goal_dist = tsqrt(dx * dx + dy * dy);
        const bool done = goal_dist < c.term_dist || step_count >= c.max_steps;
        // perpendicular distance to the edge centre line: |(p2-p1) x (p1-p3)| / |p2-p1|
        const T p1x = c.stim_pos[0] - c.edge_len * ce, p1y = c.stim_pos[1] - c.edge_len * se;
        const T d21x = gx - p1x, d21y = gy - p1y, d13x = p1x - ptcp.x, d13y = p1y - ptcp.y;
        const T edge_dist = tabs(d21x * d13y - d21y * d13x) / tsqrt(d21x * d21x + d21y * d21y);
        T reward;
        if (c.reward_mode == TG_REWARD_SPARSE) reward = goal_dist < c.term_dist ? T(1) : T(0);
        else reward = -((T(1) * goal_dist) + (T(10) * edge_dist) + T(0));
        st.reward[env] = (float)reward;
        st.done[env] = done ? 1 : 0;
        done_flag = done;
        episode_step(st, env, (float)reward, done, step_count);
    }
    TG_KSTAMP(5)
    if ((write_reward_done || c.reward_mode == TG_REWARD_SPARSE) && c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        // get_step_data / dense_reward (base_surface_env.py:664-684, 703-760; surface_follow_auto_env.py:75-94)
        const int R = c.surf_rows, Cc = c.surf_cols;
        int ti = digitize_linspace((double)ptcp.y, c.ybin_lo, c.ybin_hi, Cc);   // xy_to_surface_idx (:284-300)
        int tj = digitize_linspace((double)ptcp.x, c.xbin_lo, c.xbin_hi, R);
        if (ti == Cc) ti -= 1;
        if (tj == R) tj -= 1;
        const double* H = st.heights + ((size_t)(st.hsel[env] & 3) * n + env) * R * Cc;
        const T surf_z = (T)(H[(size_t)ti * Cc + tj] + (double)c.stim_pos[2]);
        const T gy_ = (T)grad_axis(H + tj, ti, R, Cc, c.surf_scale);               // np.gradient axis 0
        const T gx_ = (T)grad_axis(H + (size_t)ti * Cc, tj, Cc, 1, c.surf_scale);  // axis 1
        V3<T> nrm{-gx_, -gy_, T(1)};
        nrm = (T(1) / norm(nrm)) * nrm;
        const T gdx = ptcp.x - (T)st.goal[0 * n + env], gdy = ptcp.y - (T)st.goal[1 * n + env], gdz = ptcp.z - (T)st.goal[2 * n + env];
        T reward;
        if (c.surf_vertical) {   // the `vertical_simplex` branches (:703-758) on the flipped surface_array / normals (:486-516); vert_env :66-81
            nrm = mul(c.stim_R, nrm);
            const V3<T> tipv = mul(Rtcp, mk(T(-1), T(0), T(0)));
            const V3<T> emb = mul(Rtcp, mk(-c.embed_default, T(0), T(0)));
            const T surf_x = (T)((double)c.stim_pos[0] - H[(size_t)ti * Cc + tj]);   // flipped point: (sx - h, y_bins[i], sz + x_bins[j] - sx)
            const T surf_dist = tabs((ptcp.x + emb.x) - surf_x);
            const T cos_sim = dot(nrm, tipv) / (norm(nrm) * norm(tipv));
            reward = -((T(10) * surf_dist) + (T(3) * (T(1) - cos_sim)));
        } else {
            const V3<T> tipv = mul(Rtcp, mk(T(0), T(0), T(-1)));
            const V3<T> emb = mul(Rtcp, mk(T(0), T(0), -c.embed_default));
            const T surf_dist = tabs((ptcp.z + emb.z) - surf_z);
            const T cos_sim = dot(nrm, tipv) / (norm(nrm) * norm(tipv));
            const T w_norm = (c.movement_mode == TG_SMOVE_YZ || c.movement_mode == TG_SMOVE_XYZ) ? T(0) : T(1);
            reward = c.surf_goal ? -((T(1) * tsqrt(gdx * gdx + gdy * gdy)) + (T(10) * surf_dist) + (w_norm * (T(1) - cos_sim)))   // goal_env :69-90
                                 : -((T(1) * surf_dist) + (w_norm * (T(1) - cos_sim)));
        }
        const bool at_goal = tsqrt(gdx * gdx + gdy * gdy + gdz * gdz) < c.term_dist;
        T out = reward;
        if (c.reward_mode == TG_REWARD_SPARSE) {   // sparse_reward (surface_follow_auto_env.py:59-73): the dense reward is accumulated over the
            const double acc = st.accum[env] + (double)reward;   // episode (from the reset pose on, base_surface_env.py:640) and paid out at the goal
            st.accum[env] = acc;
            out = at_goal ? (T)acc : T(0);
        }
        if (write_reward_done) {
            st.reward[env] = (float)out;
            st.done[env] = (at_goal || step_count >= c.max_steps) ? 1 : 0;
            done_flag = at_goal || step_count >= c.max_steps;
            episode_step(st, env, (float)out, at_goal || step_count >= c.max_steps, step_count);
        }
    }
    if (c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO && st.feature != nullptr) {
        // extended_feature of surface_follow-v1 / -v2 (surface_follow_goal_env.py:92-110, surface_follow_vert_env.py:83-100): TCP position and
        // goal position in the work frame (6 of the 12-wide row); at the end of a step also the copy the auto-reset leaves alone
        const V3<T> tw_ = load_v3(c.work_inv_pos) + mul(c.work_Rinv, ptcp);
        const V3<T> gw_ = load_v3(c.work_inv_pos) + mul(c.work_Rinv, mk((T)st.goal[0 * n + env], (T)st.goal[1 * n + env], (T)st.goal[2 * n + env]));
        const float f[6] = {(float)tw_.x, (float)tw_.y, (float)tw_.z, (float)gw_.x, (float)gw_.y, (float)gw_.z};
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            st.feature[(size_t)env * 12 + e] = f[e];
            if (write_reward_done) st.term_feature[(size_t)env * 12 + e] = f[e];
        }
    }
    TG_KSTAMP(6)
    // camera frame = sensor-body frame o cam offset; eye axes (right, up, -forward) with forward = R[:,0], up = R[:,2]
    V3<T> pb; M3<T> Rb;
    link_frame<T, TOPO>(k, m.sensor_link, m.sensor_pos, m.sensor_rot, pb, Rb);
    const V3<T> pc = pb + mul(Rb, load_v3(c.cam_pos));
    const M3<T> Rc = mul(Rb, c.cam_rot);
    V3<T> f{Rc.m[0], Rc.m[3], Rc.m[6]}, up{Rc.m[2], Rc.m[5], Rc.m[8]};
    f = (T(1) / norm(f)) * f;
    V3<T> s = cross(f, up);
    s = (T(1) / norm(s)) * s;
    const V3<T> u = cross(s, f);
    // object rotation: yaw about z by edge_ang
    V3<T> ox{ce, se, T(0)}, oy{-se, ce, T(0)}, oz{T(0), T(0), T(1)};
    if (c.surf_vertical) {       // upright heightfield: the columns of its fixed rotation
        ox = mk(c.stim_R.m[0], c.stim_R.m[3], c.stim_R.m[6]); oy = mk(c.stim_R.m[1], c.stim_R.m[4], c.stim_R.m[7]);
        oz = mk(c.stim_R.m[2], c.stim_R.m[5], c.stim_R.m[8]);
    }
    const V3<T> dp = load_v3(c.stim_pos) - pc;
    const V3<T> nf = mk<T>(0, 0, 0) - f;
    float* X = st.stim_xform;
    X[0 * n + env] = (float)dot(s, ox);  X[1 * n + env] = (float)dot(s, oy);  X[2 * n + env] = (float)dot(s, oz);
    X[3 * n + env] = (float)dot(u, ox);  X[4 * n + env] = (float)dot(u, oy);  X[5 * n + env] = (float)dot(u, oz);
    X[6 * n + env] = (float)dot(nf, ox); X[7 * n + env] = (float)dot(nf, oy); X[8 * n + env] = (float)dot(nf, oz);
    X[9 * n + env] = (float)dot(s, dp);  X[10 * n + env] = (float)dot(u, dp); X[11 * n + env] = (float)dot(nf, dp);
    return done_flag;
}

// observation_mode "oracle" (get_oracle_obs of every env class) for the whole batch, float32 [n][dim], from the device state:
//   edge_follow    10  TCP pos, TCP lin vel, goal pos (work frame), edge angle                        edge_follow_env.py:454-476
//   surface_follow 20  TCP pos, orn, lin / ang vel, goal pos, surface z under the tip, surface normal  base_surface_env.py:789-819
//   object_balance 26  TCP pos, orn, lin / ang vel; pole pos, orn, lin / ang vel                       object_balance_env.py:528-563
//   object_push    30  TCP pos, rpy, lin / ang vel; cube pos, rpy, lin / ang vel; goal pos, rpy        object_push_env.py:571-609
//   object_roll    34  TCP ...; marble pos, orn, lin / ang vel; goal pos, orn (TCP frame); radius      object_roll_env.py:367-407
// Velocities are Jacobian x qd (getLinkState(computeLinkVelocity = 1)); poses go through the reference's quaternion / euler chain.
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_oracle_obs(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st, int dim,
                                                   float* __restrict__ out) {
    constexpr int N = Topo<TOPO>::N;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int n = c.num_envs, env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    T J[6][N];
    tcp_jacobian<T, TOPO>(m, k, ptcp, J);
    T tw[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < N; ++i) tw[r] += J[r][i] * qd[i];
    // object_roll: the work-frame origin follows the episode's marble radius and embed distance (update_workframe :192-201)
    const T work_dz = c.env_kind == TG_ENV_OBJECT_ROLL ? (T)((2.0 * st.obj_mass[env] - st.embed[env]) - (double)c.work_pos[2]) : T(0);
    V3<T> tp; T trpy[3], rpyw[3];
    world_to_work(c, mk(ptcp.x, ptcp.y, ptcp.z - work_dz), Rtcp, tp, trpy, rpyw);
    const Q4<T> tq = quat_from_euler(trpy[0], trpy[1], trpy[2]);
    const V3<T> tl = mul(c.work_Rinv, mk(tw[0], tw[1], tw[2])), ta = mul(c.work_Rinv, mk(tw[3], tw[4], tw[5]));
    float* o = out + (size_t)env * dim;
    int w = 0;
    auto put3 = [&](V3<T> v) { o[w++] = (float)v.x; o[w++] = (float)v.y; o[w++] = (float)v.z; };
    auto putq = [&](Q4<T> v) { o[w++] = (float)v.x; o[w++] = (float)v.y; o[w++] = (float)v.z; o[w++] = (float)v.w; };
    if (c.env_kind == TG_ENV_EDGE_FOLLOW) {
        const T ang = (T)st.edge_ang[env];
        T se, ce;
        tsincos(ang, &se, &ce);
        const V3<T> goal = mk(c.stim_pos[0] + c.edge_len * ce, c.stim_pos[1] + c.edge_len * se, c.stim_pos[2] + c.edge_height);
        put3(tp); put3(tl); put3(load_v3(c.work_inv_pos) + mul(c.work_Rinv, goal));
        o[w++] = (float)ang;
        return;
    }
    if (c.env_kind == TG_ENV_OBJECT_PUSH) { put3(tp); put3(mk(trpy[0], trpy[1], trpy[2])); }
    else { put3(tp); putq(tq); }
    put3(tl); put3(ta);
    if (c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        const int R = c.surf_rows, Cc = c.surf_cols;
        int ti = digitize_linspace((double)ptcp.y, c.ybin_lo, c.ybin_hi, Cc);   // xy_to_surface_idx (:284-300)
        int tj = digitize_linspace((double)ptcp.x, c.xbin_lo, c.xbin_hi, R);
        if (ti == Cc) ti -= 1;
        if (tj == R) tj -= 1;
        const double* H = st.heights + ((size_t)(st.hsel[env] & 3) * n + env) * R * Cc;
        const T gy_ = (T)grad_axis(H + tj, ti, R, Cc, c.surf_scale), gx_ = (T)grad_axis(H + (size_t)ti * Cc, tj, Cc, 1, c.surf_scale);
        V3<T> nrm{-gx_, -gy_, T(1)};
        nrm = (T(1) / norm(nrm)) * nrm;
        T surf_z = (T)(H[(size_t)ti * Cc + tj] + (double)c.stim_pos[2]);
        if (c.surf_vertical) {   // the flipped surface_array / normals (:486-516)
            nrm = mul(c.stim_R, nrm);
            surf_z = (T)((double)c.stim_pos[2] + (linspace_value(c.xbin_lo, c.xbin_hi, R, tj) - (double)c.stim_pos[0]));
        }
        put3(load_v3(c.work_inv_pos) + mul(c.work_Rinv, mk((T)st.goal[0 * n + env], (T)st.goal[1 * n + env], (T)st.goal[2 * n + env])));
        o[w++] = (float)surf_z;
        put3(mul(c.work_Rinv, nrm));
        return;
    }
    // the free body (get_obj_pos_workframe / get_obj_vel_workframe, base_object_env.py:118-139)
    // (spinning_plate: the env's object is the dish - State::dish -, body_* is the spool on the constraint)
    const bool dish = st.dish != nullptr;
    M3<T> Rb;
#pragma unroll
    for (int e = 0; e < 9; ++e) Rb.m[e] = dish ? (T)st.dish[(3 + e) * n + env] : (T)st.body_rot[e * n + env];
    const V3<T> pb = dish ? mk((T)st.dish[0 * n + env], (T)st.dish[1 * n + env], (T)st.dish[2 * n + env] - work_dz)
                          : mk((T)st.body_pos[0 * n + env], (T)st.body_pos[1 * n + env], (T)st.body_pos[2 * n + env] - work_dz);
    V3<T> op; T orpy[3], orpyw[3];
    world_to_work(c, pb, Rb, op, orpy, orpyw);
    const V3<T> ol = mul(c.work_Rinv, dish ? mk((T)st.dish[12 * n + env], (T)st.dish[13 * n + env], (T)st.dish[14 * n + env])
                                           : mk((T)st.body_v[0 * n + env], (T)st.body_v[1 * n + env], (T)st.body_v[2 * n + env]));
    const V3<T> oa = mul(c.work_Rinv, dish ? mk((T)st.dish[15 * n + env], (T)st.dish[16 * n + env], (T)st.dish[17 * n + env])
                                           : mk((T)st.body_w[0 * n + env], (T)st.body_w[1 * n + env], (T)st.body_w[2 * n + env]));
    if (c.env_kind == TG_ENV_OBJECT_PUSH) {
        put3(op); put3(mk(orpy[0], orpy[1], orpy[2])); put3(ol); put3(oa);
        const int gid = st.goal_id[env], gi = gid < c.traj_n ? gid : c.traj_n - 1;
        o[w++] = (float)st.traj[(0 * TG_MAX_TRAJ_POINTS + gi) * n + env]; o[w++] = (float)st.traj[(1 * TG_MAX_TRAJ_POINTS + gi) * n + env]; o[w++] = 0.0f;
        o[w++] = 0.0f; o[w++] = 0.0f; o[w++] = (float)st.traj[(2 * TG_MAX_TRAJ_POINTS + gi) * n + env];
        return;
    }
    put3(op); putq(quat_from_euler(orpy[0], orpy[1], orpy[2])); put3(ol); put3(oa);
    if (c.env_kind == TG_ENV_OBJECT_ROLL) {
        o[w++] = (float)st.goal[0 * n + env]; o[w++] = (float)st.goal[1 * n + env]; o[w++] = (float)st.goal[2 * n + env];
        o[w++] = 0.0f; o[w++] = 0.0f; o[w++] = 0.0f; o[w++] = 1.0f;
        o[w++] = (float)st.obj_mass[env];
    }
}

// Scene camera (get_visual_obs, base_tactile_env.py:212-245): eye <- frame transforms of one env, rounded once to float:
// frame 0 the world, 1 + i moving link i, N + 1 the task's stimulus / free body (the frame its tactile mesh is expressed in).
struct SceneView { double R[9], t[3]; };   // world -> eye
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_scene_xf(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st, SceneView view,
                                                 const uint8_t* __restrict__ mask, float* __restrict__ xf, float* __restrict__ spheres, int n_spheres) {
    constexpr int N = Topo<TOPO>::N;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int n = c.num_envs, env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n || (mask != nullptr && mask[env] == 0)) return;
    T q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = (T)st.q[i * n + env];
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    float* out = xf + (size_t)env * (N + 2) * 12;
    auto put = [&](int f, const double (&R)[9], const double (&p)[3]) {
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc)
                out[f * 12 + 3 * r + cc] = (float)(view.R[3 * r + 0] * R[cc] + view.R[3 * r + 1] * R[3 + cc] + view.R[3 * r + 2] * R[6 + cc]);
            out[f * 12 + 9 + r] = (float)(view.R[3 * r + 0] * p[0] + view.R[3 * r + 1] * p[1] + view.R[3 * r + 2] * p[2] + view.t[r]);
        }
    };
    {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
        put(0, I, z);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double R[9];
        for (int e = 0; e < 9; ++e) R[e] = (double)k.R[i].m[e];
        const double p[3] = {(double)k.o[i].x, (double)k.o[i].y, (double)k.o[i].z};
        put(1 + i, R, p);
    }
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {(double)c.stim_pos[0], (double)c.stim_pos[1], (double)c.stim_pos[2]};
    if (c.env_kind == TG_ENV_EDGE_FOLLOW) {
        const double se = st.edge_sc[0 * n + env], ce = st.edge_sc[1 * n + env];
        R[0] = ce; R[1] = -se; R[3] = se; R[4] = ce;
    } else if (c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        for (int e = 0; e < 9; ++e) R[e] = (double)c.stim_R.m[e];              // identity, or the upright surface's fixed rotation (-v2)
    } else if (c.env_kind == TG_ENV_OBJECT_BALANCE || c.env_kind == TG_ENV_OBJECT_PUSH || c.env_kind == TG_ENV_OBJECT_ROLL) {
        const double scale = c.env_kind == TG_ENV_OBJECT_ROLL ? st.obj_mass[env] / c.roll_radius : 1.0;   // globalScaling scales the visual
        for (int e = 0; e < 9; ++e) R[e] = st.body_rot[e * n + env] * scale;
        for (int e = 0; e < 3; ++e) p[e] = st.body_pos[e * n + env];
    }
    put(N + 1, R, p);
    // The env's translucent visuals (tg_scene.h: SceneParams::spheres): the goal indicator `sphere_indicator.urdf` (radius 0.01, rgba 1 0 0 0.5)
    // at goal_pos_worldframe (edge_follow_env.py:230-234, 281-283; base_surface_env.py:395-400, 576), object_push's trajectory markers
    // (object_push_env.py:239-250, 281-282 green; 360-366 current target blue, reached ones red), object_roll's goal (its marble URDF, radius
    // 0.0025, painted 1 0 0 0.5: object_roll_env.py:174, 258-284, base_object_env.py:72-75).  Centres: world (double) -> eye, rounded once.
    // Slot 0 is the robot's own translucent visual, loaded before the task's: the TCP marker of every arm URDF (tcp_link: <sphere radius="0.001">,
    // material TransparentRed = rgba 0.9 0 0.2 0.5, e.g. ur5_with_standard_tactip.urdf:25, 335-343) - a tenth of a pixel at these cameras.
    if (spheres == nullptr || n_spheres <= 0) return;
    float* S = spheres + (size_t)env * n_spheres * 8;
    auto put_sphere = [&](int k, const double (&w)[3], float rad, float cr, float cg, float cb) {
        for (int r = 0; r < 3; ++r) S[8 * k + r] = (float)(view.R[3 * r + 0] * w[0] + view.R[3 * r + 1] * w[1] + view.R[3 * r + 2] * w[2] + view.t[r]);
        S[8 * k + 3] = rad; S[8 * k + 4] = cr; S[8 * k + 5] = cg; S[8 * k + 6] = cb; S[8 * k + 7] = 0.5f;
    };
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    {
        const double g[3] = {(double)ptcp.x, (double)ptcp.y, (double)ptcp.z};
        put_sphere(0, g, 0.001f, 229.5f, 0.0f, 51.0f);
    }
    S += 8;                                   // the task's visuals follow
    n_spheres -= 1;
    if (c.env_kind == TG_ENV_EDGE_FOLLOW) {
        const double se = st.edge_sc[0 * n + env], ce = st.edge_sc[1 * n + env];
        const double g[3] = {(double)c.stim_pos[0] + (double)c.edge_len * ce, (double)c.stim_pos[1] + (double)c.edge_len * se, (double)c.stim_pos[2] + (double)c.edge_height};
        put_sphere(0, g, 0.01f, 255.0f, 0.0f, 0.0f);
    } else if (c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        const double g[3] = {st.goal[0 * n + env], st.goal[1 * n + env], st.goal[2 * n + env]};
        put_sphere(0, g, 0.01f, 255.0f, 0.0f, 0.0f);
    } else if (c.env_kind == TG_ENV_OBJECT_PUSH) {
        const int gid = st.goal_id[env];
        for (int i = 0; i < n_spheres; ++i) {
            if (i >= c.traj_n) { S[8 * i + 7] = 0.0f; continue; }
            // workframe_to_worldframe of (x, y, 0) (object_push_env.py:276-281)
            const double wx = st.traj[(0 * TG_MAX_TRAJ_POINTS + i) * n + env], wy = st.traj[(1 * TG_MAX_TRAJ_POINTS + i) * n + env];
            double g[3];
            for (int r = 0; r < 3; ++r) g[r] = (double)c.work_pos[r] + ((double)c.work_R.m[3 * r + 0] * wx + (double)c.work_R.m[3 * r + 1] * wy);
            put_sphere(i, g, 0.01f, i < gid ? 255.0f : 0.0f, i > gid ? 255.0f : 0.0f, i == gid ? 255.0f : 0.0f);
        }
    } else if (c.env_kind == TG_ENV_OBJECT_ROLL) {
        const M3<T> Rq = mat_from_quat(quat_from_mat(Rtcp));                     // multiplyTransforms(tcp pose, goal_pos_tcp) goes through the quaternion
        const V3<T> gw = ptcp + mul(Rq, mk((T)st.goal[0 * n + env], (T)st.goal[1 * n + env], (T)st.goal[2 * n + env]));
        const double g[3] = {(double)gw.x, (double)gw.y, (double)gw.z};
        put_sphere(0, g, 0.0025f, 255.0f, 0.0f, 0.0f);
    } else {
        for (int i = 0; i < n_spheres; ++i) S[8 * i + 7] = 0.0f;
    }
}

// scale_actions (base_tactile_env.py:141-164): clip to [min_action, max_action], affine map to the physical range per dimension
template <typename T> __device__ __forceinline__ void scale_actions(const EnvConst<T>& c, const T (&enc)[6], T (&vels)[6]) {
    const T in_range = c.max_action - c.min_action;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        T x = enc[d];
        x = x < c.min_action ? c.min_action : (x > c.max_action ? c.max_action : x);
        vels[d] = (((x - c.min_action) * (c.act_hi[d] - c.act_lo[d])) / in_range) + c.act_lo[d];
    }
}

// BaseRobotArm.tcp_velocity_control (base_robot_arm.py:281-332): TCP limit check, work -> world twist, Jacobian inverse.
template <typename T, int TOPO>
__device__ __forceinline__ void tcp_velocity_control(const DevRobot<T>& m, const EnvConst<T>& c, const T (&q)[Topo<TOPO>::N], T (&vels)[6],
                                                     T (&qd_des)[Topo<TOPO>::N], const JointTrig<T, Topo<TOPO>::N>* trig = nullptr,
                                                     T work_dz = T(0) /* per-env z offset of the work-frame origin (object_roll) */) {
    constexpr int N = Topo<TOPO>::N;
    Kin<T, TOPO> k;
    if (trig != nullptr) forward_kinematics<T, TOPO, true>(m, q, k, trig);
    else forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    TG_KSTAMP(10)
    V3<T> wpos; T wrpy[3] = {T(0), T(0), T(0)}, rpyw[3];
    // The work-frame orientation of the TCP (matrix -> quaternion -> euler -> quaternion -> multiply -> euler: nine f64 transcendentals)
    // only enters the limit check of the rotational components; a movement mode without rotational velocity (a zero stays a zero in that
    // check) needs the position alone.  Wave-uniform.
    if (__any(vels[3] != T(0) || vels[4] != T(0) || vels[5] != T(0)))
        world_to_work<T, false>(c, mk(ptcp.x, ptcp.y, ptcp.z - work_dz), Rtcp, wpos, wrpy, rpyw);
    else
        wpos = load_v3(c.work_inv_pos) + mul(c.work_Rinv, mk(ptcp.x, ptcp.y, ptcp.z - work_dz));
    const T cur[6] = {wpos.x, wpos.y, wpos.z, wrpy[0], wrpy[1], wrpy[2]};
#pragma unroll
    for (int d = 0; d < 6; ++d) {   // check_TCP_vel_lims (base_robot_arm.py:357-380)
        const bool ex = (cur[d] < c.tcp_lims[d][0] && vels[d] < T(0)) || (cur[d] > c.tcp_lims[d][1] && vels[d] > T(0));
        if (ex) vels[d] = T(0);
    }
    const V3<T> lin = mul(c.work_R, mk(vels[0], vels[1], vels[2]));   // workvel_to_worldvel (:96-105)
    const V3<T> ang = mul(c.work_R, mk(vels[3], vels[4], vels[5]));
    TG_KSTAMP(11)
    T J[6][N];
    tcp_jacobian<T, TOPO>(m, k, ptcp, J);
    TG_KSTAMP(12)
    if (N == 6) {  // square: inverse (reference takes np.linalg.inv when rank is full, :316-319)
        T A[6][6], b[6] = {lin.x, lin.y, lin.z, ang.x, ang.y, ang.z}, x[6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) A[r][cc] = J[r][cc < N ? cc : 0];
        solve_pivoted<T, 6>(A, b, x);
#pragma unroll
        for (int i = 0; i < N; ++i) qd_des[i] = x[i < 6 ? i : 0];
    } else {       // MG400 (mg400.py:77-129): the Jacobian is 6 x 8, always the pseudo-inverse, then the parallel-linkage joints
        const T b[6] = {lin.x, lin.y, lin.z, ang.x, ang.y, ang.z};   // (j2_2, j3_2, j4_2) are slaved to (j2_1, j3_1) by hand (:115-120)
        pinv_apply<T, N>(J, b, qd_des);
        if (N == 8) { qd_des[N - 3] = qd_des[1]; qd_des[N - 2] = -qd_des[1]; qd_des[N - 1] = qd_des[1] + qd_des[2]; }
    }
}

// encode_actions of the arm-only tasks: the policy's dimensions scattered into the 6-vector the controller takes.
template <typename T>
__device__ __forceinline__ void encode_arm_actions(const EnvConst<T>& c, const State& st, int env, const float* __restrict__ a, T (&enc)[6]) {
    const int n = c.num_envs;
    if (c.env_kind == TG_ENV_EDGE_FOLLOW) {               // encode_actions (edge_follow_env.py:345-369)
        enc[0] = (T)a[0]; enc[1] = (T)a[1];
        if (c.movement_mode == TG_MOVE_XYZ) enc[2] = (T)a[2];
        else if (c.movement_mode == TG_MOVE_XYRZ) enc[5] = (T)a[2];
        else if (c.movement_mode == TG_MOVE_XYZRZ) { enc[2] = (T)a[2]; enc[5] = (T)a[3]; }
    } else if (c.surf_goal) {                             // surface_follow_goal_env.py:27-52: every dimension from the agent
        if (c.movement_mode == TG_SMOVE_YZ) { enc[1] = (T)a[0]; enc[2] = (T)a[1]; }
        else if (c.movement_mode == TG_SMOVE_YZRX) { enc[1] = (T)a[0]; enc[2] = (T)a[1]; enc[3] = (T)a[2]; }
        else {
            enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[2] = (T)a[2];
            if (c.movement_mode == TG_SMOVE_XYZRXRY) { enc[3] = (T)a[3]; enc[4] = (T)a[4]; }
        }
    } else if (c.surf_vertical) {                         // surface_follow_vert_env.py:29-48: y is driven toward the goal
        enc[1] = (T)((st.dir[1 * n + env] * (double)c.max_action) * c.auto_scale);
        enc[0] = (T)a[0]; enc[5] = (T)a[1];
    } else {                                              // surface_follow_auto_env.py:27-57: xy are driven toward the goal
        enc[0] = (T)((st.dir[0 * n + env] * (double)c.max_action) * c.auto_scale);
        enc[1] = (T)((st.dir[1 * n + env] * (double)c.max_action) * c.auto_scale);
        enc[2] = (T)a[0];
        if (c.movement_mode == TG_SMOVE_YZRX) enc[3] = (T)a[1];
        else if (c.movement_mode == TG_SMOVE_XYZRXRY) { enc[3] = (T)a[1]; enc[4] = (T)a[2]; }
    }
}

// ------------------------------------------------------------------------------------------------ step kernel
// BaseTactileEnv.step (base_tactile_env.py:166-185): encode + scale the action, tcp_velocity_control
// (base_robot_arm.py:281-332), action_repeat sim ticks (robot.py:182-183), reward / done, render transform.
// One env's step, one lane (k_step: 64 consecutive envs per wavefront; k_step_render in tg_fused.hip: the envs of one render wavefront).
// Returns whether this step ran a full solve (development stamps only).
template <typename T, int TOPO>
__device__ __forceinline__ bool step_env(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const float* __restrict__ actions,
                                         bool* done_out = nullptr /* what the step wrote to st.done[env] */) {
    constexpr int N = Topo<TOPO>::N;
    const int n = c.num_envs;
    TG_KSTAMP(0)
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    if (st.draw != nullptr) {
        // tg_step_random: action_space.sample() for this env inside the step (element i = env * act_dim + j of draw `counter`, the arithmetic of
        // k_sample_actions: the same floats); the workgroup that finishes last moves the counter on
        const uint64_t counter = st.draw[0] + 1, seed = st.draw[1];
        const float lo = (float)c.min_action, hi = (float)c.max_action;
        float abuf[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (j < c.act_dim) {
                const int i = env * c.act_dim + j;
                const uint64_t z = mix64(mix64(seed + kGolden * (counter + 1)) + kGolden * (uint64_t)(i + 1));
                const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
                abuf[j] = lo + (hi - lo) * u;
                st.act_out[i] = abuf[j];
            }
        }
        encode_arm_actions<T>(c, st, env, abuf, enc);
    } else {
        encode_arm_actions<T>(c, st, env, actions + (size_t)env * c.act_dim, enc);
    }
    T vels[6];
    scale_actions<T>(c, enc, vels);
    const int step_count = st.step_count[env] + 1;
    st.step_count[env] = step_count;
    JointTrig<T, N> trig;             // sin/cos of the joint angles, advanced by angle addition through the ticks.  While the licence
    const int lic = st.licence[env];  // below holds they are carried over from the last step (q has not changed in between: a reset or
    if (__all(lic > 0)) {             // tg_set_joint_state drops the licence), i.e. they are evaluated exactly once per 8 steps
#pragma unroll
        for (int i = 0; i < N; ++i) { trig.s[i] = (T)st.trig_sc[i * n + env]; trig.c[i] = (T)st.trig_sc[(8 + i) * n + env]; }
    } else {
        trig_init<T, N>(q, trig);
    }
    T qd_des[N];
    TG_KSTAMP(1)
    tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des, &trig);
    TG_KSTAMP(2)
#pragma unroll
    for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = (double)qd_des[i];

    T qdummy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qdummy[i] = T(0);
    // Licence for sim_tick's analytic fixed point.  A full solve that converged to the last bit within 80 % of the sweep budget arms it
    // for the remaining ticks of this step and for the next 7 steps (<= 0.8 s, < 0.1 rad of joint motion: the Gauss-Seidel contraction
    // is a smooth function of the configuration and the margin is a factor > 2 in sweeps); a reset drops it.  Wave-uniform.
    int verified = __all(lic > 0) ? 24 : 0;
    bool ran_full = false;
    int sweeps = 0;
    for (int t = 0; t < c.action_repeat; ++t) {
        const int before = verified;
        sim_tick<T, TOPO, kMotorVelocity, true, true>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, &trig,
                                                      &verified, &sweeps);
        const bool analytic = before > 0 && verified == before - 1;   // the analytic path decrements; a full solve sets 24 or -1
        if (t == 0) TG_KSTAMP(7)
        if (!analytic) ran_full = true;
        if (verified < 0) verified = 0;
        // Fast-forward.  After an analytic tick qd == des (the velocity motors' constant target), so every remaining tick of this step
        // sees the same inputs to sim_tick's a-priori test (jump 0, damping term of the same qd): if it holds once it holds for all of
        // them and each is just q += dt des - the same additions in the same order, without re-evaluating the test and the angle-addition
        // update 23 more times.  The sines / cosines are re-anchored exactly at the end.  Wave-uniform like the licence itself.
        const int remaining = c.action_repeat - t - 1;
        if (analytic && remaining > 0 && verified >= remaining && c.solver_iters >= 0) {
            T v2 = T(0);
#pragma unroll
            for (int i = 0; i < N; ++i) v2 += qd[i] * qd[i];
            const T lam_star = c.dt * (m.joint_damp + T(4) * (m.lin_damp + m.ang_damp) * m.trace_bound) * T(3) * tsqrt_fast(v2);
            if (__all(T(2.5) * lam_star < m.max_force * c.dt)) {
                T dq[N];
#pragma unroll
                for (int i = 0; i < N; ++i) dq[i] = c.dt * qd[i];
                for (int r = 0; r < remaining; ++r) {
#pragma unroll
                    for (int i = 0; i < N; ++i) q[i] += dq[i];
                }
                T dqt[N];                           // sines / cosines by one angle addition over the whole jump (exact
#pragma unroll                                  // evaluation when it exceeds 0.02 rad); they are re-anchored at every step start
                for (int i = 0; i < N; ++i) dqt[i] = (T)remaining * dq[i];
                trig_advance<T, N>(q, dqt, trig);
                verified -= remaining;
                break;
            }
        }
    }
    st.licence[env] = ran_full ? (verified > 0 ? 8 : 0) : lic - 1;
    if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
    TG_KSTAMP(3)

#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { st.trig_sc[i * n + env] = (double)trig.s[i]; st.trig_sc[(8 + i) * n + env] = (double)trig.c[i]; }
    TG_KSTAMP(4)
    const bool done = finish_env<T, TOPO>(m, c, st, env, q, (T)st.edge_ang[env], step_count, true, &trig, true);
    if (done_out != nullptr) *done_out = done;
    return ran_full;
}
// tg_step_random: every lane of every workgroup has read the draw counter long ago: the last workgroup to get here moves it on
// (st.draw: kDrawWords words - [0] counter, [1] seed, [2] ticket, [16 + 16 g] ticket of workgroup group g, each on a 128-byte line of its own;
//  a launch of many workgroups - k_step_render: one per env - elects in two levels.)  No fence: a workgroup takes its ticket after it has USED
//  the counter value it read (its actions are computed), so the last ticket holder's store cannot reach any of those loads, and the next
//  launch sees the store across the kernel boundary.  An agent-scope fence here is an L2 write-back + invalidate on a multi-XCD part: with one
//  workgroup per env it cost 22 us per launch (profiles/r5_exp_fused_step.txt).
constexpr int kDrawGroups = 32, kDrawWords = 16 + 16 * kDrawGroups;
__device__ __forceinline__ void draw_counter_advance(const State& st) {
    if (st.draw != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            bool last = true;
            if (gridDim.x > 2 * kDrawGroups) {
                const unsigned g = blockIdx.x % kDrawGroups;
                const unsigned long long members = (gridDim.x - g + kDrawGroups - 1) / kDrawGroups;
                unsigned long long* tk = st.draw + 16 + 16 * g;
                last = atomicAdd(tk, 1ull) == members - 1;
                if (last) { atomicExch(tk, 0ull); last = atomicAdd(st.draw + 2, 1ull) == (unsigned long long)kDrawGroups - 1; }
            } else {
                last = atomicAdd(st.draw + 2, 1ull) == (unsigned long long)gridDim.x - 1;
            }
            if (last) { atomicExch(st.draw + 2, 0ull); atomicAdd(st.draw + 0, 1ull); }
        }
    }
}

// The same election in two halves, for a launch of one wavefront per env (k_step_body_wave): lane 0 takes the first-level ticket as soon as its
// draws are computed (`used`: a value derived from the counter it read - the ticket depends on it, so the load has returned) and looks at the
// ticket at the end of the launch, when the atomic's round trip is long over.
__device__ __forceinline__ unsigned long long draw_ticket_take(const State& st, float used) {
    unsigned long long* tk = gridDim.x > 2 * kDrawGroups ? st.draw + 16 + 16 * (blockIdx.x % kDrawGroups) : st.draw + 2;
    return atomicAdd(tk, __float_as_uint(used) == 0x7fc12345u ? 2ull : 1ull);      // (a NaN payload no draw produces: always 1)
}
__device__ __forceinline__ void draw_ticket_resolve(const State& st, unsigned long long ticket) {
    bool last;
    if (gridDim.x > 2 * kDrawGroups) {
        const unsigned g = blockIdx.x % kDrawGroups;
        const unsigned long long members = (gridDim.x - g + kDrawGroups - 1) / kDrawGroups;
        last = ticket == members - 1;
        if (last) { atomicExch(st.draw + 16 + 16 * g, 0ull); last = atomicAdd(st.draw + 2, 1ull) == (unsigned long long)kDrawGroups - 1; }
    } else {
        last = ticket == (unsigned long long)gridDim.x - 1;
    }
    if (last) { atomicExch(st.draw + 2, 0ull); atomicAdd(st.draw + 0, 1ull); }
}

template <typename T, int TOPO> __device__ __forceinline__ void reset_or_swap(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, const State& st,
                                                                             int env, bool in_step, int phase, const BankDev* __restrict__ bd);
// RESET (round 6): the auto-reset of the envs this step finished, inside the step's own launch - reset_or_swap, the body of k_reset, on the
// env's lane straight after its step (reset_phase 0: the whole reset, edge_follow; 1: the task draws / the bank's swap-in, surface_follow,
// whose k_gen_surface + phase-2 launches follow as before).  Same loads, same stores, same order per env as the k_reset launch it replaces
// (an env's reset reads nothing another env's step writes): byte-identical, one dependent launch and its dispatch gap fewer per step.
template <typename T, int TOPO, bool RESET = false>
__global__ __launch_bounds__(64) void k_step(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                             const float* __restrict__ actions, int reset_phase, const BankDev* __restrict__ bd) {
    KtScope kt_scope_(st.kt);
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    TG_TL(st.tl, 1);
    if (env >= cp->num_envs) return;
    bool done = false;
    const bool ran_full = step_env<T, TOPO>(*mp, *cp, st, env, actions, &done);
    if (RESET) {   // (the flag from a register: reading st.done[env] back is a memory round trip at the end of the step's chain, +1 us)
        if (done) reset_or_swap<T, TOPO>(mp, cp, st, env, true, reset_phase, bd);   // (out of line - a noinline wrapper - was measured: k_step 21.7 us, the call changes the whole kernel's register allocation)
    }
    draw_counter_advance(st);
    TG_KSTAMP(9)
#ifdef TG_KSTEP_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_kstep_stamps[15] = ran_full ? 1 : 0;
#else
    (void)ran_full;
#endif
}

// ------------------------------------------------------------------------------------------------ reset kernel
template <typename T> __device__ __forceinline__ V3<T> rot_error(const M3<T>& Rt, const M3<T>& R) {
    // rotation vector of Rt R^T
    M3<T> E;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E.m[3 * i + j] = Rt.m[3 * i] * R.m[3 * j] + Rt.m[3 * i + 1] * R.m[3 * j + 1] + Rt.m[3 * i + 2] * R.m[3 * j + 2];
    const T cosang = T(0.5) * (E.m[0] + E.m[4] + E.m[8] - T(1));
    const V3<T> ax{E.m[7] - E.m[5], E.m[2] - E.m[6], E.m[3] - E.m[1]};
    const T s = norm(ax);                         // = 2 sin(angle)
    const T ang = tatan2(T(0.5) * s, cosang);     // well conditioned for small angles (acos is not)
    if (s < T(1e-12)) return T(0.5) * ax;
    return (ang / s) * ax;
}

// calculateInverseKinematics(TCP link, pos, orn, maxNumIterations, residualThreshold) (base_robot_arm.py:201-209):
// damped least squares on the 6-D pose error, q updated in place from its starting value; returns iterations used.
template <typename T, int TOPO>
__device__ __forceinline__ int inverse_kinematics(const DevRobot<T>& m, V3<T> tpos, const M3<T>& Rt, T (&qik)[Topo<TOPO>::N], int max_iters,
                                                  T threshold) {
    constexpr int N = Topo<TOPO>::N;
    int it = 0;
    for (; it < max_iters; ++it) {
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, qik, k);
        V3<T> p; M3<T> R;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, p, R);
        const V3<T> ep = tpos - p, er = rot_error(Rt, R);
        T e[6] = {ep.x, ep.y, ep.z, er.x, er.y, er.z};
        T res = T(0);
#pragma unroll
        for (int d = 0; d < 6; ++d) res += e[d] * e[d];
        if (tsqrt(res) <= threshold) break;
        T J[6][N];
        tcp_jacobian<T, TOPO>(m, k, p, J);
        T A[6][6], y[6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
                T acc = (r == cc) ? T(1e-8) : T(0);
#pragma unroll
                for (int i = 0; i < N; ++i) acc += J[r][i] * J[cc][i];
                A[r][cc] = acc;
            }
        solve_pivoted<T, 6>(A, e, y);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < 6; ++r) acc += J[r][i] * y[r];
            qik[i] += acc;
        }
    }
    return it;
}

// BaseRobotArm.tcp_position_control (base_robot_arm.py:228-279; MG400 override mg400.py:131-190): target pose = clip(current
// work-frame pose + delta, TCP_lims) (check_TCP_pos_lims, :349-355), back to the world (workframe_to_worldframe, :46-60), inverse
// kinematics from the current joint state.  The POSITION_CONTROL motors then track `qik` (gains pos_gain / vel_gain, max_force).
template <typename T, int TOPO>
__device__ __forceinline__ void tcp_position_target(const DevRobot<T>& m, const EnvConst<T>& c, const T (&q)[Topo<TOPO>::N], const T (&delta)[6],
                                                    V3<T>& tpos, Q4<T>& tq, T (&qik)[Topo<TOPO>::N],
                                                    T work_dz = T(0) /* per-env z offset of the work-frame origin (object_roll) */) {
    constexpr int N = Topo<TOPO>::N;
    {
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> ptcp; M3<T> Rtcp;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
        V3<T> wpos; T wrpy[3], rpyw[3];
        world_to_work<T, false>(c, mk(ptcp.x, ptcp.y, ptcp.z - work_dz), Rtcp, wpos, wrpy, rpyw);
        T tgt[6] = {wpos.x + delta[0], wpos.y + delta[1], wpos.z + delta[2], wrpy[0] + delta[3], wrpy[1] + delta[4], wrpy[2] + delta[5]};
#pragma unroll
        for (int d = 0; d < 6; ++d) tgt[d] = tgt[d] < c.tcp_lims[d][0] ? c.tcp_lims[d][0] : (tgt[d] > c.tcp_lims[d][1] ? c.tcp_lims[d][1] : tgt[d]);
        tpos = load_v3(c.work_pos) + mul(c.work_R, mk(tgt[0], tgt[1], tgt[2]));
        tpos.z += work_dz;
        T trpy[3];
        euler_from_quat(quat_mul(c.work_q, quat_from_euler(tgt[3], tgt[4], tgt[5])), trpy[0], trpy[1], trpy[2]);
        tq = quat_from_euler(trpy[0], trpy[1], trpy[2]);
    }
    const M3<T> Rt = mat_from_quat(tq);
#pragma unroll
    for (int i = 0; i < N; ++i) qik[i] = q[i];
    inverse_kinematics<T, TOPO>(m, tpos, Rt, qik, 100, T(1e-8));
    if (N == 8) { qik[N - 3] = qik[1]; qik[N - 2] = -qik[1]; qik[N - 1] = qik[1] + qik[2]; }   // mg400.py:167-172
}

// blocking_move's exit test (robot.py:216-258), evaluated on the state BEFORE the tick it follows: pose error of the TCP and the
// summed joint speed.
template <typename T, int TOPO>
__device__ __forceinline__ bool pose_reached(const DevRobot<T>& m, const T (&q)[Topo<TOPO>::N], const T (&qd)[Topo<TOPO>::N], const V3<T>& tpos,
                                             const Q4<T>& tq) {
    constexpr int N = Topo<TOPO>::N;
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    V3<T> p; M3<T> R;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, p, R);
    const Q4<T> cq = quat_from_mat(R);
    T total_v = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) total_v += tabs(qd[i]);
    const T pos_err = tabs(tpos.x - p.x) + tabs(tpos.y - p.y) + tabs(tpos.z - p.z);
    const T ip = tq.x * cq.x + tq.y * cq.y + tq.z * cq.z + tq.w * cq.w;
    T ca = T(2) * ip * ip - T(1);
    ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
    return pos_err < T(2e-4) && tacos(ca) < T(1e-3) && total_v < T(0.1);
}

// TCP_position_control step (robot.py:156-186): BaseRobotArm.tcp_position_control (base_robot_arm.py:228-279; MG400 override
// mg400.py:131-190) then Robot.blocking_move(max_steps = _max_blocking_pos_move_steps, constant_vel = None) (robot.py:188-260).
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_step_pos(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                 const float* __restrict__ actions) {
    KtScope kt_scope_(st.kt);
    constexpr int N = Topo<TOPO>::N;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = c.num_envs;
    if (env >= n) return;
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    encode_arm_actions<T>(c, st, env, actions + (size_t)env * c.act_dim, enc);
    T delta[6];
    scale_actions<T>(c, enc, delta);
    const int step_count = st.step_count[env] + 1;
    st.step_count[env] = step_count;

    V3<T> tpos; Q4<T> tq;
    T qik[N], zero[N];
    tcp_position_target<T, TOPO>(m, c, q, delta, tpos, tq, qik);
#pragma unroll
    for (int i = 0; i < N; ++i) { zero[i] = T(0); st.qd_target[i * n + env] = 0.0; }
    int verified = 0, sweeps = 0;
    for (int it = 0; it < c.max_blocking; ++it) {
        const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
        sim_tick<T, TOPO, kMotorPosition>(m, q, qd, qik, zero, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, nullptr, &verified, &sweeps);
        if (verified < 0) verified = 0;
        if (stop) break;
    }
    if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
    st.licence[env] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
    finish_env<T, TOPO>(m, c, st, env, q, (T)st.edge_ang[env], step_count, true);
}

// EdgeFollowEnv.reset (edge_follow_env.py:311-336): reset_task (:285-299), Robot.reset (robot.py:114-125) =
// rest pose + IK to the start pose (base_robot_arm.py:191-226) + blocking_move (robot.py:188-260).
template <typename T, int TOPO>
__device__ __forceinline__ void reset_env(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env,
                                          int phase /*0 all, 1 task draws only, 2 robot only*/) {
    constexpr int N = Topo<TOPO>::N;
    const int n = c.num_envs;

    double embed = (double)c.embed_default, edge_ang = 0.0;
    if (phase != 2) {                                     // reset_task: identical draw order to the reference / oracle
        uint64_t rs = st.rng[env];
        if (c.env_kind == TG_ENV_EDGE_FOLLOW) {           // edge_follow_env.py:285-299, :237-241
            if (c.noise_mode == TG_NOISE_RAND_HEIGHT) embed = rng_uniform(rs, c.embed_lo, c.embed_hi);
            edge_ang = rng_uniform(rs, -3.141592653589793, 3.141592653589793);
        } else {                                          // base_surface_env.py:448 (simplex seed), :520-534 (goal direction)
            st.accum[env] = 0.0;                              // make_goal, base_surface_env.py:589-591
            if (c.noise_mode == TG_SNOISE_SIMPLEX || c.noise_mode == TG_SNOISE_VERTICAL_SIMPLEX) {
                st.noise_seed[env] = (int64_t)rng_uniform(rs, 0.0, 1.0e8);
            } else if (c.noise_mode == TG_SNOISE_RANDOM) {   // gen_heigtfield_noisey draws (rows/2)(cols/2) uniforms: k_gen_surface
                st.noise_seed[env] = (int64_t)rs;            // evaluates them from this state, the stream moves past them here
                rs += (uint64_t)((c.surf_rows / 2) * (c.surf_cols / 2)) * kGolden;
            }
            if (c.movement_mode == TG_SMOVE_YZ || c.movement_mode == TG_SMOVE_YZRX || c.movement_mode == TG_SMOVE_XRZ) {   // np_random.choice([-1, 1])
                st.dir[0 * n + env] = 0.0;
                st.dir[1 * n + env] = rng_uniform(rs, 0.0, 1.0) < 0.5 ? -1.0 : 1.0;
            } else {
                const double ang = rng_uniform(rs, -3.141592653589793, 3.141592653589793);
                st.dir[0 * n + env] = cos(ang);
                st.dir[1 * n + env] = sin(ang);
            }
        }
        st.rng[env] = rs;
        st.embed[env] = embed;
        st.edge_ang[env] = edge_ang;
        st.step_count[env] = 0;
        if (phase == 1) return;
    } else {
        embed = st.embed[env];
        edge_ang = st.edge_ang[env];
    }
    V3<T> init_world = load_v3(c.work_pos) + mul(c.work_R, mk(T(0), T(0), (T)embed));   // edge: work-frame (0, 0, embed)
    if (c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        const int R = c.surf_rows, Cc = c.surf_cols;
        const double* H = st.heights + ((size_t)(st.hsel[env] & 3) * n + env) * R * Cc;
        // make_goal (base_surface_env.py:518-575): goal on the surface, x_y_extent away along the drive direction
        const V3<T> wd = mul(c.work_R, mk((T)st.dir[0 * n + env], (T)st.dir[1 * n + env], T(0)));
        const double gx = (double)c.stim_pos[0] + c.surf_extent * (double)wd.x, gy = (double)c.stim_pos[1] + c.surf_extent * (double)wd.y;
        int gi = digitize_linspace(gy, c.ybin_lo, c.ybin_hi, Cc), gj = digitize_linspace(gx, c.xbin_lo, c.xbin_hi, R);
        if (gi == Cc) gi -= 1;
        if (gj == R) gj -= 1;
        const double hc = H[(size_t)(R / 2) * Cc + (Cc / 2)];
        V3<T> pw;
        if (c.surf_vertical) {   // goal = the flipped surface_array[gi, gj] (:486-498, :547-555); init pose :596-603
            st.goal[0 * n + env] = (double)c.stim_pos[0] - H[(size_t)gi * Cc + gj];
            st.goal[1 * n + env] = linspace_value(c.ybin_lo, c.ybin_hi, Cc, gi);
            st.goal[2 * n + env] = (double)c.stim_pos[2] + (linspace_value(c.xbin_lo, c.xbin_hi, R, gj) - (double)c.stim_pos[0]);
            pw = mk((T)((double)c.stim_pos[0] - (hc - embed)), c.stim_pos[1], c.stim_pos[2]);
        } else {
            st.goal[0 * n + env] = gx; st.goal[1 * n + env] = gy; st.goal[2 * n + env] = H[(size_t)gi * Cc + gj] + (double)c.stim_pos[2];
            // update_init_pose (:590-613): above the surface centre, embed_dist deep, expressed in and back out of the work frame
            pw = mk(c.stim_pos[0], c.stim_pos[1], (T)((double)c.stim_pos[2] + hc - embed));
        }
        const V3<T> pwork = load_v3(c.work_inv_pos) + mul(c.work_Rinv, pw);
        init_world = load_v3(c.work_pos) + mul(c.work_R, pwork);
    }

    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = m.rest_q[i]; qd[i] = T(0); }

    // start pose in the world frame (workframe_to_worldframe, base_robot_arm.py:46-60), rpy 0 in the work frame
    const V3<T> tpos = init_world;
    T trpy[3];
    euler_from_quat(quat_mul(c.work_q, quat_from_euler(T(0), T(0), T(0))), trpy[0], trpy[1], trpy[2]);
    const Q4<T> tq = quat_from_euler(trpy[0], trpy[1], trpy[2]);
    const M3<T> Rt = mat_from_quat(tq);

    // calculateInverseKinematics: damped least squares from the rest pose
    T qik[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qik[i] = q[i];
    inverse_kinematics<T, TOPO>(m, tpos, Rt, qik, 100, T(1e-8));
    if (N == 8) { qik[N - 3] = qik[1]; qik[N - 2] = -qik[1]; qik[N - 1] = qik[1] + qik[2]; }   // mg400.py:222-227 target_joints override

    // blocking_move(max_steps=1000, constant_vel=0.001)
    T cv = T(0.001);
    T zero[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zero[i] = T(0);
    int used = 0, verified = 0;
    for (int it = 0; it < 1000; ++it) {
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> p; M3<T> R;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, p, R);
        const Q4<T> cq = quat_from_mat(R);
        T diff[N], step_j[N], nrm2 = T(0), total_v = T(0);
        bool all_small = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diff[i] = qik[i] - q[i];
            nrm2 += diff[i] * diff[i];
            all_small = all_small && (tabs(diff[i]) < cv);
            total_v += tabs(qd[i]);
        }
        const T nrm = tsqrt(nrm2);
#pragma unroll
        for (int i = 0; i < N; ++i) step_j[i] = q[i] + ((nrm > T(0)) ? diff[i] / nrm : T(0)) * cv;
        if (all_small) cv = cv / T(2);
        sim_tick<T, TOPO, kMotorPosition>(m, q, qd, step_j, zero, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, nullptr, &verified);
        if (verified < 0) verified = 0;
        ++used;
        const T pos_err = tabs(tpos.x - p.x) + tabs(tpos.y - p.y) + tabs(tpos.z - p.z);
        const T ip = tq.x * cq.x + tq.y * cq.y + tq.z * cq.z + tq.w * cq.w;
        T ca = T(2) * ip * ip - T(1);
        ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
        const T orn_err = tacos(ca);
        if (pos_err < T(2e-4) && orn_err < T(1e-3) && total_v < T(0.1)) break;
    }
    st.reset_ticks[env] = used;
    // The licence is renewed inside the reset (round 5).  Dropped here - as until round 5 - it sends the env's whole wavefront through a full
    // solve in the next step, and with episodes that do not end in the same step (any RL run) some wavefront does that in every launch: k_step
    // 49 instead of 17 us (tools/desync_rate.py).  The blocking move's own solves do not qualify as the demonstration - their jumps shrink to
    // nothing, and a solve of nothing converges at once on any arm - so one more tick is solved here, in the step's motor mode, from the
    // reset pose, for a jump with a component along every joint, and thrown away: only its verdict is kept (converged to the last bit within
    // 80 % of the sweep budget or not: the UR5 does, after ~55 of 150 sweeps; the MG400 never does and keeps solving in full).
    int lic = 0;
    if (c.solver_iters > 0 && c.control_mode == TG_CONTROL_TCP_VELOCITY) {
        T qv[N], qdv[N], des[N], qdummy[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { qv[i] = q[i]; qdv[i] = qd[i]; qdummy[i] = T(0); des[i] = qd[i] + T(1e-5) * T((i & 1) ? -(3 + i) : (2 + i)); }   // (small: far from any row's limit, and the exit test is relative)
        int ver = 0;
        sim_tick<T, TOPO, kMotorVelocity, true, false>(m, qv, qdv, qdummy, des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, nullptr, &ver);
        lic = ver > 0 ? 8 : 0;
    }
    st.licence[env] = lic;      // (the sines / cosines a licensed step carries over: the exact ones of the reset pose)
    {
        JointTrig<T, N> trig;
        trig_init<T, N>(q, trig);
#pragma unroll
        for (int i = 0; i < N; ++i) { st.trig_sc[i * n + env] = (double)trig.s[i]; st.trig_sc[(8 + i) * n + env] = (double)trig.c[i]; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
    finish_env<T, TOPO>(m, c, st, env, q, (T)edge_ang, 0, false);
}

// A finished env takes its precomputed post-reset state: everything reset_env writes, copied from the bank view, plus the fields a reset
// sets to constants.  (surface_follow: the 32 KB of heights are not copied - the refill wrote them to the env's spare third, State::hsel.)
template <typename T, int TOPO>
__device__ __forceinline__ void bank_swap_in(const EnvConst<T>& c, const State& st, const State& bk, int env) {
    constexpr int N = Topo<TOPO>::N;
    const int n = c.num_envs;
    // Every load first, then every store: the State pointers may alias as far as the compiler knows, so a copy written field by field is a chain
    // of load - store round trips, one memory latency each (~60 of them: k_reset 15 us in every step of a rollout whose episodes are out of phase).
    const bool surf = c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO, feat = surf && st.feature != nullptr;
    double q[N], qd[N], ts[N], tc[N], pos[3], rpy[3], dir[2] = {0, 0}, goal[3] = {0, 0, 0}, accum = 0;
    float xf[12], ft[6] = {0, 0, 0, 0, 0, 0};
    int64_t nseed = 0;
    uint8_t hs = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = bk.q[i * n + env]; qd[i] = bk.qd[i * n + env]; ts[i] = bk.trig_sc[i * n + env]; tc[i] = bk.trig_sc[(8 + i) * n + env]; }
    const double embed = bk.embed[env], edge_ang = bk.edge_ang[env], esc0 = bk.edge_sc[0 * n + env], esc1 = bk.edge_sc[1 * n + env];
    const int32_t rticks = bk.reset_ticks[env], lic = bk.licence[env];
#pragma unroll
    for (int k = 0; k < 3; ++k) { pos[k] = bk.tcp_pos[k * n + env]; rpy[k] = bk.tcp_rpy[k * n + env]; }
#pragma unroll
    for (int k = 0; k < 12; ++k) xf[k] = bk.stim_xform[k * n + env];
    if (surf) {
        accum = bk.accum[env]; nseed = bk.noise_seed[env]; dir[0] = bk.dir[0 * n + env]; dir[1] = bk.dir[1 * n + env];
#pragma unroll
        for (int k = 0; k < 3; ++k) goal[k] = bk.goal[k * n + env];
        hs = (uint8_t)((bk.hsel[env] & 3) | ((st.hsel[env] & 3) << 2));   // the entry's surface (and its z offset) sit in the third the refill wrote: the env moves there
        if (feat) {
#pragma unroll
            for (int e = 0; e < 6; ++e) ft[e] = bk.feature[(size_t)env * 12 + e];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        st.q[i * n + env] = q[i]; st.qd[i * n + env] = qd[i]; st.qd_target[i * n + env] = 0.0;
        st.trig_sc[i * n + env] = ts[i]; st.trig_sc[(8 + i) * n + env] = tc[i];
    }
    st.embed[env] = embed; st.edge_ang[env] = edge_ang; st.edge_sc[0 * n + env] = esc0; st.edge_sc[1 * n + env] = esc1;
    st.reset_ticks[env] = rticks; st.step_count[env] = 0; st.licence[env] = lic;
#pragma unroll
    for (int k = 0; k < 3; ++k) { st.tcp_pos[k * n + env] = pos[k]; st.tcp_rpy[k * n + env] = rpy[k]; }
#pragma unroll
    for (int k = 0; k < 12; ++k) st.stim_xform[k * n + env] = xf[k];
    if (surf) {
        st.accum[env] = accum; st.noise_seed[env] = nseed; st.dir[0 * n + env] = dir[0]; st.dir[1 * n + env] = dir[1];
#pragma unroll
        for (int k = 0; k < 3; ++k) st.goal[k * n + env] = goal[k];
        st.hsel[env] = hs;
        if (feat) {
#pragma unroll
            for (int e = 0; e < 6; ++e) st.feature[(size_t)env * 12 + e] = ft[e];
        }
    }
}

// bd == nullptr  env.reset() of the masked envs (tg_reset; auto-reset without a bank): reset_env on the main state.
// bd != nullptr  auto-reset inside tg_step with the bank: a finished env whose bank entry belongs to its current RNG state takes it, any other
//                finished env is reset on the spot.  surface_follow (phase 1, k_gen_surface, phase 2): phase 1 leaves swapped[env] / late[env]
//                for the two launches behind it, which clear them again (both masks are all zero between steps).
// (the body of k_reset for one env that is to be reset; in_step: auto-reset inside tg_step)
template <typename T, int TOPO>
__device__ __forceinline__ void reset_or_swap(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, const State& st, int env,
                                              bool in_step, int phase, const BankDev* __restrict__ bd) {
    if (cp->fused_reset && in_step) {           // auto-reset inside tg_step: keep the terminal observation's camera transform; the render
        const int n = cp->num_envs;             // launch that follows draws both images of this env
        float xf[12];                           // (loads, then stores: see bank_swap_in)
#pragma unroll
        for (int k = 0; k < 12; ++k) xf[k] = st.stim_xform[k * n + env];
#pragma unroll
        for (int k = 0; k < 12; ++k) st.term_xform[k * n + env] = xf[k];
    }
    if (bd != nullptr && phase != 2) {
        const BankAux& aux = bd->aux;
        const unsigned long long r = (unsigned long long)st.rng[env];
        const unsigned long long t = __hip_atomic_load(aux.tag + env, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (t == r) {
            bank_swap_in<T, TOPO>(*cp, st, bd->bk, env);
            // the new RNG state is what tells the refill stream that this entry is spent: released after every read of the entry above
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(st.rng) + env, (unsigned long long)bd->bk.rng[env], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (phase == 1) aux.swapped[env] = 1;
            atomicAdd(aux.stats + 0, 1ull);
            return;
        }
        if (phase == 1) aux.late[env] = 1;
        atomicAdd(aux.stats + 1, 1ull);
    }
    // surface_follow, a reset on the spot (phase 1 = the task draws; k_gen_surface, which follows, writes the env's LIVE third): live and last
    // trade places - the finished episode's surface stays readable for this step's terminal image, the new one overwrites the episode before
    // it, and the spare third, which a refill may be writing at this moment, is left alone
    if (phase == 1 && cp->env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        const uint8_t b = st.hsel[env];
        st.hsel[env] = (uint8_t)(((b >> 2) & 3) | ((b & 3) << 2));
    }
    reset_env<T, TOPO>(*mp, *cp, st, env, phase);
    if (bd != nullptr && phase == 2) bd->aux.late[env] = 0;
}
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_reset(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                              const uint8_t* __restrict__ mask, int phase, const BankDev* __restrict__ bd) {
    KtScope kt_scope_(st.kt);
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    TG_TL(st.tl, 2);
    if (env >= cp->num_envs) return;
    if (mask != nullptr && mask[env] == 0) return;
    reset_or_swap<T, TOPO>(mp, cp, st, env, mask != nullptr, phase, bd);
}

// The refill (second stream, outside the step graph): for every env whose bank entry does not belong to its current RNG state, the same
// reset_env on the bank view, started from the main RNG state.  Phases as k_reset's: 0 all in one (edge_follow); 1 the task draws, then
// k_gen_surface on the bank's heights with mask = need, then 2 the robot half (surface_follow).  The last phase publishes the tag.
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_bank_refill(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st, State bk,
                                                    BankAux aux, int phase) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= cp->num_envs) return;
    if (phase != 2) {
        // Relaxed, cache-bypassing read of the RNG state; the acquire (an L2 invalidate on this part: measured, an idle refill with an acquire
        // LOAD doubled the duration of the k_step running beside it, 14.5 -> 31.6 us) is a fence taken only by wavefronts that have work.
        const unsigned long long r = __hip_atomic_load(reinterpret_cast<unsigned long long*>(st.rng) + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool need = aux.tag[env] != r;
        aux.need[env] = need ? 1 : 0;
        if (!need) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // While these launches rewrite the entry its OLD tag must not stay published: an env whose RNG state returns to that value in between
        // (tg_seed with the same seeds does exactly that) would take a half-written entry.  ~r never equals r, and k_reset compares with the
        // env's current state, which is r or something newer - a state that happens to equal ~r is as unlikely as any 64-bit collision.
        __hip_atomic_store(aux.tag + env, ~r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        aux.rng_in[env] = r;
        bk.rng[env] = (uint64_t)r;
        if (cp->env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {   // the spare third - neither live nor the last episode's - takes this entry's surface (the bank view's "live" slot)
            const uint8_t b = st.hsel[env];          // (one byte: a reset on the spot running beside this launch swaps the two fields, the spare stays the spare)
            bk.hsel[env] = (uint8_t)(3 - (b & 3) - ((b >> 2) & 3));
        }
    } else if (aux.need[env] == 0) {
        return;
    }
    reset_env<T, TOPO>(*mp, *cp, bk, env, phase);
    if (phase != 1) __hip_atomic_store(aux.tag + env, aux.rng_in[env], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------ object_balance kernels
template <typename T> __device__ __forceinline__ FreeBody<T> load_body(const State& st, int n, int env) {
    FreeBody<T> b;
    b.pos = mk((T)st.body_pos[0 * n + env], (T)st.body_pos[1 * n + env], (T)st.body_pos[2 * n + env]);
#pragma unroll
    for (int e = 0; e < 9; ++e) b.R.m[e] = (T)st.body_rot[e * n + env];
    b.v = mk((T)st.body_v[0 * n + env], (T)st.body_v[1 * n + env], (T)st.body_v[2 * n + env]);
    b.w = mk((T)st.body_w[0 * n + env], (T)st.body_w[1 * n + env], (T)st.body_w[2 * n + env]);
    return b;
}
template <typename T> __device__ __forceinline__ void store_body(const State& st, int n, int env, const FreeBody<T>& b) {
    st.body_pos[0 * n + env] = (double)b.pos.x; st.body_pos[1 * n + env] = (double)b.pos.y; st.body_pos[2 * n + env] = (double)b.pos.z;
#pragma unroll
    for (int e = 0; e < 9; ++e) st.body_rot[e * n + env] = (double)b.R.m[e];
    st.body_v[0 * n + env] = (double)b.v.x; st.body_v[1 * n + env] = (double)b.v.y; st.body_v[2 * n + env] = (double)b.v.z;
    st.body_w[0 * n + env] = (double)b.w.x; st.body_w[1 * n + env] = (double)b.w.y; st.body_w[2 * n + env] = (double)b.w.z;
}
template <typename T> __device__ __forceinline__ Ball<T> load_ball(const State& st, int n, int env) {
    Ball<T> k;
    k.pos = mk((T)st.ball[0 * n + env], (T)st.ball[1 * n + env], (T)st.ball[2 * n + env]);
    k.v = mk((T)st.ball[3 * n + env], (T)st.ball[4 * n + env], (T)st.ball[5 * n + env]);
    k.w = mk((T)st.ball[6 * n + env], (T)st.ball[7 * n + env], (T)st.ball[8 * n + env]);
    return k;
}
template <typename T> __device__ __forceinline__ void store_ball(const State& st, int n, int env, const Ball<T>& k, T impulse) {
    st.ball[0 * n + env] = (double)k.pos.x; st.ball[1 * n + env] = (double)k.pos.y; st.ball[2 * n + env] = (double)k.pos.z;
    st.ball[3 * n + env] = (double)k.v.x; st.ball[4 * n + env] = (double)k.v.y; st.ball[5 * n + env] = (double)k.v.z;
    st.ball[6 * n + env] = (double)k.w.x; st.ball[7 * n + env] = (double)k.w.y; st.ball[8 * n + env] = (double)k.w.z;
    st.ball[12 * n + env] = (double)impulse;
}
template <typename T> __device__ __forceinline__ T wrap_deg(T d) {   // ((d + 180) % 360) - 180 with numpy's sign-of-divisor modulo
    const T x = d + T(180);
    return (x - T(360) * floor(x / T(360))) - T(180);
}

// spinning_plate: init_obj_pos - setup_object puts the dish on the spool (:215-219), reset_task's version for rand_embed_dist leaves the
// buffer height out (:317-321), as upstream does
template <typename T> __device__ __forceinline__ V3<T> spin_init_obj_pos(const EnvConst<T>& c, T embed) {
    return mk(c.work_pos[0], c.work_pos[1], c.work_pos[2] + (c.rand_embed ? T(0) : c.spin.buffer_height) + (c.obj_base_height / T(2)) - embed);
}
template <typename T> __device__ __forceinline__ FreeBody<T> load_dish(const State& st, int n, int env) {
    FreeBody<T> d;
    d.pos = mk((T)st.dish[0 * n + env], (T)st.dish[1 * n + env], (T)st.dish[2 * n + env]);
#pragma unroll
    for (int e = 0; e < 9; ++e) d.R.m[e] = (T)st.dish[(3 + e) * n + env];
    d.v = mk((T)st.dish[12 * n + env], (T)st.dish[13 * n + env], (T)st.dish[14 * n + env]);
    d.w = mk((T)st.dish[15 * n + env], (T)st.dish[16 * n + env], (T)st.dish[17 * n + env]);
    return d;
}
template <typename T> __device__ __forceinline__ void store_dish(const State& st, int n, int env, const FreeBody<T>& d, T impulse, int contacts) {
    st.dish[0 * n + env] = (double)d.pos.x; st.dish[1 * n + env] = (double)d.pos.y; st.dish[2 * n + env] = (double)d.pos.z;
#pragma unroll
    for (int e = 0; e < 9; ++e) st.dish[(3 + e) * n + env] = (double)d.R.m[e];
    st.dish[12 * n + env] = (double)d.v.x; st.dish[13 * n + env] = (double)d.v.y; st.dish[14 * n + env] = (double)d.v.z;
    st.dish[15 * n + env] = (double)d.w.x; st.dish[16 * n + env] = (double)d.w.y; st.dish[17 * n + env] = (double)d.w.z;
    st.dish[18 * n + env] = (double)impulse; st.dish[19 * n + env] = (double)contacts;
}
// get_step_data / check_obj_fall / termination (object_balance_env.py:426-497) + camera<-object transform
// (the frames of the TCP and of the sensor link at the env's q are the caller's: finish_body below takes them from its own forward kinematics,
//  k_step_body_wave from the lane of its licensed walk that already stands at the step's last q)
//  Returns the env's `done` (false without write_reward_done).
template <typename T, int TOPO>
__device__ __forceinline__ bool finish_body_frames(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const V3<T>& ptcp, const M3<T>& Rtcp,
                                                   const V3<T>& pb, const M3<T>& Rb, const FreeBody<T>& b, T embed, int step_count, bool write_reward_done,
                                                   const FreeBody<T>* obj = nullptr /* spinning_plate: the env's object (the dish); b = what stands on the sensor */) {
    const int n = c.num_envs;
    bool env_done = false;
    T rpy[3];
    { Q4<T> qq = quat_from_mat(Rtcp); euler_from_quat(qq, rpy[0], rpy[1], rpy[2]); }
    st.tcp_pos[0 * n + env] = (double)ptcp.x; st.tcp_pos[1 * n + env] = (double)ptcp.y; st.tcp_pos[2 * n + env] = (double)ptcp.z;
    st.tcp_rpy[0 * n + env] = (double)rpy[0]; st.tcp_rpy[1 * n + env] = (double)rpy[1]; st.tcp_rpy[2 * n + env] = (double)rpy[2];
    if (write_reward_done) {
        const FreeBody<T>& o = obj != nullptr ? *obj : b;
        T orpy[3];
        { Q4<T> qq = quat_from_mat(o.R); euler_from_quat(qq, orpy[0], orpy[1], orpy[2]); }
        const T r2d = T(180) / T(3.141592653589793);
        const T d0 = tabs(wrap_deg(orpy[0] * r2d - c.obj_init_rpy_deg[0])), d1 = tabs(wrap_deg(orpy[1] * r2d - c.obj_init_rpy_deg[1]));
        const V3<T> init = obj != nullptr ? spin_init_obj_pos<T>(c, embed) : mk(c.work_pos[0], c.work_pos[1], c.work_pos[2] + (c.obj_base_height / T(2)) - embed);
        const bool fell = d0 > c.term_deg || d1 > c.term_deg || norm(o.pos - init) > c.term_pos;
        const bool done = fell || step_count >= c.max_steps;
        const T reward = (c.reward_mode == TG_REWARD_SPARSE) ? (fell ? T(-1) : T(0)) : T(1);
        st.reward[env] = (float)reward;
        st.done[env] = done ? 1 : 0;
        env_done = done;
        episode_step(st, env, (float)reward, done, step_count);
    }
    const V3<T> pc = pb + mul(Rb, load_v3(c.cam_pos));
    const M3<T> Rc = mul(Rb, c.cam_rot);
    V3<T> f{Rc.m[0], Rc.m[3], Rc.m[6]}, up{Rc.m[2], Rc.m[5], Rc.m[8]};
    f = (T(1) / norm(f)) * f;
    V3<T> s = cross(f, up);
    s = (T(1) / norm(s)) * s;
    const V3<T> u = cross(s, f);
    const V3<T> ox{b.R.m[0], b.R.m[3], b.R.m[6]}, oy{b.R.m[1], b.R.m[4], b.R.m[7]}, oz{b.R.m[2], b.R.m[5], b.R.m[8]};
    const V3<T> dp = b.pos - pc;
    const V3<T> nf = mk<T>(0, 0, 0) - f;
    const float xv[12] = {(float)dot(s, ox),  (float)dot(s, oy),  (float)dot(s, oz),  (float)dot(u, ox), (float)dot(u, oy), (float)dot(u, oz),
                          (float)dot(nf, ox), (float)dot(nf, oy), (float)dot(nf, oz), (float)dot(s, dp), (float)dot(u, dp), (float)dot(nf, dp)};
#pragma unroll
    for (int i = 0; i < 12; ++i) st.stim_xform[i * n + env] = xv[i];
    // end of a step: a second copy that the reset of the envs that just finished leaves alone, so that the step's observations can be
    // drawn (from this copy) while those envs are being reset on a second stream (enqueue_step)
    if (write_reward_done) {
#pragma unroll
        for (int i = 0; i < 12; ++i) st.term_xform[i * n + env] = xv[i];
    }
    return env_done;
}

template <typename T> struct LinkFrames { V3<T> ptcp, pb; M3<T> Rtcp, Rb; };   // the TCP's and the sensor link's frames at some q
template <typename T, int TOPO>
__device__ __forceinline__ void finish_body(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const T (&q)[Topo<TOPO>::N],
                                            const FreeBody<T>& b, T embed, int step_count, bool write_reward_done) {
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp, pb; M3<T> Rtcp, Rb;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    link_frame<T, TOPO>(k, m.sensor_link, m.sensor_pos, m.sensor_rot, pb, Rb);
    (void)finish_body_frames<T, TOPO>(m, c, st, env, ptcp, Rtcp, pb, Rb, b, embed, step_count, write_reward_done);
}

template <typename T, int TOPO, bool POS, bool BALL = false>
__global__ __launch_bounds__(64) void k_step_body(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                  const float* __restrict__ actions) {
    // (no KtScope here: with it the ball_on_plate instantiation - 134 spilled VGPRs, > 1000 spilled SGPRs - came out of hipcc 7.2 producing NaNs in
    //  random envs, tests/test_gpu_config_scale.py::test_long_horizon_ball_on_plate_matches_oracle; its duration is taken from HIP events.
    //  -DTG_KT_BODY puts it back: the A/B builds of profiles/r6_nan_audit.txt)
#ifdef TG_KT_BODY
    KtScope kt_scope_(st.kt);
#endif
    constexpr int N = Topo<TOPO>::N;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = c.num_envs;
    if (env >= n) return;
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    FreeBody<T> b = load_body<T>(st, n, env);
    T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};   // encode_actions (object_balance_env.py:398-424)
    const float* a = actions + (size_t)env * c.act_dim;
    if (c.movement_mode == TG_BMOVE_XY) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; }
    else if (c.movement_mode == TG_BMOVE_XYZ) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[2] = (T)a[2]; }
    else if (c.movement_mode == TG_BMOVE_RXRY) { enc[3] = (T)a[0]; enc[4] = (T)a[1]; }
    else { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[3] = (T)a[2]; enc[4] = (T)a[3]; }
    T vels[6];
    scale_actions<T>(c, enc, vels);
    const int step_count = st.step_count[env] + 1;
    st.step_count[env] = step_count;
    T qd_des[N];
    V3<T> tpos; Q4<T> tq;
    if constexpr (POS) {                                  // TCP_position_control: qd_des carries the joint targets
        tcp_position_target<T, TOPO>(m, c, q, vels, tpos, tq, qd_des);
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = 0.0;
    } else {
        tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des);
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = (double)qd_des[i];
    }
    const T embed = (T)st.embed[env];
    const V3<T> grav = mk(T(0), T(0), (T)st.gravity[env]);
    const V3<T> pivot_b = mk(T(0), T(0), -c.obj_base_height / T(2) + embed);
    const V3<T> fext = load_v3(c.ext_force);
    const V3<T> pext = mk((T)st.ext_pos[0 * n + env], (T)st.ext_pos[1 * n + env], (T)st.ext_pos[2 * n + env]);
    const bool pending = st.ext_pending[env] != 0;
    T qdummy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qdummy[i] = T(0);
    int verified = 0, sweeps = 0;
    if constexpr (BALL) {                                 // ball_on_plate: the plate + ball tick (no plate force; the pending one-shot is the ball's torque)
        Ball<T> ball = load_ball<T>(st, n, env);
        const V3<T> btq = mk((T)st.ball[9 * n + env], (T)st.ball[10 * n + env], (T)st.ball[11 * n + env]);
        T imp = T(0);
        if constexpr (POS) {
            for (int t = 0; t < c.max_blocking; ++t) {
                const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
                sim_tick_body_ball<T, TOPO, kMotorPosition>(m, q, qd, qd_des, qdummy, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, grav,
                                                            b, c.body, pivot_b, ball, c.ball, btq, pending && t == 0, imp, &sweeps);
                if (stop) break;
            }
        } else {
            for (int t = 0; t < c.action_repeat; ++t)
                sim_tick_body_ball<T, TOPO, kMotorVelocity>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, b,
                                                            c.body, pivot_b, ball, c.ball, btq, pending && t == 0, imp, &sweeps);
        }
        store_ball<T>(st, n, env, ball, imp);
    } else if constexpr (POS) {                           // blocking_move(max_steps, constant_vel=None), robot.py:188-260
        for (int t = 0; t < c.max_blocking; ++t) {
            const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
            sim_tick_body<T, TOPO, kMotorPosition>(m, q, qd, qd_des, qdummy, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, b,
                                                   c.body, pivot_b, fext, pext, pending && t == 0, &verified, nullptr, &sweeps);
            if (stop) break;
        }
    } else {
        JointTrig<T, N> trig;              // sines / cosines of the joint angles, advanced by angle addition through the analytic ticks
        trig_init<T, N>(q, trig);
        for (int t = 0; t < c.action_repeat; ++t)
            sim_tick_body<T, TOPO, kMotorVelocity>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, b, c.body,
                                                   pivot_b, fext, pext, pending && t == 0, &verified, &trig, &sweeps);
    }
    if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
    st.ext_pending[env] = 0;
#ifdef TG_DEBUG_FINITE   // debug builds: a non-finite state stops the kernel where it appears instead of surfacing as a parity failure later
    {
        T chk = b.pos.x + b.pos.y + b.pos.z + b.v.x + b.v.y + b.v.z + b.w.x + b.w.y + b.w.z;
#pragma unroll
        for (int i = 0; i < N; ++i) chk += q[i] + qd[i];
#pragma unroll
        for (int e = 0; e < 9; ++e) chk += b.R.m[e];
        if (!(chk - chk == T(0))) __builtin_trap();
    }
#endif
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
    store_body<T>(st, n, env, b);
    finish_body<T, TOPO>(m, c, st, env, q, b, embed, step_count, true);
}

// BaseObjectEnv.reset (base_object_env.py:146-173) for object_balance: reset_task (gravity, embed), Robot.reset with the pole
// still tied to the TCP, reset_object (teleport + one-shot random force).
// (the reset of ONE env; k_reset_body below is its lane-per-env launch, k_step_body_wave calls the FAST form in its epilogue)
template <typename T, int TOPO, bool BALL = false, bool FAST = false /* the template is known to be valid: no inverse kinematics / blocking move in the binary */,
          bool SPIN = false /* spinning_plate: `b` is the spool (the arm moves back with it on the constraint; the dish lies where it fell and is not part
                               of that move - the position motors prescribe the arm's velocity whatever hangs on it, see below) */>
__device__ __forceinline__ void reset_body_env(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env,
                                               const LinkFrames<T>* tmpl_frames = nullptr /* FAST: the frames at the template's q, if the caller has them */) {
    constexpr int N = Topo<TOPO>::N;
    const int n = c.num_envs;
    uint64_t rs = st.rng[env];
    const double gz = c.rand_gravity ? rng_uniform(rs, c.gravity_lo, c.gravity_hi) : c.gravity_default;   // reset_task :301-306
    double embed = st.embed[env];
    if (c.rand_embed) embed = rng_uniform(rs, c.embed_lo, c.embed_hi);                                    // :308-316
    st.gravity[env] = gz;
    st.embed[env] = embed;
    st.step_count[env] = 0;
    const V3<T> grav = mk(T(0), T(0), (T)gz);
    const V3<T> pivot_b = SPIN ? mk(T(0), T(0), -c.spin.buffer_height / T(2) + c.spin.embed0) : mk(T(0), T(0), -c.obj_base_height / T(2) + (T)embed);
    FreeBody<T> b = load_body<T>(st, n, env);
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = m.rest_q[i]; qd[i] = T(0); }
    // Robot.reset: IK to the work-frame origin, rpy 0 (update_init_pose, base_object_env.py:96-103)
    const V3<T> tpos = load_v3(c.work_pos);
    T trpy[3];
    euler_from_quat(quat_mul(c.work_q, quat_from_euler(T(0), T(0), T(0))), trpy[0], trpy[1], trpy[2]);
    const Q4<T> tq = quat_from_euler(trpy[0], trpy[1], trpy[2]);
    const M3<T> Rt = mat_from_quat(tq);
    // Robot.reset starts from the rest pose and drives to a constant target with position motors of 1e5 N m: the motor rows prescribe the
    // arm's velocity whatever hangs on the constraint (sim_tick_body: qd = des exactly on the analytic ticks, to the last bit of the
    // converged Gauss-Seidel on the full ones), and the object is teleported afterwards (reset_object).  So the arm's state after the
    // blocking move and the number of ticks it took are the same for every reset of every env: they are computed once - by env 0 in the
    // first reset that includes it - and taken from `reset_tmpl` from then on (k_reset_body 0.28 ms -> a few us per launch; the difference
    // to recomputing with the fallen object attached is the last-bit residue of the full ticks, tests/test_gpu_reset_bank.py).
    // tg_config.reset_bank = TG_BANK_OFF (or TG_RESET_BANK=0) recomputes every time.
    // Only the FAST instantiation reads the template, and the host launches it only after a launch that wrote it has been enqueued
    // (tg_ctx::tmpl_ready): the launch that computes the template never also consumes it, so which envs recompute does not depend on how
    // that launch's workgroups happen to be scheduled.
    constexpr bool use_tmpl = FAST;
    const V3<T> z3 = mk<T>(0, 0, 0);
    int used = 0, verified = 0;
    Ball<T> ball;                                         // ball_on_plate: the ball lies where the last episode left it while the arm moves back
    T imp = T(0);
    if constexpr (BALL) ball = load_ball<T>(st, n, env);
    if (use_tmpl) {
#pragma unroll
        for (int i = 0; i < N; ++i) { q[i] = (T)st.reset_tmpl[i]; qd[i] = (T)st.reset_tmpl[N + i]; }
        used = (int)st.reset_tmpl[2 * N];
    } else if constexpr (!FAST) {
    T qik[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qik[i] = q[i];
    inverse_kinematics<T, TOPO>(m, tpos, Rt, qik, 100, T(1e-8));
    T cv = T(0.001);
    T zero[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zero[i] = T(0);
    for (int it = 0; it < 1000; ++it) {
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> p; M3<T> R;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, p, R);
        const Q4<T> cq = quat_from_mat(R);
        T diff[N], step_j[N], nrm2 = T(0), total_v = T(0);
        bool all_small = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diff[i] = qik[i] - q[i];
            nrm2 += diff[i] * diff[i];
            all_small = all_small && (tabs(diff[i]) < cv);
            total_v += tabs(qd[i]);
        }
        const T nrm = tsqrt(nrm2);
#pragma unroll
        for (int i = 0; i < N; ++i) step_j[i] = q[i] + ((nrm > T(0)) ? diff[i] / nrm : T(0)) * cv;
        if (all_small) cv = cv / T(2);
        if constexpr (BALL)
            sim_tick_body_ball<T, TOPO, kMotorPosition>(m, q, qd, step_j, zero, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, grav, b,
                                                        c.body, pivot_b, ball, c.ball, z3, false, imp);
        else
            sim_tick_body<T, TOPO, kMotorPosition>(m, q, qd, step_j, zero, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, grav, b, c.body,
                                                   pivot_b, z3, z3, false, &verified);
        ++used;
        const T pos_err = tabs(tpos.x - p.x) + tabs(tpos.y - p.y) + tabs(tpos.z - p.z);
        const T ip = tq.x * cq.x + tq.y * cq.y + tq.z * cq.z + tq.w * cq.w;
        T ca = T(2) * ip * ip - T(1);
        ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
        if (pos_err < T(2e-4) && tacos(ca) < T(1e-3) && total_v < T(0.1)) break;
    }
    if (st.reset_tmpl != nullptr && env == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) { st.reset_tmpl[i] = (double)q[i]; st.reset_tmpl[N + i] = (double)qd[i]; }
        st.reset_tmpl[2 * N] = (double)used;
        __threadfence();
        st.reset_tmpl[2 * N + 1] = 1.0;
    }
    }
    st.reset_ticks[env] = used;
    st.licence[env] = 0;   // a new configuration: the next step verifies its solve again (k_step_body_wave)
    if constexpr (SPIN) {
        // reset_object (:330-358): the dish back on init_obj_pos, the spool on init_buffer_pos (:228-233, reset_plate_buffer), a new manifold, the
        // one-tick torque (apply_random_torque_obj: no draw) and force (apply_random_force_base: four draws, about the DISH's init position)
        FreeBody<T> d;
        d.pos = spin_init_obj_pos<T>(c, (T)embed);
        d.R = c.obj_init_rot;
        d.v = z3; d.w = z3;
        b.pos = mk(c.work_pos[0], c.work_pos[1], c.work_pos[2] + c.spin.buffer_height / T(2));
#pragma unroll
        for (int e = 0; e < 9; ++e) b.R.m[e] = (e % 4 == 0) ? T(1) : T(0);
        b.v = z3; b.w = z3;
        st.mani[(size_t)36 * n + env] = 0.0;
        const double sx = rng_uniform(rs, 0.0, 1.0) < 0.5 ? -1.0 : 1.0;
        const double rx = rng_uniform(rs, 0.0, 1.0);
        const double sy = rng_uniform(rs, 0.0, 1.0) < 0.5 ? -1.0 : 1.0;
        const double ry = rng_uniform(rs, 0.0, 1.0);
        st.rng[env] = rs;
        st.ext_pos[0 * n + env] = (double)d.pos.x + sx * rx * (double)c.obj_base_width / 2.0;
        st.ext_pos[1 * n + env] = (double)d.pos.y + sy * ry * (double)c.obj_base_width / 2.0;
        st.ext_pos[2 * n + env] = (double)d.pos.z;
        st.ext_pending[env] = 1;
#pragma unroll
        for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
        store_body<T>(st, n, env, b);
        store_dish<T>(st, n, env, d, T(0), 0);
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> ptcp, pbs; M3<T> Rtcp, Rbs;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
        link_frame<T, TOPO>(k, m.sensor_link, m.sensor_pos, m.sensor_rot, pbs, Rbs);
        (void)finish_body_frames<T, TOPO>(m, c, st, env, ptcp, Rtcp, pbs, Rbs, b, (T)embed, 0, false, &d);
        return;
    }
    // reset_object (object_balance_env.py:330-381): teleport, then a one-shot downward force at a random point of the base plate
    b.pos = mk(c.work_pos[0], c.work_pos[1], c.work_pos[2] + (c.obj_base_height / T(2)) - (T)embed);
    b.R = c.obj_init_rot;
    b.v = z3; b.w = z3;
    if constexpr (BALL) {   // reset_ball + apply_random_torque_ball(0.001) (:325-326, 350-353, 393-401); LINK_FRAME of a ball just reset = world
        ball.pos = mk(c.work_pos[0], c.work_pos[1], c.work_pos[2] + c.ball.radius);
        ball.v = z3; ball.w = z3;
        const double u1 = rng_uniform(rs, -1.0, 1.0);
        const double u2 = rng_uniform(rs, -1.0, 1.0);
        st.rng[env] = rs;
        st.ball[9 * n + env] = u1 * 0.001; st.ball[10 * n + env] = u2 * 0.001; st.ball[11 * n + env] = 0.0;
        st.ext_pending[env] = 1;
        store_ball<T>(st, n, env, ball, T(0));
#pragma unroll
        for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
        store_body<T>(st, n, env, b);
        finish_body<T, TOPO>(m, c, st, env, q, b, (T)embed, 0, false);
        return;
    }
    const double sx = rng_uniform(rs, 0.0, 1.0) < 0.5 ? -1.0 : 1.0;
    const double rx = rng_uniform(rs, 0.0, 1.0);
    const double sy = rng_uniform(rs, 0.0, 1.0) < 0.5 ? -1.0 : 1.0;
    const double ry = rng_uniform(rs, 0.0, 1.0);
    st.rng[env] = rs;
    st.ext_pos[0 * n + env] = (double)b.pos.x + sx * rx * (double)c.obj_base_width / 2.0;
    st.ext_pos[1 * n + env] = (double)b.pos.y + sy * ry * (double)c.obj_base_width / 2.0;
    st.ext_pos[2 * n + env] = (double)b.pos.z;
    st.ext_pending[env] = 1;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
    store_body<T>(st, n, env, b);
    if (FAST && tmpl_frames != nullptr)
        (void)finish_body_frames<T, TOPO>(m, c, st, env, tmpl_frames->ptcp, tmpl_frames->Rtcp, tmpl_frames->pb, tmpl_frames->Rb, b, (T)embed, 0, false);
    else
        finish_body<T, TOPO>(m, c, st, env, q, b, (T)embed, 0, false);
}
template <typename T, int TOPO, bool BALL = false, bool FAST = false, bool SPIN = false>
__global__ __launch_bounds__(64) void k_reset_body(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                   const uint8_t* __restrict__ mask) {
    // (no KtScope here: with it the ball_on_plate instantiation - 134 spilled VGPRs, > 1000 spilled SGPRs - came out of hipcc 7.2 producing NaNs in
    //  random envs, tests/test_gpu_config_scale.py::test_long_horizon_ball_on_plate_matches_oracle; its duration is taken from HIP events)
#ifdef TG_KT_BODY
    KtScope kt_scope_(st.kt);
#endif
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= cp->num_envs) return;
    if (mask != nullptr && mask[env] == 0) return;
    reset_body_env<T, TOPO, BALL, FAST, SPIN>(*mp, *cp, st, env);
}

// ------------------------------------------------------------------------------------------------ object_push kernels
// get_step_data (object_push_env.py:456-569): reward, goal advance / termination, extended_feature (:611-629), camera<-cube.
template <typename T, int TOPO>
__device__ __forceinline__ void finish_push(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const T (&q)[Topo<TOPO>::N],
                                            const FreeBody<T>& b, int step_count, bool write_reward_done) {
    const int n = c.num_envs;
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    const Q4<T> qtcp = quat_from_mat(Rtcp);
    V3<T> wpos; T wrpy[3], rpy[3];
    world_to_work(c, ptcp, Rtcp, wpos, wrpy, rpy);
    st.tcp_pos[0 * n + env] = (double)ptcp.x; st.tcp_pos[1 * n + env] = (double)ptcp.y; st.tcp_pos[2 * n + env] = (double)ptcp.z;
    st.tcp_rpy[0 * n + env] = (double)rpy[0]; st.tcp_rpy[1 * n + env] = (double)rpy[1]; st.tcp_rpy[2 * n + env] = (double)rpy[2];
    int gid = st.goal_id[env];
    bool done = false;
    if (write_reward_done) {
        const int gi = gid < c.traj_n ? gid : c.traj_n - 1;
        const V3<T> gw = mk((T)st.traj[(0 * TG_MAX_TRAJ_POINTS + gi) * n + env], (T)st.traj[(1 * TG_MAX_TRAJ_POINTS + gi) * n + env], T(0));
        const T gyaw = (T)st.traj[(2 * TG_MAX_TRAJ_POINTS + gi) * n + env];
        const V3<T> gpos = load_v3(c.work_pos) + mul(c.work_R, gw);                       // workframe_to_worldframe (:305-313)
        T grpy[3];
        euler_from_quat(quat_mul(c.work_q, quat_from_euler(T(0), T(0), gyaw)), grpy[0], grpy[1], grpy[2]);
        const Q4<T> gq = quat_from_euler(grpy[0], grpy[1], grpy[2]);
        const Q4<T> oq = quat_from_mat(b.R);
        const T pos_dist = norm(b.pos - gpos);
        T reward;
        if (c.reward_mode == TG_REWARD_SPARSE) reward = pos_dist < c.term_dist ? T(1) : T(0);
        else {
            const T ip = gq.x * oq.x + gq.y * oq.y + gq.z * oq.z + gq.w * oq.w;
            T ca = T(2) * (ip * ip) - T(1);
            ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
            const T orn_dist = tacos(ca);
            const M3<T> Rq = mat_from_quat(qtcp);
            const V3<T> ov = mk(b.R.m[0], b.R.m[3], b.R.m[6]), tv = mk(Rq.m[0], Rq.m[3], Rq.m[6]);
            const T cos_dist = T(1) - dot(ov, tv) / (norm(ov) * norm(tv));
            reward = -((T(1) * pos_dist) + (T(1) * orn_dist) + (T(1) * cos_dist));
        }
        if (pos_dist < c.term_dist) {                                                     // termination (:520-537), update_goal (:342-370)
            gid += 1;
            if (gid >= c.traj_n) done = true;
            st.goal_id[env] = gid;
        }
        if (step_count >= c.max_steps) done = true;
        st.reward[env] = (float)reward;
        st.done[env] = done ? 1 : 0;
        episode_step(st, env, (float)reward, done, step_count);
    }
    {   // extended_feature: TCP pose and current goal pose in the work frame
        const int gi = gid < c.traj_n ? gid : c.traj_n - 1;
        float f[12] = {(float)wpos.x, (float)wpos.y, (float)wpos.z, (float)wrpy[0], (float)wrpy[1], (float)wrpy[2],
                       (float)st.traj[(0 * TG_MAX_TRAJ_POINTS + gi) * n + env], (float)st.traj[(1 * TG_MAX_TRAJ_POINTS + gi) * n + env], 0.0f,
                       0.0f, 0.0f, (float)st.traj[(2 * TG_MAX_TRAJ_POINTS + gi) * n + env]};
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            if (write_reward_done) st.term_feature[(size_t)env * 12 + e] = f[e];
            st.feature[(size_t)env * 12 + e] = f[e];
        }
    }
    V3<T> pb; M3<T> Rb;
    link_frame<T, TOPO>(k, m.sensor_link, m.sensor_pos, m.sensor_rot, pb, Rb);
    const V3<T> pc = pb + mul(Rb, load_v3(c.cam_pos));
    const M3<T> Rc = mul(Rb, c.cam_rot);
    V3<T> f{Rc.m[0], Rc.m[3], Rc.m[6]}, up{Rc.m[2], Rc.m[5], Rc.m[8]};
    f = (T(1) / norm(f)) * f;
    V3<T> s = cross(f, up);
    s = (T(1) / norm(s)) * s;
    const V3<T> u = cross(s, f);
    const V3<T> ox{b.R.m[0], b.R.m[3], b.R.m[6]}, oy{b.R.m[1], b.R.m[4], b.R.m[7]}, oz{b.R.m[2], b.R.m[5], b.R.m[8]};
    const V3<T> dp = b.pos - pc;
    const V3<T> nf = mk<T>(0, 0, 0) - f;
    float* X = st.stim_xform;
    X[0 * n + env] = (float)dot(s, ox);  X[1 * n + env] = (float)dot(s, oy);  X[2 * n + env] = (float)dot(s, oz);
    X[3 * n + env] = (float)dot(u, ox);  X[4 * n + env] = (float)dot(u, oy);  X[5 * n + env] = (float)dot(u, oz);
    X[6 * n + env] = (float)dot(nf, ox); X[7 * n + env] = (float)dot(nf, oy); X[8 * n + env] = (float)dot(nf, oz);
    X[9 * n + env] = (float)dot(s, dp);  X[10 * n + env] = (float)dot(u, dp); X[11 * n + env] = (float)dot(nf, dp);
}

template <typename T, int TOPO, bool POS>
__global__ __launch_bounds__(64) void k_step_push(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                  const float* __restrict__ actions) {
    // (no KtScope: these kernels run for milliseconds and live at the edge of the register file - the scope costs them up to 1.2 KB more scratch)
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double push_lds_raw[];             // kPushLdsWords * 64 words of T (83 KB in f64: dynamic, above the 64 KB static cap)
    const lds_ptr<T> lds = (lds_ptr<T>)push_lds_raw;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = c.num_envs;
    if (env >= n) return;
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    FreeBody<T> b = load_body<T>(st, n, env);
    T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};   // encode_actions (object_push_env.py:372-454)
    const float* a = actions + (size_t)env * c.act_dim;
    if (c.movement_mode == TG_PMOVE_Y) { enc[0] = c.max_action; enc[1] = (T)a[0]; }
    else if (c.movement_mode == TG_PMOVE_YRZ) { enc[0] = c.max_action; enc[1] = (T)a[0]; enc[5] = (T)a[1]; }
    else if (c.movement_mode == TG_PMOVE_XYRZ) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[5] = (T)a[2]; }
    else {                                               // TCP-frame moves: along / across the sensor's pointing direction
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> ptcp; M3<T> Rtcp;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
        const M3<T> Rq = mat_from_quat(quat_from_mat(Rtcp));
        const V3<T> par = mul(c.work_Rinv, mul(Rq, mk(T(1), T(0), T(0)))), perp = mul(c.work_Rinv, mul(Rq, mk(T(0), T(-1), T(0))));
        if (c.movement_mode == TG_PMOVE_TYRZ) {
            const T pa_ = T(1) * c.max_action;
            enc[0] += perp.x * (T)a[0] + par.x * pa_;
            enc[1] += perp.y * (T)a[0] + par.y * pa_;
            enc[5] += (T)a[1];
        } else {
            enc[0] += perp.x * (T)a[1] + par.x * (T)a[0];
            enc[1] += perp.y * (T)a[1] + par.y * (T)a[0];
            enc[5] += (T)a[2];
        }
    }
    T vels[6];
    scale_actions<T>(c, enc, vels);
    const int step_count = st.step_count[env] + 1;
    st.step_count[env] = step_count;
    T qd_des[N];
    V3<T> tpos; Q4<T> tq;
    if constexpr (POS) {                                  // TCP_position_control: qd_des carries the joint targets
        tcp_position_target<T, TOPO>(m, c, q, vels, tpos, tq, qd_des);
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = 0.0;
    } else {
        tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des);
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = (double)qd_des[i];
    }
    const T mass = (T)st.obj_mass[env];
    int ccode = 0, sweeps = 0;
    T qdummy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qdummy[i] = T(0);
    if constexpr (POS) {                                  // blocking_move(max_steps, constant_vel=None), robot.py:188-260
        for (int t = 0; t < c.max_blocking; ++t) {
            const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
            sim_tick_push<T, TOPO, kMotorPosition>(m, q, qd, qd_des, qdummy, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, b, c.push,
                                                   (const T*)st.tip_verts, mass, lds + threadIdx.x, ccode, &sweeps);
            if (stop) break;
        }
    } else {
        for (int t = 0; t < c.action_repeat; ++t)
            sim_tick_push<T, TOPO, kMotorVelocity>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, b, c.push,
                                                   (const T*)st.tip_verts, mass, lds + threadIdx.x, ccode, &sweeps);
    }
    if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
    st.contact_code[env] = ccode;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
    store_body<T>(st, n, env, b);
    finish_push<T, TOPO>(m, c, st, env, q, b, step_count, true);
}

// BaseObjectEnv.reset (base_object_env.py:146-173) for object_push: Robot.reset with the cube where the last episode left it,
// reset_object (teleport; rand_init_orn / rand_obj_mass draws, object_push_env.py:168-194), make_goal (:316-340; the simplex
// trajectory itself is filled in by k_gen_traj from the seed drawn here).
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_reset_push(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                   const uint8_t* __restrict__ mask) {
    // (no KtScope: these kernels run for milliseconds and live at the edge of the register file - the scope costs them up to 1.2 KB more scratch)
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double push_lds_raw[];
    const lds_ptr<T> lds = (lds_ptr<T>)push_lds_raw;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = c.num_envs;
    if (env >= n) return;
    if (mask != nullptr && mask[env] == 0) return;
    uint64_t rs = st.rng[env];
    const double ang = c.rand_init_orn ? rng_uniform(rs, -c.init_orn_range, c.init_orn_range) : 0.0;
    const double old_mass = st.obj_mass[env];
    const double new_mass = c.rand_obj_mass ? rng_uniform(rs, c.mass_lo, c.mass_hi) : old_mass;
    if (c.traj_type == TG_TRAJ_SIMPLEX) st.noise_seed[env] = (int64_t)rng_uniform(rs, 0.0, 1.0e8);
    else {
        const double ta = rng_uniform(rs, -c.traj_ang_range, c.traj_ang_range);
        double ys[TG_MAX_TRAJ_POINTS];
        for (int i = 0; i < c.traj_n; ++i) {
            const double dist = (double)i * c.traj_spacing;
            st.traj[(0 * TG_MAX_TRAJ_POINTS + i) * n + env] = c.traj_init_offset + dist * cos(ta);
            ys[i] = dist * sin(ta);
            st.traj[(1 * TG_MAX_TRAJ_POINTS + i) * n + env] = ys[i];
        }
        for (int i = 0; i < c.traj_n; ++i) {
            double g;
            if (i == 0) g = (ys[1] - ys[0]) / c.traj_spacing;
            else if (i == c.traj_n - 1) g = (ys[i] - ys[i - 1]) / c.traj_spacing;
            else g = (ys[i + 1] - ys[i - 1]) / (2.0 * c.traj_spacing);
            st.traj[(2 * TG_MAX_TRAJ_POINTS + i) * n + env] = g;
        }
    }
    st.rng[env] = rs;
    st.step_count[env] = 0;
    st.goal_id[env] = c.reset_goal_id;   // get_step_data at the end of reset may already have advanced the goal (tg_config.reset_goal_id)
    FreeBody<T> b = load_body<T>(st, n, env);
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = m.rest_q[i]; qd[i] = T(0); }
    const V3<T> tpos = load_v3(c.work_pos);               // update_init_pose: work-frame origin, rpy 0 (base_object_env.py:96-103)
    T trpy[3];
    euler_from_quat(quat_mul(c.work_q, quat_from_euler(T(0), T(0), T(0))), trpy[0], trpy[1], trpy[2]);
    const Q4<T> tq = quat_from_euler(trpy[0], trpy[1], trpy[2]);
    const M3<T> Rt = mat_from_quat(tq);
    T qik[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qik[i] = q[i];
    inverse_kinematics<T, TOPO>(m, tpos, Rt, qik, 100, T(1e-8));
    if (N == 8) { qik[N - 3] = qik[1]; qik[N - 2] = -qik[1]; qik[N - 1] = qik[1] + qik[2]; }
    T cv = T(0.001);
    T zero[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zero[i] = T(0);
    int used = 0, ccode = 0;
    for (int it = 0; it < 1000; ++it) {
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> p; M3<T> R;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, p, R);
        const Q4<T> cq = quat_from_mat(R);
        T diff[N], step_j[N], nrm2 = T(0), total_v = T(0);
        bool all_small = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diff[i] = qik[i] - q[i];
            nrm2 += diff[i] * diff[i];
            all_small = all_small && (tabs(diff[i]) < cv);
            total_v += tabs(qd[i]);
        }
        const T nrm = tsqrt(nrm2);
#pragma unroll
        for (int i = 0; i < N; ++i) step_j[i] = q[i] + ((nrm > T(0)) ? diff[i] / nrm : T(0)) * cv;
        if (all_small) cv = cv / T(2);
        sim_tick_push<T, TOPO, kMotorPosition>(m, q, qd, step_j, zero, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, b, c.push,
                                               (const T*)st.tip_verts, (T)old_mass, lds + threadIdx.x, ccode);
        ++used;
        const T pos_err = tabs(tpos.x - p.x) + tabs(tpos.y - p.y) + tabs(tpos.z - p.z);
        const T ip = tq.x * cq.x + tq.y * cq.y + tq.z * cq.z + tq.w * cq.w;
        T ca = T(2) * ip * ip - T(1);
        ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
        if (pos_err < T(2e-4) && tacos(ca) < T(1e-3) && total_v < T(0.1)) break;
    }
    st.reset_ticks[env] = used;
    st.contact_code[env] = ccode;
    // reset_object: resetBasePositionAndOrientation(init_obj_pos, init_obj_orn), velocities zeroed
    b.pos = load_v3(c.obj_init_pos);
    b.R = mat_from_quat(quat_from_euler((T)c.obj_init_rpy[0], (T)c.obj_init_rpy[1], (T)(c.obj_init_rpy[2] + ang)));
    b.v = mk<T>(0, 0, 0); b.w = mk<T>(0, 0, 0);
    st.obj_mass[env] = new_mass;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
    store_body<T>(st, n, env, b);
    finish_push<T, TOPO>(m, c, st, env, q, b, 0, false);
}

// ------------------------------------------------------------------------------------------------ object_roll
// get_step_data (object_roll_env.py:311-365): the goal, given in the TCP frame, is carried along with the TCP (update_goal :268-295);
// reward -|obj_xy - goal_xy|, done below 1 mm; extended_feature = goal_pos_tcp (:409-415); camera <- marble transform with the
// episode's scale (loadURDF globalScaling scales the visual).
template <typename T, int TOPO>
__device__ __forceinline__ void finish_roll(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const T (&q)[Topo<TOPO>::N],
                                            const FreeBody<T>& b, T scale, int step_count, bool write_reward_done) {
    const int n = c.num_envs;
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    T rpy[3];
    const Q4<T> qtcp = quat_from_mat(Rtcp);
    euler_from_quat(qtcp, rpy[0], rpy[1], rpy[2]);
    st.tcp_pos[0 * n + env] = (double)ptcp.x; st.tcp_pos[1 * n + env] = (double)ptcp.y; st.tcp_pos[2 * n + env] = (double)ptcp.z;
    st.tcp_rpy[0 * n + env] = (double)rpy[0]; st.tcp_rpy[1 * n + env] = (double)rpy[1]; st.tcp_rpy[2 * n + env] = (double)rpy[2];
    const V3<T> gt = mk((T)st.goal[0 * n + env], (T)st.goal[1 * n + env], (T)st.goal[2 * n + env]);
    if (write_reward_done) {
        const V3<T> gw = ptcp + mul(mat_from_quat(qtcp), gt);                              // multiplyTransforms(tcp pose, goal_pos_tcp)
        const T dx = b.pos.x - gw.x, dy = b.pos.y - gw.y;
        const T dist = tsqrt(dx * dx + dy * dy);                                           // xy_obj_dist_to_goal
        const bool at_goal = dist < c.term_dist;
        const float rw = (float)(c.reward_mode == TG_REWARD_SPARSE ? (at_goal ? T(1) : T(0)) : -(T(1) * dist));
        st.reward[env] = rw;
        st.done[env] = (at_goal || step_count >= c.max_steps) ? 1 : 0;
        episode_step(st, env, rw, at_goal || step_count >= c.max_steps, step_count);
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const float f = e == 0 ? (float)gt.x : (e == 1 ? (float)gt.y : (e == 2 ? (float)gt.z : 0.0f));
        if (write_reward_done) st.term_feature[(size_t)env * 12 + e] = f;
        st.feature[(size_t)env * 12 + e] = f;
    }
    V3<T> pb; M3<T> Rb;
    link_frame<T, TOPO>(k, m.sensor_link, m.sensor_pos, m.sensor_rot, pb, Rb);
    const V3<T> pc = pb + mul(Rb, load_v3(c.cam_pos));
    const M3<T> Rc = mul(Rb, c.cam_rot);
    V3<T> f{Rc.m[0], Rc.m[3], Rc.m[6]}, up{Rc.m[2], Rc.m[5], Rc.m[8]};
    f = (T(1) / norm(f)) * f;
    V3<T> s = cross(f, up);
    s = (T(1) / norm(s)) * s;
    const V3<T> u = cross(s, f);
    const V3<T> ox = scale * mk(b.R.m[0], b.R.m[3], b.R.m[6]), oy = scale * mk(b.R.m[1], b.R.m[4], b.R.m[7]), oz = scale * mk(b.R.m[2], b.R.m[5], b.R.m[8]);
    const V3<T> dp = b.pos - pc;
    const V3<T> nf = mk<T>(0, 0, 0) - f;
    float* X = st.stim_xform;
    X[0 * n + env] = (float)dot(s, ox);  X[1 * n + env] = (float)dot(s, oy);  X[2 * n + env] = (float)dot(s, oz);
    X[3 * n + env] = (float)dot(u, ox);  X[4 * n + env] = (float)dot(u, oy);  X[5 * n + env] = (float)dot(u, oz);
    X[6 * n + env] = (float)dot(nf, ox); X[7 * n + env] = (float)dot(nf, oy); X[8 * n + env] = (float)dot(nf, oz);
    X[9 * n + env] = (float)dot(s, dp);  X[10 * n + env] = (float)dot(u, dp); X[11 * n + env] = (float)dot(nf, dp);
}

template <typename T, int TOPO, bool POS>
__global__ __launch_bounds__(64) void k_step_roll(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                  const float* __restrict__ actions) {
    // (no KtScope: these kernels run for milliseconds and live at the edge of the register file - the scope costs them up to 1.2 KB more scratch)
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double push_lds_raw[];
    const lds_ptr<T> lds = (lds_ptr<T>)push_lds_raw;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = c.num_envs;
    if (env >= n) return;
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    FreeBody<T> b = load_body<T>(st, n, env);
    const float* a = actions + (size_t)env * c.act_dim;
    T enc[6] = {(T)a[0], (T)a[1], T(0), T(0), T(0), T(0)};   // encode_actions "xy" (object_roll_env.py:297-309)
    T vels[6];
    scale_actions<T>(c, enc, vels);
    const int step_count = st.step_count[env] + 1;
    st.step_count[env] = step_count;
    const T radius = (T)st.obj_mass[env];                                  // the episode's radius
    int ccode = 0, sweeps = 0;
    const T work_dz = (T)((2.0 * st.obj_mass[env] - st.embed[env]) - (double)c.work_pos[2]);   // update_workframe (:192-201)
    T qd_des[N], zero[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zero[i] = T(0);
    if constexpr (POS) {                                  // TCP_position_control: qd_des carries the joint targets
        V3<T> tpos; Q4<T> tq;
        tcp_position_target<T, TOPO>(m, c, q, vels, tpos, tq, qd_des, work_dz);
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = 0.0;
        for (int t = 0; t < c.max_blocking; ++t) {        // blocking_move(max_steps, constant_vel=None), robot.py:188-260
            const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
            sim_tick_push<T, TOPO, kMotorPosition, 1>(m, q, qd, qd_des, zero, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, b, c.push,
                                                      nullptr, radius, lds + threadIdx.x, ccode, &sweeps);
            if (stop) break;
        }
    } else {
        tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des, nullptr, work_dz);
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = (double)qd_des[i];
        for (int t = 0; t < c.action_repeat; ++t)
            sim_tick_push<T, TOPO, kMotorVelocity, 1>(m, q, qd, zero, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, b, c.push, nullptr,
                                                      radius, lds + threadIdx.x, ccode, &sweeps);
    }
    if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
    st.contact_code[env] = ccode;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
    store_body<T>(st, n, env, b);
    finish_roll<T, TOPO>(m, c, st, env, q, b, radius / (T)c.roll_radius, step_count, true);
}

// BaseObjectEnv.reset (base_object_env.py:153-190) for object_roll: reset_task (marble size, embed distance; :176-190), update_workframe,
// Robot.reset with the marble of the last episode still in the world, reset_object (:203-248), make_goal (:250-266).
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_reset_roll(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                   const uint8_t* __restrict__ mask) {
    // (no KtScope: these kernels run for milliseconds and live at the edge of the register file - the scope costs them up to 1.2 KB more scratch)
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double push_lds_raw[];
    const lds_ptr<T> lds = (lds_ptr<T>)push_lds_raw;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = c.num_envs;
    if (env >= n) return;
    if (mask != nullptr && mask[env] == 0) return;
    uint64_t rs = st.rng[env];
    const double old_radius = st.obj_mass[env];
    const double scaling = c.roll_rand_size ? rng_uniform(rs, 1.0, 2.0) : 1.0;
    const double new_radius = c.roll_radius * scaling;
    double embed = st.embed[env];
    if (c.roll_rand_embed) embed = rng_uniform(rs, c.embed_lo, c.embed_hi);
    double ix = 0.0, iy = 0.0;
    if (c.roll_rand_init_pos) { ix = rng_uniform(rs, -c.roll_init_range, c.roll_init_range); iy = rng_uniform(rs, -c.roll_init_range, c.roll_init_range); }
    const double gang = rng_uniform(rs, -3.141592653589793, 3.141592653589793);
    const double gdist = rng_uniform(rs, c.roll_goal_lo, c.roll_goal_hi);
    st.rng[env] = rs;
    st.step_count[env] = 0;
    st.embed[env] = embed;
    st.goal[0 * n + env] = gdist * cos(gang); st.goal[1 * n + env] = gdist * sin(gang); st.goal[2 * n + env] = 0.0;
    FreeBody<T> b = load_body<T>(st, n, env);
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = m.rest_q[i]; qd[i] = T(0); }
    const V3<T> tpos = mk(c.work_pos[0], c.work_pos[1], (T)(2.0 * new_radius - embed));   // work-frame origin of this episode, rpy 0
    T trpy[3];
    euler_from_quat(quat_mul(c.work_q, quat_from_euler(T(0), T(0), T(0))), trpy[0], trpy[1], trpy[2]);
    const Q4<T> tq = quat_from_euler(trpy[0], trpy[1], trpy[2]);
    const M3<T> Rt = mat_from_quat(tq);
    T qik[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qik[i] = q[i];
    inverse_kinematics<T, TOPO>(m, tpos, Rt, qik, 100, T(1e-8));
    if (N == 8) { qik[N - 3] = qik[1]; qik[N - 2] = -qik[1]; qik[N - 1] = qik[1] + qik[2]; }
    T cv = T(0.001);
    T zero[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zero[i] = T(0);
    int used = 0, ccode = 0;
    for (int it = 0; it < 1000; ++it) {
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> p; M3<T> R;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, p, R);
        const Q4<T> cq = quat_from_mat(R);
        T diff[N], step_j[N], nrm2 = T(0), total_v = T(0);
        bool all_small = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diff[i] = qik[i] - q[i];
            nrm2 += diff[i] * diff[i];
            all_small = all_small && (tabs(diff[i]) < cv);
            total_v += tabs(qd[i]);
        }
        const T nrm = tsqrt(nrm2);
#pragma unroll
        for (int i = 0; i < N; ++i) step_j[i] = q[i] + ((nrm > T(0)) ? diff[i] / nrm : T(0)) * cv;
        if (all_small) cv = cv / T(2);
        sim_tick_push<T, TOPO, kMotorPosition, 1>(m, q, qd, step_j, zero, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, b, c.push,
                                                  nullptr, (T)old_radius, lds + threadIdx.x, ccode);
        ++used;
        const T pos_err = tabs(tpos.x - p.x) + tabs(tpos.y - p.y) + tabs(tpos.z - p.z);
        const T ip = tq.x * cq.x + tq.y * cq.y + tq.z * cq.z + tq.w * cq.w;
        T ca = T(2) * ip * ip - T(1);
        ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
        if (pos_err < T(2e-4) && tacos(ca) < T(1e-3) && total_v < T(0.1)) break;
    }
    st.reset_ticks[env] = used;
    st.contact_code[env] = ccode;
    // reset_object: teleport (or reload with the new scale), velocities zeroed
    b.pos = mk((T)((double)c.obj_init_pos[0] + ix), (T)((double)c.obj_init_pos[1] + iy), (T)new_radius);
    const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
#pragma unroll
    for (int k = 0; k < 9; ++k) b.R.m[k] = ident[k];
    b.v = mk<T>(0, 0, 0); b.w = mk<T>(0, 0, 0);
    st.obj_mass[env] = new_radius;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
    store_body<T>(st, n, env, b);
    finish_roll<T, TOPO>(m, c, st, env, q, b, (T)scaling, 0, false);
}

// Recompute cached read-backs after tg_set_joint_state.
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_refresh(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st) {
    constexpr int N = Topo<TOPO>::N;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = cp->num_envs;
    if (env >= n) return;
    T q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = (T)st.q[i * n + env];
    finish_env<T, TOPO>(*mp, *cp, st, env, q, (T)st.edge_ang[env], st.step_count[env], false);
}

// tg_get_state's tcp_rpy for the envs stepped by k_step (edge_follow, surface_follow), whose steps leave that read-back alone: the TCP
// frame's world euler angles from the current joint angles, nothing else touched.
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_refresh_rpy(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st) {
    constexpr int N = Topo<TOPO>::N;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = cp->num_envs;
    if (env >= n) return;
    T q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = (T)st.q[i * n + env];
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(*mp, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, mp->tcp_link, mp->tcp_pos, mp->tcp_rot, ptcp, Rtcp);
    T rpy[3];
    { Q4<T> qq = quat_from_mat(Rtcp); euler_from_quat(qq, rpy[0], rpy[1], rpy[2]); }
    st.tcp_rpy[0 * n + env] = (double)rpy[0]; st.tcp_rpy[1 * n + env] = (double)rpy[1]; st.tcp_rpy[2 * n + env] = (double)rpy[2];
}

// ------------------------------------------------------------------------------------------------ function-level kernels
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_inverse_dynamics(const DevRobot<T>* __restrict__ mp, int n, const double* q, const double* qd, const double* qdd, double* tau) {
    constexpr int N = Topo<TOPO>::N;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    T qq[N], qv[N], hb[N], qdm[N], Minv[N][N], trM;
#pragma unroll
    for (int i = 0; i < N; ++i) { qq[i] = (T)q[s * N + i]; qv[i] = (T)qd[s * N + i]; }
    { Kin<T, TOPO> kk; dynamics_terms<T, TOPO>(*mp, qq, qv, hb, qdm, Minv, trM, load_v3(mp->gravity), kk); }
    // tau = M qdd + h ; M qdd obtained by solving Minv x = qdd would be circular, so rebuild M from Minv^-1 is avoided:
    // use linearity  ID(q, qd, qdd) = h + M qdd with M = inverse(Minv) computed by the pivoted solver column by column.
    T A[N][N], b[N], x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) A[i][j] = Minv[i][j];
        b[i] = (T)qdd[s * N + i];
    }
    solve_pivoted<T, N>(A, b, x);   // x = M qdd
#pragma unroll
    for (int i = 0; i < N; ++i) tau[s * N + i] = (double)(hb[i] + x[i]);
}

template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_mass_matrix(const DevRobot<T>* __restrict__ mp, int n, const double* q, double* M) {
    constexpr int N = Topo<TOPO>::N;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    T qq[N], qv[N], hb[N], qdm[N], Minv[N][N], trM;
#pragma unroll
    for (int i = 0; i < N; ++i) { qq[i] = (T)q[s * N + i]; qv[i] = T(0); }
    { Kin<T, TOPO> kk; dynamics_terms<T, TOPO>(*mp, qq, qv, hb, qdm, Minv, trM, load_v3(mp->gravity), kk); }
#pragma unroll
    for (int j = 0; j < N; ++j) {   // column j of M = solve(Minv, e_j)
        T A[N][N], b[N], x[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int jj = 0; jj < N; ++jj) A[i][jj] = Minv[i][jj];
            b[i] = (i == j) ? T(1) : T(0);
        }
        solve_pivoted<T, N>(A, b, x);
#pragma unroll
        for (int i = 0; i < N; ++i) M[(size_t)s * N * N + i * N + j] = (double)x[i];
    }
}

template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_jacobian(const DevRobot<T>* __restrict__ mp, int n, const double* q, double* J, double* pos, double* rot) {
    constexpr int N = Topo<TOPO>::N;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    T qq[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qq[i] = (T)q[s * N + i];
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(*mp, qq, k);
    V3<T> p; M3<T> R;
    link_frame<T, TOPO>(k, mp->tcp_link, mp->tcp_pos, mp->tcp_rot, p, R);
    T Jm[6][N];
    tcp_jacobian<T, TOPO>(*mp, k, p, Jm);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < N; ++i) J[(size_t)s * 6 * N + r * N + i] = (double)Jm[r][i];
    pos[s * 3 + 0] = (double)p.x; pos[s * 3 + 1] = (double)p.y; pos[s * 3 + 2] = (double)p.z;
#pragma unroll
    for (int e = 0; e < 9; ++e) rot[s * 9 + e] = (double)R.m[e];
}

template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_sim_ticks(const DevRobot<T>* __restrict__ mp, int n, int n_ticks, int iters, double dt, int motor_mode, const double* q_des,
                            const double* qd_des, double max_force, double* q, double* qd) {
    constexpr int N = Topo<TOPO>::N;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    T qq[N], qv[N], qdes[N], vdes[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        qq[i] = (T)q[s * N + i]; qv[i] = (T)qd[s * N + i];
        qdes[i] = q_des ? (T)q_des[s * N + i] : T(0); vdes[i] = qd_des ? (T)qd_des[s * N + i] : T(0);
    }
    for (int t = 0; t < n_ticks; ++t) {
        if (motor_mode == kMotorVelocity) sim_tick<T, TOPO, kMotorVelocity>(*mp, qq, qv, qdes, vdes, T(0), mp->vel_gain, (T)max_force, (T)dt, iters);
        else if (motor_mode == kMotorPosition) sim_tick<T, TOPO, kMotorPosition>(*mp, qq, qv, qdes, vdes, mp->pos_gain, mp->vel_gain, (T)max_force, (T)dt, iters);
        else sim_tick<T, TOPO, kMotorOff>(*mp, qq, qv, qdes, vdes, T(0), T(0), T(0), (T)dt, iters);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { q[s * N + i] = (double)qq[i]; qd[s * N + i] = (double)qv[i]; }
}

template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_ik(const DevRobot<T>* __restrict__ mp, int n, const double* q0, const double* tpos, const double* trot,
                                           int max_iters, double threshold, double* q_out, int32_t* iters) {
    constexpr int N = Topo<TOPO>::N;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    T q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = (T)q0[s * N + i];
    M3<T> Rt;
#pragma unroll
    for (int e = 0; e < 9; ++e) Rt.m[e] = (T)trot[s * 9 + e];
    const int it = inverse_kinematics<T, TOPO>(*mp, mk((T)tpos[s * 3], (T)tpos[s * 3 + 1], (T)tpos[s * 3 + 2]), Rt, q, max_iters, (T)threshold);
#pragma unroll
    for (int i = 0; i < N; ++i) q_out[s * N + i] = (double)q[i];
    iters[s] = it;
}

}  // namespace tg
