// tg_api.hip — env-step / reset kernels and the C ABI of libtactile_gym_hip.so (see include/tactile_gym_hip.h).
//
// Per-env state lives in HBM as struct-of-arrays with the env index minor ([field][num_envs], doubles), so a
// wavefront of 64 consecutive envs reads or writes one 512-byte contiguous run per field: the whole dynamic state
// crosses HBM exactly once per env step (in) and once (out); the 24 sim ticks in between run out of registers.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tactile_gym_hip.h"
#include "tg_physics.hpp"
#include "tg_noise.h"
#include "tg_raster.h"

namespace tg {

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define TG_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

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
    double *dir, *goal, *heights;   // [2][n], [3][n], [n][rows*cols]
    double* accum;                  // [n] sparse reward: the episode's accumulated dense reward
    float* surf_zoff;               // [n]
    int64_t* noise_seed;            // [n]
    // object_balance
    double *body_pos, *body_rot, *body_v, *body_w, *ext_pos, *gravity;   // [3][n], [9][n], [3][n], [3][n], [3][n], [n]
    uint8_t* ext_pending;           // [n]
    // object_push
    double *traj, *obj_mass;        // [3][TG_MAX_TRAJ_POINTS][n] work-frame x, y, yaw; [n]
    int32_t* goal_id;               // [n]
    int32_t* contact_code;          // [n] contact pairs of the last sim tick (sim_tick_push's contact_code)
    float *feature, *term_feature;  // [n][12] extended_feature observation (object_push_env.py:611-629), AoS
    const void* tip_verts;          // [n_tip][3] in the physics dtype
};

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
template <typename T>
__device__ __forceinline__ void world_to_work(const EnvConst<T>& c, V3<T> pos, const M3<T>& R, V3<T>& wpos, T (&wrpy)[3], T (&rpy_world)[3]) {
    Q4<T> q = quat_from_mat(R);
    euler_from_quat(q, rpy_world[0], rpy_world[1], rpy_world[2]);
    const Q4<T> q2 = quat_from_euler(rpy_world[0], rpy_world[1], rpy_world[2]);
    wpos = load_v3(c.work_inv_pos) + mul(c.work_Rinv, pos);
    const Q4<T> qw = quat_mul(c.work_qinv, q2);
    euler_from_quat(qw, wrpy[0], wrpy[1], wrpy[2]);
}

// np.digitize(v, np.linspace(lo, hi, n)) for increasing bins: the number of bin edges <= v.  Edges are formed exactly like
// numpy's linspace (k * step + lo in two roundings, last edge = hi), hence no FMA contraction here.
__device__ inline int digitize_linspace(double v, double lo, double hi, int n) {
#pragma clang fp contract(off)
    const double step = (hi - lo) / (double)(n - 1);
    int count = 0;
    for (int k = 0; k < n; ++k) {
        const double prod = (double)k * step;
        const double edge = (k == n - 1) ? hi : prod + lo;
        count += (edge <= v) ? 1 : 0;
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

// Everything that follows the physics of a step or a reset: TCP pose read-back, reward / termination
// (edge_follow_env.py:371-452) and the camera<-stimulus transform handed to the raster (tactile_sensor.py:150-229).
template <typename T, int TOPO>
__device__ __forceinline__ void finish_env(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const T (&q)[Topo<TOPO>::N],
                                           T edge_ang, int step_count, bool write_reward_done, const JointTrig<T, Topo<TOPO>::N>* trig = nullptr,
                                           bool lazy_rpy = false /* edge_follow steps: tcp_rpy is a read-back only, tg_get_state refreshes it */) {
    const int n = c.num_envs;
    Kin<T, TOPO> k;
    if (trig != nullptr) forward_kinematics<T, TOPO, true>(m, q, k, trig);
    else forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    st.tcp_pos[0 * n + env] = (double)ptcp.x; st.tcp_pos[1 * n + env] = (double)ptcp.y; st.tcp_pos[2 * n + env] = (double)ptcp.z;
    if (!(lazy_rpy && c.env_kind == TG_ENV_EDGE_FOLLOW)) {
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
    }
    if ((write_reward_done || c.reward_mode == TG_REWARD_SPARSE) && c.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        // get_step_data / dense_reward (base_surface_env.py:664-684, 703-760; surface_follow_auto_env.py:75-94)
        const int R = c.surf_rows, Cc = c.surf_cols;
        int ti = digitize_linspace((double)ptcp.y, c.ybin_lo, c.ybin_hi, Cc);   // xy_to_surface_idx (:284-300)
        int tj = digitize_linspace((double)ptcp.x, c.xbin_lo, c.xbin_hi, R);
        if (ti == Cc) ti -= 1;
        if (tj == R) tj -= 1;
        const double* H = st.heights + (size_t)env * R * Cc;
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
        }
    }
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
    V3<T> wpos; T wrpy[3] = {T(0), T(0), T(0)}, rpyw[3];
    // The work-frame orientation of the TCP (matrix -> quaternion -> euler -> quaternion -> multiply -> euler: nine f64 transcendentals)
    // only enters the limit check of the rotational components; a movement mode without rotational velocity (a zero stays a zero in that
    // check) needs the position alone.  Wave-uniform.
    if (__any(vels[3] != T(0) || vels[4] != T(0) || vels[5] != T(0)))
        world_to_work(c, mk(ptcp.x, ptcp.y, ptcp.z - work_dz), Rtcp, wpos, wrpy, rpyw);
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
    T J[6][N];
    tcp_jacobian<T, TOPO>(m, k, ptcp, J);
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
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_step(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                             const float* __restrict__ actions) {
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
    tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des, &trig);
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
    for (int t = 0; t < c.action_repeat; ++t) {
        const int before = verified;
        sim_tick<T, TOPO, kMotorVelocity, true, true>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, &trig,
                                                      &verified);
        const bool analytic = before > 0 && verified == before - 1;   // the analytic path decrements; a full solve sets 24 or -1
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

#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { st.trig_sc[i * n + env] = (double)trig.s[i]; st.trig_sc[(8 + i) * n + env] = (double)trig.c[i]; }
    finish_env<T, TOPO>(m, c, st, env, q, (T)st.edge_ang[env], step_count, true, &trig, true);
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
        world_to_work(c, mk(ptcp.x, ptcp.y, ptcp.z - work_dz), Rtcp, wpos, wrpy, rpyw);
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
    int verified = 0;
    for (int it = 0; it < c.max_blocking; ++it) {
        const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
        sim_tick<T, TOPO, kMotorPosition>(m, q, qd, qik, zero, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, nullptr, &verified);
        if (verified < 0) verified = 0;
        if (stop) break;
    }
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
        const double* H = st.heights + (size_t)env * R * Cc;
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
    st.licence[env] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
    finish_env<T, TOPO>(m, c, st, env, q, (T)edge_ang, 0, false);
}

template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_reset(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                              const uint8_t* __restrict__ mask, int phase) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= cp->num_envs) return;
    if (mask != nullptr && mask[env] == 0) return;
    if (cp->fused_reset && mask != nullptr) {   // auto-reset inside tg_step: keep the terminal observation's camera transform; the render
        const int n = cp->num_envs;             // launch that follows draws both images of this env
#pragma unroll
        for (int k = 0; k < 12; ++k) st.term_xform[k * n + env] = st.stim_xform[k * n + env];
    }
    reset_env<T, TOPO>(*mp, *cp, st, env, phase);
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
template <typename T> __device__ __forceinline__ T wrap_deg(T d) {   // ((d + 180) % 360) - 180 with numpy's sign-of-divisor modulo
    const T x = d + T(180);
    return (x - T(360) * floor(x / T(360))) - T(180);
}

// get_step_data / check_obj_fall / termination (object_balance_env.py:426-497) + camera<-object transform
template <typename T, int TOPO>
__device__ __forceinline__ void finish_body(const DevRobot<T>& m, const EnvConst<T>& c, const State& st, int env, const T (&q)[Topo<TOPO>::N],
                                            const FreeBody<T>& b, T embed, int step_count, bool write_reward_done) {
    const int n = c.num_envs;
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp; M3<T> Rtcp;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    T rpy[3];
    { Q4<T> qq = quat_from_mat(Rtcp); euler_from_quat(qq, rpy[0], rpy[1], rpy[2]); }
    st.tcp_pos[0 * n + env] = (double)ptcp.x; st.tcp_pos[1 * n + env] = (double)ptcp.y; st.tcp_pos[2 * n + env] = (double)ptcp.z;
    st.tcp_rpy[0 * n + env] = (double)rpy[0]; st.tcp_rpy[1 * n + env] = (double)rpy[1]; st.tcp_rpy[2 * n + env] = (double)rpy[2];
    if (write_reward_done) {
        T orpy[3];
        { Q4<T> qq = quat_from_mat(b.R); euler_from_quat(qq, orpy[0], orpy[1], orpy[2]); }
        const T r2d = T(180) / T(3.141592653589793);
        const T d0 = tabs(wrap_deg(orpy[0] * r2d - c.obj_init_rpy_deg[0])), d1 = tabs(wrap_deg(orpy[1] * r2d - c.obj_init_rpy_deg[1]));
        const V3<T> init = mk(c.work_pos[0], c.work_pos[1], c.work_pos[2] + (c.obj_base_height / T(2)) - embed);
        const bool fell = d0 > c.term_deg || d1 > c.term_deg || norm(b.pos - init) > c.term_pos;
        const bool done = fell || step_count >= c.max_steps;
        const T reward = (c.reward_mode == TG_REWARD_SPARSE) ? (fell ? T(-1) : T(0)) : T(1);
        st.reward[env] = (float)reward;
        st.done[env] = done ? 1 : 0;
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
}

template <typename T, int TOPO, bool POS>
__global__ __launch_bounds__(64) void k_step_body(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                  const float* __restrict__ actions) {
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
    int verified = 0;
    if constexpr (POS) {                                  // blocking_move(max_steps, constant_vel=None), robot.py:188-260
        for (int t = 0; t < c.max_blocking; ++t) {
            const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
            sim_tick_body<T, TOPO, kMotorPosition>(m, q, qd, qd_des, qdummy, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, b,
                                                   c.body, pivot_b, fext, pext, pending && t == 0, &verified);
            if (stop) break;
        }
    } else {
        JointTrig<T, N> trig;              // sines / cosines of the joint angles, advanced by angle addition through the analytic ticks
        trig_init<T, N>(q, trig);
        for (int t = 0; t < c.action_repeat; ++t)
            sim_tick_body<T, TOPO, kMotorVelocity>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, b, c.body,
                                                   pivot_b, fext, pext, pending && t == 0, &verified, &trig);
    }
    st.ext_pending[env] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
    store_body<T>(st, n, env, b);
    finish_body<T, TOPO>(m, c, st, env, q, b, embed, step_count, true);
}

// BaseObjectEnv.reset (base_object_env.py:146-173) for object_balance: reset_task (gravity, embed), Robot.reset with the pole
// still tied to the TCP, reset_object (teleport + one-shot random force).
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_reset_body(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                   const uint8_t* __restrict__ mask) {
    constexpr int N = Topo<TOPO>::N;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = c.num_envs;
    if (env >= n) return;
    if (mask != nullptr && mask[env] == 0) return;
    uint64_t rs = st.rng[env];
    const double gz = c.rand_gravity ? rng_uniform(rs, c.gravity_lo, c.gravity_hi) : c.gravity_default;   // reset_task :301-306
    double embed = st.embed[env];
    if (c.rand_embed) embed = rng_uniform(rs, c.embed_lo, c.embed_hi);                                    // :308-316
    st.gravity[env] = gz;
    st.embed[env] = embed;
    st.step_count[env] = 0;
    const V3<T> grav = mk(T(0), T(0), (T)gz);
    const V3<T> pivot_b = mk(T(0), T(0), -c.obj_base_height / T(2) + (T)embed);
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
    T qik[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qik[i] = q[i];
    inverse_kinematics<T, TOPO>(m, tpos, Rt, qik, 100, T(1e-8));
    T cv = T(0.001);
    T zero[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zero[i] = T(0);
    const V3<T> z3 = mk<T>(0, 0, 0);
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
        sim_tick_body<T, TOPO, kMotorPosition>(m, q, qd, step_j, zero, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, grav, b, c.body,
                                               pivot_b, z3, z3, false, &verified);
        ++used;
        const T pos_err = tabs(tpos.x - p.x) + tabs(tpos.y - p.y) + tabs(tpos.z - p.z);
        const T ip = tq.x * cq.x + tq.y * cq.y + tq.z * cq.z + tq.w * cq.w;
        T ca = T(2) * ip * ip - T(1);
        ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
        if (pos_err < T(2e-4) && tacos(ca) < T(1e-3) && total_v < T(0.1)) break;
    }
    st.reset_ticks[env] = used;
    // reset_object (object_balance_env.py:330-381): teleport, then a one-shot downward force at a random point of the base plate
    b.pos = mk(c.work_pos[0], c.work_pos[1], c.work_pos[2] + (c.obj_base_height / T(2)) - (T)embed);
    b.R = c.obj_init_rot;
    b.v = z3; b.w = z3;
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
    finish_body<T, TOPO>(m, c, st, env, q, b, (T)embed, 0, false);
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
    int ccode = 0;
    T qdummy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qdummy[i] = T(0);
    if constexpr (POS) {                                  // blocking_move(max_steps, constant_vel=None), robot.py:188-260
        for (int t = 0; t < c.max_blocking; ++t) {
            const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
            sim_tick_push<T, TOPO, kMotorPosition>(m, q, qd, qd_des, qdummy, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, b, c.push,
                                                   (const T*)st.tip_verts, mass, lds + threadIdx.x, ccode);
            if (stop) break;
        }
    } else {
        for (int t = 0; t < c.action_repeat; ++t)
            sim_tick_push<T, TOPO, kMotorVelocity>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, b, c.push,
                                                   (const T*)st.tip_verts, mass, lds + threadIdx.x, ccode);
    }
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
        st.reward[env] = (float)(c.reward_mode == TG_REWARD_SPARSE ? (at_goal ? T(1) : T(0)) : -(T(1) * dist));
        st.done[env] = (at_goal || step_count >= c.max_steps) ? 1 : 0;
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
    int ccode = 0;
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
                                                      nullptr, radius, lds + threadIdx.x, ccode);
            if (stop) break;
        }
    } else {
        tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des, nullptr, work_dz);
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = (double)qd_des[i];
        for (int t = 0; t < c.action_repeat; ++t)
            sim_tick_push<T, TOPO, kMotorVelocity, 1>(m, q, qd, zero, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, b, c.push, nullptr,
                                                      radius, lds + threadIdx.x, ccode);
    }
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

// ------------------------------------------------------------------------------------------------ host helpers
static void h_quat_from_euler(const double* rpy, double* q) {
    const double phi = 0.5 * rpy[0], the = 0.5 * rpy[1], psi = 0.5 * rpy[2];
    q[0] = sin(phi) * cos(the) * cos(psi) - cos(phi) * sin(the) * sin(psi);
    q[1] = cos(phi) * sin(the) * cos(psi) + sin(phi) * cos(the) * sin(psi);
    q[2] = cos(phi) * cos(the) * sin(psi) - sin(phi) * sin(the) * cos(psi);
    q[3] = cos(phi) * cos(the) * cos(psi) + sin(phi) * sin(the) * sin(psi);
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) q[k] /= nrm;
}
static void h_mat_from_quat(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys,
                 yz = y * zs, zz = z * zs;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
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

template <typename T> static int build_env_const(const tg_config& cfg, const tg_sensor& sen, const tg_robot& rob, EnvConst<T>& c) {
    memset(&c, 0, sizeof c);
    c.num_envs = cfg.num_envs;
    c.env_kind = cfg.env_kind;
    c.movement_mode = cfg.movement_mode; c.noise_mode = cfg.noise_mode; c.reward_mode = cfg.reward_mode;
    if (cfg.env_kind == TG_ENV_EDGE_FOLLOW) {
        switch (cfg.movement_mode) {     // get_act_dim, edge_follow_env.py:478-486
            case TG_MOVE_XY: c.act_dim = 2; break;
            case TG_MOVE_XYZ: case TG_MOVE_XYRZ: c.act_dim = 3; break;
            case TG_MOVE_XYZRZ: c.act_dim = 4; break;
            default: return fail(-1, "Incorrect movement mode specified");
        }
    } else if (cfg.env_kind == TG_ENV_OBJECT_BALANCE) {
        switch (cfg.movement_mode) {     // get_act_dim, object_balance_env.py:565-576
            case TG_BMOVE_XY: case TG_BMOVE_RXRY: c.act_dim = 2; break;
            case TG_BMOVE_XYZ: c.act_dim = 3; break;
            case TG_BMOVE_XYRXRY: c.act_dim = 4; break;
            default: return fail(-1, "Incorrect movement mode specified");
        }
        if (cfg.obj_mass <= 0) return fail(-1, "object_balance: object mass must be positive");
        c.body.mass = (T)cfg.obj_mass;
        c.body.com = {(T)cfg.obj_com[0], (T)cfg.obj_com[1], (T)cfg.obj_com[2]};
        c.body.inertia = {(T)cfg.obj_inertia[0], (T)cfg.obj_inertia[1], (T)cfg.obj_inertia[2], (T)cfg.obj_inertia[4], (T)cfg.obj_inertia[5],
                          (T)cfg.obj_inertia[8]};
        c.body.erp = (T)cfg.p2p_erp; c.body.max_impulse = (T)cfg.p2p_max_impulse;
        c.body.link = rob.tcp_link;   // createConstraint parent = TCP link, parentFramePosition 0 in its inertial frame (:271-283)
        c.body.pivot_a = {(T)rob.tcp_pos[0], (T)rob.tcp_pos[1], (T)rob.tcp_pos[2]};
        double oq[4], oR[9];
        h_quat_from_euler(cfg.obj_init_rpy, oq);
        h_mat_from_quat(oq, oR);
        for (int k = 0; k < 9; ++k) c.obj_init_rot.m[k] = (T)oR[k];
        for (int k = 0; k < 3; ++k) { c.obj_init_rpy_deg[k] = (T)(cfg.obj_init_rpy[k] * 180.0 / 3.141592653589793); c.ext_force[k] = (T)cfg.ext_force[k]; }
        c.obj_base_width = (T)cfg.obj_base_width; c.obj_base_height = (T)cfg.obj_base_height;
        c.term_deg = (T)cfg.term_deg; c.term_pos = (T)cfg.term_pos;
        c.rand_gravity = cfg.rand_gravity; c.rand_embed = cfg.rand_embed;
        c.gravity_lo = cfg.gravity_lo; c.gravity_hi = cfg.gravity_hi; c.gravity_default = cfg.gravity_default;
    } else if (cfg.env_kind == TG_ENV_OBJECT_ROLL) {
        c.act_dim = 2;                   // movement_mode "xy" (object_roll_env.py:417-422)
        if (cfg.movement_mode != 0) return fail(-1, "Incorrect movement mode specified");
        if (cfg.obj_mass <= 0 || cfg.roll_radius <= 0) return fail(-1, "object_roll: mass and radius must be positive");
        if (cfg.tip_link < 0 || cfg.tip_link >= rob.ndof) return fail(-1, "object_roll: tip_link out of range");
        if (rob.topology == 1 && cfg.tip_link >= Topo<1>::NP) return fail(-1, "object_roll: the tip must hang off the MG400's main chain (j1..j5)");
        PushScene<T>& ps = c.push;
        ps.table_z = (T)cfg.table_z;
        ps.mu_table = (T)cfg.mu_table; ps.mu_tip = (T)cfg.mu_tip;
        ps.breaking = (T)cfg.contact_breaking; ps.erp = (T)cfg.contact_erp; ps.tip_stiffness = (T)cfg.tip_stiffness; ps.tip_damping = (T)cfg.tip_damping;
        ps.lin_damp = (T)cfg.obj_lin_damp; ps.ang_damp = (T)cfg.obj_ang_damp;
        const double i0 = 0.4 * cfg.obj_mass * cfg.roll_radius * cfg.roll_radius;   // solid sphere (btSphereShape::calculateLocalInertia)
        ps.inertia0[0] = (T)i0; ps.inertia0[3] = (T)i0; ps.inertia0[5] = (T)i0;
        ps.mass0 = (T)cfg.obj_mass; ps.radius0 = (T)cfg.roll_radius;
        ps.tip_link = cfg.tip_link; ps.n_tip = 0; ps.cone_friction = cfg.cone_friction;
        for (int k = 0; k < 3; ++k) { ps.cyl_pos[k] = (T)cfg.tip_cyl_pos[k]; c.obj_init_pos[k] = (T)cfg.obj_init_pos[k]; }
        for (int k = 0; k < 9; ++k) ps.cyl_rot.m[k] = (T)cfg.tip_cyl_rot[k];
        ps.cyl_hl = (T)cfg.tip_cyl_half_len; ps.cyl_r = (T)cfg.tip_cyl_radius;
        c.roll_rand_init_pos = cfg.roll_rand_init_pos; c.roll_rand_size = cfg.roll_rand_size; c.roll_rand_embed = cfg.roll_rand_embed;
        c.roll_radius = cfg.roll_radius; c.roll_init_range = cfg.roll_init_range; c.roll_goal_lo = cfg.roll_goal_lo; c.roll_goal_hi = cfg.roll_goal_hi;
        c.obj_mass0 = cfg.obj_mass;
    } else if (cfg.env_kind == TG_ENV_OBJECT_PUSH) {
        switch (cfg.movement_mode) {     // get_act_dim, object_push_env.py:631-644
            case TG_PMOVE_Y: c.act_dim = 1; break;
            case TG_PMOVE_YRZ: case TG_PMOVE_TYRZ: c.act_dim = 2; break;
            case TG_PMOVE_XYRZ: case TG_PMOVE_TXTYRZ: c.act_dim = 3; break;
            default: return fail(-1, "Incorrect movement mode specified");
        }
        if (cfg.obj_mass <= 0) return fail(-1, "object_push: object mass must be positive");
        if (cfg.traj_n_points < 2 || cfg.traj_n_points > TG_MAX_TRAJ_POINTS) return fail(-1, "object_push: traj_n_points must be in [2, TG_MAX_TRAJ_POINTS]");
        if (cfg.traj_type != TG_TRAJ_SIMPLEX && cfg.traj_type != TG_TRAJ_STRAIGHT) return fail(-1, "object_push: unknown traj_type");
        if (cfg.n_tip_verts <= 0 || !cfg.tip_verts) return fail(-1, "object_push: the tip collision hull is missing");
        if (cfg.tip_link < 0 || cfg.tip_link >= rob.ndof) return fail(-1, "object_push: tip_link out of range");
        if (rob.topology == 1 && cfg.tip_link >= Topo<1>::NP) return fail(-1, "object_push: the tip must hang off the MG400's main chain (j1..j5)");
        PushScene<T>& ps = c.push;
        ps.table_z = (T)cfg.table_z;
        for (int k = 0; k < 3; ++k) { ps.half[k] = (T)cfg.obj_half[k]; ps.com[k] = (T)cfg.obj_com[k]; c.obj_init_pos[k] = (T)cfg.obj_init_pos[k]; c.obj_init_rpy[k] = cfg.obj_init_rpy[k]; }
        ps.mu_table = (T)cfg.mu_table; ps.mu_tip = (T)cfg.mu_tip; ps.margin_cube = (T)cfg.margin_cube; ps.margin_tip = (T)cfg.margin_tip;
        ps.breaking = (T)cfg.contact_breaking; ps.erp = (T)cfg.contact_erp; ps.tip_stiffness = (T)cfg.tip_stiffness; ps.tip_damping = (T)cfg.tip_damping;
        ps.lin_damp = (T)cfg.obj_lin_damp; ps.ang_damp = (T)cfg.obj_ang_damp;
        const int sym[6] = {0, 1, 2, 4, 5, 8};
        for (int k = 0; k < 6; ++k) ps.inertia0[k] = (T)cfg.obj_inertia[sym[k]];
        ps.mass0 = (T)cfg.obj_mass;
        ps.tip_link = cfg.tip_link; ps.n_tip = cfg.n_tip_verts; ps.cone_friction = cfg.cone_friction;
        if (cfg.reset_goal_id < 0 || cfg.reset_goal_id > 1) return fail(-1, "object_push: reset_goal_id must be 0 or 1");
        c.reset_goal_id = cfg.reset_goal_id;
        c.traj_type = cfg.traj_type; c.traj_n = cfg.traj_n_points; c.rand_init_orn = cfg.rand_init_orn; c.rand_obj_mass = cfg.rand_obj_mass;
        c.traj_spacing = cfg.traj_spacing; c.traj_max_perturb = cfg.traj_max_perturb; c.traj_init_offset = cfg.traj_init_offset;
        c.mass_lo = cfg.mass_lo; c.mass_hi = cfg.mass_hi; c.init_orn_range = cfg.init_orn_range; c.traj_ang_range = cfg.traj_ang_range;
        c.obj_mass0 = cfg.obj_mass;
    } else {
        switch (cfg.movement_mode) {     // surface_follow_auto_env.py:96-107 / surface_follow_goal_env.py:112-123
            case TG_SMOVE_YZ: c.act_dim = cfg.surf_goal_variant ? 2 : 1; break;
            case TG_SMOVE_XYZ: c.act_dim = cfg.surf_goal_variant ? 3 : 1; break;
            case TG_SMOVE_YZRX: c.act_dim = cfg.surf_goal_variant ? 3 : 2; break;
            case TG_SMOVE_XYZRXRY: c.act_dim = cfg.surf_goal_variant ? 5 : 3; break;
            case TG_SMOVE_XRZ: c.act_dim = 2; break;                         // surface_follow_vert_env.py:102-113
            default: return fail(-1, "Incorrect movement mode specified");
        }
        c.surf_goal = cfg.surf_goal_variant ? 1 : 0;
        c.surf_vertical = cfg.surf_vertical ? 1 : 0;
        if (c.surf_vertical != (cfg.movement_mode == TG_SMOVE_XRZ ? 1 : 0) || c.surf_vertical != (cfg.noise_mode == TG_SNOISE_VERTICAL_SIMPLEX ? 1 : 0))
            return fail(-1, "Incorrect movement mode specified");   // xRz <-> vertical_simplex (base_surface_env.py:462-470)
        if (c.surf_vertical && c.surf_goal) return fail(-1, "surface_follow: the goal variant has no vertical surface");
        {
            const double srpy[3] = {0.0, c.surf_vertical ? -1.5707963267948966 : 0.0, 0.0};   // surface_orn (:263)
            double sq[4], sR[9];
            h_quat_from_euler(srpy, sq);
            h_mat_from_quat(sq, sR);
            for (int k = 0; k < 9; ++k) c.stim_R.m[k] = (T)sR[k];
        }

        if (cfg.noise_mode < TG_SNOISE_SIMPLEX || cfg.noise_mode > TG_SNOISE_VERTICAL_SIMPLEX) return fail(-1, "Incorrect noise mode specified");
        if (cfg.surf_rows < 2 || cfg.surf_cols < 2) return fail(-1, "surface_follow: heightfield needs at least 2x2 samples");
        c.surf_rows = cfg.surf_rows; c.surf_cols = cfg.surf_cols;
        c.surf_scale = cfg.surf_grid_scale; c.surf_range = cfg.surf_height_range; c.surf_interp = cfg.surf_interp;
        c.surf_extent = cfg.surf_xy_extent; c.auto_scale = cfg.auto_action_scale;
        // x_bins / y_bins = linspace(pos -+ (n/2) * grid_scale, n)   (base_surface_env.py:272-282)
        c.xbin_lo = cfg.stim_pos[0] - ((cfg.surf_rows / 2.0) * cfg.surf_grid_scale);
        c.xbin_hi = cfg.stim_pos[0] + ((cfg.surf_rows / 2.0) * cfg.surf_grid_scale);
        c.ybin_lo = cfg.stim_pos[1] - ((cfg.surf_cols / 2.0) * cfg.surf_grid_scale);
        c.ybin_hi = cfg.stim_pos[1] + ((cfg.surf_cols / 2.0) * cfg.surf_grid_scale);
    }
    c.fused_reset = (cfg.auto_reset && cfg.env_kind == TG_ENV_EDGE_FOLLOW) ? 1 : 0;
    if (cfg.control_mode != TG_CONTROL_TCP_VELOCITY && cfg.control_mode != TG_CONTROL_TCP_POSITION) return fail(-1, "Incorrect control mode specified");
    if (cfg.control_mode == TG_CONTROL_TCP_POSITION) {
        if (cfg.max_blocking_steps < 1) return fail(-1, "TCP_position_control: max_blocking_steps must be >= 1");
    }
    c.control_mode = cfg.control_mode; c.max_blocking = cfg.max_blocking_steps;
    c.max_steps = cfg.max_steps; c.action_repeat = cfg.action_repeat; c.solver_iters = cfg.pgs_full_sweeps ? -cfg.solver_iterations : cfg.solver_iterations;
    c.dt = (T)cfg.sim_dt; c.min_action = (T)cfg.min_action; c.max_action = (T)cfg.max_action;
    for (int d = 0; d < 6; ++d) { c.act_lo[d] = (T)cfg.act_lo[d]; c.act_hi[d] = (T)cfg.act_hi[d]; c.tcp_lims[d][0] = (T)cfg.tcp_lims[d][0]; c.tcp_lims[d][1] = (T)cfg.tcp_lims[d][1]; }
    double wq[4], wqi[4], R[9], Ri[9];
    h_quat_from_euler(cfg.workframe_rpy, wq);
    wqi[0] = -wq[0]; wqi[1] = -wq[1]; wqi[2] = -wq[2]; wqi[3] = wq[3];
    h_mat_from_quat(wq, R); h_mat_from_quat(wqi, Ri);
    c.work_q = {(T)wq[0], (T)wq[1], (T)wq[2], (T)wq[3]};
    c.work_qinv = {(T)wqi[0], (T)wqi[1], (T)wqi[2], (T)wqi[3]};
    for (int k = 0; k < 9; ++k) { c.work_R.m[k] = (T)R[k]; c.work_Rinv.m[k] = (T)Ri[k]; }
    for (int k = 0; k < 3; ++k) {
        c.work_pos[k] = (T)cfg.workframe_pos[k];
        c.work_inv_pos[k] = (T)(-(Ri[3 * k] * cfg.workframe_pos[0] + Ri[3 * k + 1] * cfg.workframe_pos[1] + Ri[3 * k + 2] * cfg.workframe_pos[2]));
        c.stim_pos[k] = (T)cfg.stim_pos[k];
        c.cam_pos[k] = (T)sen.cam_pos[k];
    }
    c.edge_height = (T)cfg.edge_height; c.edge_len = (T)cfg.edge_len; c.term_dist = (T)cfg.termination_dist;
    c.embed_default = (T)cfg.embed_dist; c.embed_lo = cfg.embed_lo; c.embed_hi = cfg.embed_hi;
    double cq[4], cR[9];
    h_quat_from_euler(sen.cam_rpy, cq);
    h_mat_from_quat(cq, cR);
    for (int k = 0; k < 9; ++k) c.cam_rot.m[k] = (T)cR[k];
    return 0;
}

}  // namespace tg

// ==================================================================================================== context
struct tg_ctx {
    tg_config cfg;
    tg_robot robot;
    int H, W, act_dim;
    hipStream_t own_stream = nullptr, stream = nullptr;
    void *d_robot = nullptr, *d_const = nullptr;   // DevRobot<T>, EnvConst<T>
    tg::State st{};
    tg::RasterParams rp{};
    float *d_nodef_dep = nullptr, *d_verts = nullptr, *d_soup = nullptr, *d_actions = nullptr;
    uint8_t* d_nodef_gray = nullptr;   // uint8(nodef_gray)
    uint8_t *d_border = nullptr, *d_obs = nullptr, *d_term = nullptr, *d_mask = nullptr;
    size_t packed_obs_bytes = 0, packed_bytes = 0, packed_feature_off = 0;   // d_obs = [obs | pad to 16 | reward f32[n] | done u8[n] | pad to 4 | feature f32[n][12]]
    int32_t* d_tris = nullptr;
    int n_tris = 0;
    tg::Stimulus stim{};
    // hipGraph of one tg_step launch sequence, keyed by the device action pointer it was captured with (launch-bound inner loop:
    // 3-4 kernels per step, one graph launch instead)
    hipStream_t aux_stream = nullptr;                // object_balance: the reset of finished envs runs here, beside the render
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipGraphExec_t step_graph = nullptr;
    const float* step_graph_actions = nullptr;
    hipStream_t step_graph_stream = nullptr;
    bool graph_broken = false;
    // profiling
    bool profile = false;
    struct Ev { hipEvent_t a, b; int which; };
    std::vector<Ev> events;
    double prof_ms[4] = {0, 0, 0, 0};
    int64_t prof_n[4] = {0, 0, 0, 0};
};

namespace tg {

struct Timer {
    tg_ctx* c; int which; hipEvent_t a = nullptr, b = nullptr;
    Timer(tg_ctx* ctx, int w) : c(ctx), which(w) {
        if (c->profile) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, c->stream); }
    }
    ~Timer() {
        if (c->profile) { (void)hipEventRecord(b, c->stream); c->events.push_back({a, b, which}); }
    }
};

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

template <typename T, int TOPO> static void launch_step_t(tg_ctx* c, const float* d_actions) {
    const int n = c->cfg.num_envs;
    if (c->cfg.control_mode == TG_CONTROL_TCP_POSITION)
        hipLaunchKernelGGL((k_step_pos<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
    else
        hipLaunchKernelGGL((k_step<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
}
template <typename T, int TOPO> static void launch_reset_t(tg_ctx* c, const uint8_t* d_mask, int phase) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_reset<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, d_mask, phase);
}
template <typename T, int TOPO> static void launch_refresh_t(tg_ctx* c) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_refresh<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st);
}

template <typename T> static void launch_step_body_t(tg_ctx* c, const float* d_actions) {
    const int n = c->cfg.num_envs;
    if (c->cfg.control_mode == TG_CONTROL_TCP_POSITION)
        hipLaunchKernelGGL((k_step_body<T, 0, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
    else
        hipLaunchKernelGGL((k_step_body<T, 0, false>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
}
template <typename T> static void launch_reset_body_t(tg_ctx* c, const uint8_t* d_mask) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_reset_body<T, 0>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, d_mask);
}

template <typename T, int TOPO> static void launch_step_push_t(tg_ctx* c, const float* d_actions) {
    const int n = c->cfg.num_envs;
    constexpr size_t lds_bytes = (size_t)kPushLdsWords * 64 * sizeof(T);
    static bool attr_set = false;   // one context per (process, GPU): set once per instantiation
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_push<T, TOPO, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_push<T, TOPO, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    if (c->cfg.control_mode == TG_CONTROL_TCP_POSITION)
        hipLaunchKernelGGL((k_step_push<T, TOPO, true>), dim3((n + 63) / 64), dim3(64), lds_bytes, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
    else
        hipLaunchKernelGGL((k_step_push<T, TOPO, false>), dim3((n + 63) / 64), dim3(64), lds_bytes, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
}
template <typename T, int TOPO> static void launch_step_roll_t(tg_ctx* c, const float* d_actions) {
    const int n = c->cfg.num_envs;
    constexpr size_t lds_bytes = (size_t)kPushLdsWords * 64 * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_roll<T, TOPO, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_roll<T, TOPO, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    if (c->cfg.control_mode == TG_CONTROL_TCP_POSITION)
        hipLaunchKernelGGL((k_step_roll<T, TOPO, true>), dim3((n + 63) / 64), dim3(64), lds_bytes, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
    else
        hipLaunchKernelGGL((k_step_roll<T, TOPO, false>), dim3((n + 63) / 64), dim3(64), lds_bytes, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
}
template <typename T, int TOPO> static void launch_reset_roll_t(tg_ctx* c, const uint8_t* d_mask) {
    const int n = c->cfg.num_envs;
    constexpr size_t lds_bytes = (size_t)kPushLdsWords * 64 * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_reset_roll<T, TOPO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); attr_set = true; }
    hipLaunchKernelGGL((k_reset_roll<T, TOPO>), dim3((n + 63) / 64), dim3(64), lds_bytes, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, d_mask);
}
template <typename T, int TOPO> static void launch_reset_push_t(tg_ctx* c, const uint8_t* d_mask) {
    const int n = c->cfg.num_envs;
    constexpr size_t lds_bytes = (size_t)kPushLdsWords * 64 * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_reset_push<T, TOPO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); attr_set = true; }
    hipLaunchKernelGGL((k_reset_push<T, TOPO>), dim3((n + 63) / 64), dim3(64), lds_bytes, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, d_mask);
}

#define TG_DISPATCH(ctx_dtype, ctx_topo, CALL)                                               \
    do {                                                                                     \
        if ((ctx_dtype) == TG_PHYSICS_F64) {                                                 \
            if ((ctx_topo) == 0) { CALL(double, 0); } else { CALL(double, 1); }              \
        } else {                                                                             \
            if ((ctx_topo) == 0) { CALL(float, 0); } else { CALL(float, 1); }                \
        }                                                                                    \
    } while (0)

static void render(tg_ctx* c, const uint8_t* d_mask, bool save_prev) {
    Timer t(c, d_mask ? 3 : 1);
    launch_render(c->rp, c->stim, c->st.stim_xform, 1, c->cfg.num_envs, d_mask, c->d_nodef_dep, c->d_nodef_gray,
                  c->d_border, c->d_obs, save_prev ? c->d_term : nullptr, nullptr, nullptr, nullptr, c->stream);
}
// fused auto-reset: every env's observation from stim_xform; for the envs flagged in `done` also the terminal observation from term_xform
static void render_fused(tg_ctx* c) {
    Timer t(c, 1);
    launch_render(c->rp, c->stim, c->st.stim_xform, 1, c->cfg.num_envs, nullptr, c->d_nodef_dep, c->d_nodef_gray,
                  c->d_border, c->d_obs, nullptr, c->st.term_xform, c->st.done, c->d_term, c->stream);
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


// env.reset() for the masked envs: task randomisation, (surface generation), robot reset.
// tg_sample_actions: element i of draw `counter`: 24 random bits of splitmix64 over (seed, counter, i) -> lo + (hi - lo) u, u in [0, 1)
__global__ void k_sample_actions(int total, uint64_t seed, uint64_t counter, float lo, float hi, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t z = mix64(mix64(seed + kGolden * (counter + 1)) + kGolden * (uint64_t)(i + 1));
    const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
    out[i] = lo + (hi - lo) * u;
}

static void reset_sequence(tg_ctx* c, const uint8_t* d_mask) {
    Timer t(c, 2);
    if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE) {
        if (c->cfg.physics_dtype == TG_PHYSICS_F64) launch_reset_body_t<double>(c, d_mask);
        else launch_reset_body_t<float>(c, d_mask);
    } else if (c->cfg.env_kind == TG_ENV_OBJECT_ROLL) {
#define CALL(T, TOPO) launch_reset_roll_t<T, TOPO>(c, d_mask)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    } else if (c->cfg.env_kind == TG_ENV_OBJECT_PUSH) {
#define CALL(T, TOPO) launch_reset_push_t<T, TOPO>(c, d_mask)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        if (c->cfg.traj_type == TG_TRAJ_SIMPLEX)
            launch_gen_traj(c->cfg.num_envs, d_mask, c->st.noise_seed, c->cfg.traj_n_points, c->cfg.traj_spacing, c->cfg.traj_max_perturb,
                            c->cfg.traj_init_offset, c->cfg.reset_goal_id, c->st.traj, c->st.feature, c->stream);
    } else if (c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 1)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        launch_gen_surface(c->cfg.num_envs, d_mask, c->st.noise_seed, c->cfg.surf_rows, c->cfg.surf_cols, c->cfg.surf_interp,
                           c->cfg.surf_height_range, c->cfg.surf_center_z,
                           c->cfg.noise_mode == TG_SNOISE_NONE ? TG_SURF_FLAT : c->cfg.noise_mode == TG_SNOISE_RANDOM ? TG_SURF_RANDOM
                           : c->cfg.noise_mode == TG_SNOISE_VERTICAL_SIMPLEX ? TG_SURF_SIMPLEX_1D_VERT
                           : (c->cfg.movement_mode == TG_SMOVE_YZ || c->cfg.movement_mode == TG_SMOVE_YZRX) ? TG_SURF_SIMPLEX_1D : TG_SURF_SIMPLEX_2D,
                           c->st.heights, c->st.surf_zoff, c->stream);
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 2)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    } else {
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 0)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    }
}

}  // namespace tg

using namespace tg;

extern "C" {

const char* tg_last_error(void) { return g_err.c_str(); }
int tg_abi_version(void) { return TG_ABI_VERSION; }

int tg_create(const tg_config* cfg, const tg_robot* robot, const tg_sensor* sensor, const tg_mesh* stim, tg_ctx** out) {
    if (!cfg || !robot || !sensor || !out) return fail(-1, "tg_create: NULL argument");
    if (cfg->abi_version != TG_ABI_VERSION) return fail(-1, "tg_create: ABI version mismatch");
    if (cfg->env_kind != TG_ENV_EDGE_FOLLOW && cfg->env_kind != TG_ENV_SURFACE_FOLLOW_AUTO && cfg->env_kind != TG_ENV_OBJECT_BALANCE &&
        cfg->env_kind != TG_ENV_OBJECT_PUSH && cfg->env_kind != TG_ENV_OBJECT_ROLL)
        return fail(-1, "tg_create: unknown env_kind");
    if (cfg->env_kind != TG_ENV_SURFACE_FOLLOW_AUTO && !stim) return fail(-1, "tg_create: this env needs a stimulus mesh");
    if (cfg->env_kind == TG_ENV_OBJECT_BALANCE && robot->topology != 0) return fail(-1, "tg_create: object_balance is built for the UR5 chain");
    if (cfg->num_envs <= 0) return fail(-1, "tg_create: num_envs must be positive");
    if (int rc = check_robot(robot)) return rc;
    const int H = sensor->image_h, W = sensor->image_w;
    if (!((H % 128 == 0 && W % 128 == 0) || (H == 64 && W == 64))) return fail(-1, "tg_create: image size must be 64x64 or a multiple of 128");
    if (!sensor->nodef_dep || !sensor->nodef_gray || !sensor->border_mask) return fail(-1, "tg_create: sensor reference images missing");
    if (cfg->physics_dtype != TG_PHYSICS_F64 && cfg->physics_dtype != TG_PHYSICS_F32) return fail(-1, "tg_create: bad physics_dtype");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(-3, "tg_create: no HIP device visible — the tactile-env step has no CPU fallback");
    TG_HIP(hipSetDevice(cfg->device));
    tg_ctx* c = new tg_ctx();
    c->cfg = *cfg; c->robot = *robot; c->H = H; c->W = W;
    TG_HIP(hipStreamCreate(&c->own_stream));
    if (cfg->env_kind == TG_ENV_OBJECT_BALANCE) {
        TG_HIP(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
        TG_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)); TG_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    c->stream = c->own_stream;
    const int n = cfg->num_envs;
    const size_t npix = (size_t)H * W;
    if (cfg->physics_dtype == TG_PHYSICS_F64) {
        DevRobot<double> dr; EnvConst<double> ec;
        build_dev_robot(*robot, dr);
        if (int rc = build_env_const(*cfg, *sensor, *robot, ec)) { delete c; return rc; }
        c->act_dim = ec.act_dim;
        TG_HIP(hipMalloc(&c->d_robot, sizeof dr)); TG_HIP(hipMemcpy(c->d_robot, &dr, sizeof dr, hipMemcpyHostToDevice));
        TG_HIP(hipMalloc(&c->d_const, sizeof ec)); TG_HIP(hipMemcpy(c->d_const, &ec, sizeof ec, hipMemcpyHostToDevice));
    } else {
        DevRobot<float> dr; EnvConst<float> ec;
        build_dev_robot(*robot, dr);
        if (int rc = build_env_const(*cfg, *sensor, *robot, ec)) { delete c; return rc; }
        c->act_dim = ec.act_dim;
        TG_HIP(hipMalloc(&c->d_robot, sizeof dr)); TG_HIP(hipMemcpy(c->d_robot, &dr, sizeof dr, hipMemcpyHostToDevice));
        TG_HIP(hipMalloc(&c->d_const, sizeof ec)); TG_HIP(hipMemcpy(c->d_const, &ec, sizeof ec, hipMemcpyHostToDevice));
    }
    State& s = c->st;
    const size_t nd = (size_t)TG_MAX_DOF * n;
    TG_HIP(hipMalloc(&s.q, nd * 8)); TG_HIP(hipMalloc(&s.qd, nd * 8)); TG_HIP(hipMalloc(&s.qd_target, nd * 8));
    TG_HIP(hipMalloc(&s.tcp_pos, 3 * n * 8)); TG_HIP(hipMalloc(&s.tcp_rpy, 3 * n * 8));
    TG_HIP(hipMalloc(&s.edge_ang, n * 8)); TG_HIP(hipMalloc(&s.embed, n * 8));
    TG_HIP(hipMalloc(&s.stim_xform, 12 * n * 4)); TG_HIP(hipMalloc(&s.term_xform, 12 * n * 4));
    TG_HIP(hipMalloc(&s.step_count, n * 4)); TG_HIP(hipMalloc(&s.reset_ticks, n * 4)); TG_HIP(hipMalloc(&s.licence, n * 4));
    TG_HIP(hipMalloc(&s.trig_sc, (size_t)16 * n * 8)); TG_HIP(hipMalloc(&s.edge_sc, (size_t)2 * n * 8));
    TG_HIP(hipMemset(s.trig_sc, 0, (size_t)16 * n * 8)); TG_HIP(hipMemset(s.edge_sc, 0, (size_t)2 * n * 8));
    TG_HIP(hipMalloc(&s.rng, n * 8));
    TG_HIP(hipMemset(s.q, 0, nd * 8)); TG_HIP(hipMemset(s.qd, 0, nd * 8)); TG_HIP(hipMemset(s.qd_target, 0, nd * 8));
    TG_HIP(hipMemset(s.tcp_pos, 0, 3 * n * 8)); TG_HIP(hipMemset(s.tcp_rpy, 0, 3 * n * 8));
    TG_HIP(hipMemset(s.edge_ang, 0, n * 8)); TG_HIP(hipMemset(s.embed, 0, n * 8));
    TG_HIP(hipMemset(s.stim_xform, 0, 12 * n * 4)); TG_HIP(hipMemset(s.term_xform, 0, 12 * n * 4));
    TG_HIP(hipMemset(s.step_count, 0, n * 4)); TG_HIP(hipMemset(s.reset_ticks, 0, n * 4)); TG_HIP(hipMemset(s.licence, 0, n * 4));
    std::vector<uint64_t> seeds(n);
    for (int i = 0; i < n; ++i) seeds[i] = mix64((uint64_t)i + kGolden);
    TG_HIP(hipMemcpy(s.rng, seeds.data(), n * 8, hipMemcpyHostToDevice));
    TG_HIP(hipMalloc(&c->d_nodef_dep, npix * 4)); TG_HIP(hipMemcpy(c->d_nodef_dep, sensor->nodef_dep, npix * 4, hipMemcpyHostToDevice));
    {
        std::vector<uint8_t> g8(npix);
        make_gray_u8(sensor->nodef_gray, (int)npix, g8.data());
        TG_HIP(hipMalloc(&c->d_nodef_gray, npix)); TG_HIP(hipMemcpy(c->d_nodef_gray, g8.data(), npix, hipMemcpyHostToDevice));
    }
    TG_HIP(hipMalloc(&c->d_border, npix)); TG_HIP(hipMemcpy(c->d_border, sensor->border_mask, npix, hipMemcpyHostToDevice));
    if (cfg->env_kind == TG_ENV_OBJECT_BALANCE) {
        TG_HIP(hipMalloc(&s.body_pos, 3 * n * 8)); TG_HIP(hipMalloc(&s.body_rot, 9 * n * 8)); TG_HIP(hipMalloc(&s.body_v, 3 * n * 8));
        TG_HIP(hipMalloc(&s.body_w, 3 * n * 8)); TG_HIP(hipMalloc(&s.ext_pos, 3 * n * 8)); TG_HIP(hipMalloc(&s.gravity, n * 8));
        TG_HIP(hipMalloc(&s.ext_pending, n));
        TG_HIP(hipMemset(s.body_v, 0, 3 * n * 8)); TG_HIP(hipMemset(s.body_w, 0, 3 * n * 8)); TG_HIP(hipMemset(s.ext_pos, 0, 3 * n * 8));
        TG_HIP(hipMemset(s.ext_pending, 0, n));
        // load_object (base_object_env.py:66-70): loadURDF puts the object's *link* frame at init_obj_pos; the inertial frame used by
        // get/resetBasePositionAndOrientation is obj_root_inertial_pos away.  setup_object (:185-190): default embed distance.
        double oq[4], oR[9];
        h_quat_from_euler(cfg->obj_init_rpy, oq);
        h_mat_from_quat(oq, oR);
        double p0[3] = {cfg->workframe_pos[0], cfg->workframe_pos[1], cfg->workframe_pos[2] + cfg->obj_base_height / 2 - cfg->embed_dist};
        for (int a = 0; a < 3; ++a)
            p0[a] += oR[3 * a] * cfg->obj_root_inertial_pos[0] + oR[3 * a + 1] * cfg->obj_root_inertial_pos[1] + oR[3 * a + 2] * cfg->obj_root_inertial_pos[2];
        std::vector<double> bp(3 * (size_t)n), br(9 * (size_t)n), gz(n, cfg->gravity_default), em(n, cfg->embed_dist);
        for (int i = 0; i < n; ++i) {
            for (int a = 0; a < 3; ++a) bp[(size_t)a * n + i] = p0[a];
            for (int a = 0; a < 9; ++a) br[(size_t)a * n + i] = oR[a];
        }
        TG_HIP(hipMemcpy(s.body_pos, bp.data(), bp.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.body_rot, br.data(), br.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.gravity, gz.data(), gz.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.embed, em.data(), em.size() * 8, hipMemcpyHostToDevice));
    }
    TG_HIP(hipMalloc(&s.contact_code, (size_t)n * 4)); TG_HIP(hipMemset(s.contact_code, 0, (size_t)n * 4));
    if (cfg->env_kind == TG_ENV_OBJECT_ROLL) {
        TG_HIP(hipMalloc(&s.body_pos, 3 * n * 8)); TG_HIP(hipMalloc(&s.body_rot, 9 * n * 8)); TG_HIP(hipMalloc(&s.body_v, 3 * n * 8));
        TG_HIP(hipMalloc(&s.body_w, 3 * n * 8)); TG_HIP(hipMalloc(&s.obj_mass, n * 8)); TG_HIP(hipMalloc(&s.goal, 3 * n * 8));
        TG_HIP(hipMalloc(&s.term_feature, (size_t)12 * n * 4));
        TG_HIP(hipMemset(s.body_v, 0, 3 * n * 8)); TG_HIP(hipMemset(s.body_w, 0, 3 * n * 8)); TG_HIP(hipMemset(s.goal, 0, 3 * n * 8));
        TG_HIP(hipMemset(s.term_feature, 0, (size_t)12 * n * 4));
        // load_object (base_object_env.py:66-70) at init_obj_pos, identity orientation, default radius (object_roll_env.py:156-166)
        std::vector<double> bp(3 * (size_t)n), br(9 * (size_t)n, 0.0), rad(n, cfg->roll_radius), em(n, cfg->embed_dist);
        for (int i = 0; i < n; ++i) {
            for (int a = 0; a < 3; ++a) bp[(size_t)a * n + i] = cfg->obj_init_pos[a];
            br[(size_t)0 * n + i] = 1.0; br[(size_t)4 * n + i] = 1.0; br[(size_t)8 * n + i] = 1.0;
        }
        TG_HIP(hipMemcpy(s.body_pos, bp.data(), bp.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.body_rot, br.data(), br.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.obj_mass, rad.data(), rad.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.embed, em.data(), em.size() * 8, hipMemcpyHostToDevice));
    }
    if (cfg->env_kind == TG_ENV_OBJECT_PUSH) {
        TG_HIP(hipMalloc(&s.body_pos, 3 * n * 8)); TG_HIP(hipMalloc(&s.body_rot, 9 * n * 8)); TG_HIP(hipMalloc(&s.body_v, 3 * n * 8));
        TG_HIP(hipMalloc(&s.body_w, 3 * n * 8)); TG_HIP(hipMalloc(&s.noise_seed, n * 8)); TG_HIP(hipMalloc(&s.obj_mass, n * 8));
        TG_HIP(hipMalloc(&s.traj, (size_t)3 * TG_MAX_TRAJ_POINTS * n * 8)); TG_HIP(hipMalloc(&s.goal_id, n * 4));
        TG_HIP(hipMalloc(&s.term_feature, (size_t)12 * n * 4));
        TG_HIP(hipMemset(s.body_v, 0, 3 * n * 8)); TG_HIP(hipMemset(s.body_w, 0, 3 * n * 8)); TG_HIP(hipMemset(s.noise_seed, 0, n * 8));
        TG_HIP(hipMemset(s.traj, 0, (size_t)3 * TG_MAX_TRAJ_POINTS * n * 8)); TG_HIP(hipMemset(s.goal_id, 0, n * 4));
        TG_HIP(hipMemset(s.term_feature, 0, (size_t)12 * n * 4));
        // load_object (base_object_env.py:66-70) at init_obj_pos / init_obj_orn (object_push_env.py:154-160)
        double oq[4], oR[9];
        h_quat_from_euler(cfg->obj_init_rpy, oq);
        h_mat_from_quat(oq, oR);
        std::vector<double> bp(3 * (size_t)n), br(9 * (size_t)n), ms(n, cfg->obj_mass);
        for (int i = 0; i < n; ++i) {
            for (int a = 0; a < 3; ++a) bp[(size_t)a * n + i] = cfg->obj_init_pos[a];
            for (int a = 0; a < 9; ++a) br[(size_t)a * n + i] = oR[a];
        }
        TG_HIP(hipMemcpy(s.body_pos, bp.data(), bp.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.body_rot, br.data(), br.size() * 8, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(s.obj_mass, ms.data(), ms.size() * 8, hipMemcpyHostToDevice));
        const size_t nv = (size_t)cfg->n_tip_verts * 3;
        void* dv = nullptr;
        if (cfg->physics_dtype == TG_PHYSICS_F64) {
            TG_HIP(hipMalloc(&dv, nv * 8)); TG_HIP(hipMemcpy(dv, cfg->tip_verts, nv * 8, hipMemcpyHostToDevice));
        } else {
            std::vector<float> vf(nv);
            for (size_t k = 0; k < nv; ++k) vf[k] = (float)cfg->tip_verts[k];
            TG_HIP(hipMalloc(&dv, nv * 4)); TG_HIP(hipMemcpy(dv, vf.data(), nv * 4, hipMemcpyHostToDevice));
        }
        s.tip_verts = dv;
        c->cfg.tip_verts = nullptr;   // the host pointer is not kept
    }
    if (cfg->env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        const size_t cells = (size_t)cfg->surf_rows * cfg->surf_cols;
        TG_HIP(hipMalloc(&s.dir, 2 * n * 8)); TG_HIP(hipMalloc(&s.goal, 3 * n * 8)); TG_HIP(hipMalloc(&s.heights, cells * n * 8));
        TG_HIP(hipMalloc(&s.surf_zoff, n * 4)); TG_HIP(hipMalloc(&s.noise_seed, n * 8)); TG_HIP(hipMalloc(&s.accum, n * 8));
        TG_HIP(hipMemset(s.dir, 0, 2 * n * 8)); TG_HIP(hipMemset(s.goal, 0, 3 * n * 8)); TG_HIP(hipMemset(s.heights, 0, cells * n * 8));
        TG_HIP(hipMemset(s.surf_zoff, 0, n * 4)); TG_HIP(hipMemset(s.noise_seed, 0, n * 8)); TG_HIP(hipMemset(s.accum, 0, n * 8));
        c->stim.kind = 1; c->stim.heights = s.heights; c->stim.zoff = s.surf_zoff;
        c->stim.rows = cfg->surf_rows; c->stim.cols = cfg->surf_cols; c->stim.scale = (float)cfg->surf_grid_scale;
        c->stim.n_tris = (cfg->surf_rows - 1) * (cfg->surf_cols - 1) * 2;
    } else {
        TG_HIP(hipMalloc(&c->d_verts, (size_t)stim->n_verts * 12)); TG_HIP(hipMemcpy(c->d_verts, stim->verts, (size_t)stim->n_verts * 12, hipMemcpyHostToDevice));
        TG_HIP(hipMalloc(&c->d_tris, (size_t)stim->n_tris * 12)); TG_HIP(hipMemcpy(c->d_tris, stim->tris, (size_t)stim->n_tris * 12, hipMemcpyHostToDevice));
        c->n_tris = stim->n_tris;
        {
            std::vector<float> soup((size_t)stim->n_tris * 9);
            for (int t = 0; t < stim->n_tris; ++t)
                for (int k = 0; k < 3; ++k)
                    for (int a = 0; a < 3; ++a) soup[(size_t)t * 9 + 3 * k + a] = stim->verts[3 * (size_t)stim->tris[3 * t + k] + a];
            TG_HIP(hipMalloc(&c->d_soup, soup.size() * 4 + 4)); TG_HIP(hipMemcpy(c->d_soup, soup.data(), soup.size() * 4, hipMemcpyHostToDevice));
        }
        c->stim.skip_quad_reject = cfg->env_kind == TG_ENV_OBJECT_BALANCE ? 1 : 0;   // the plate fills the camera's view
        c->stim.kind = 0; c->stim.verts = c->d_verts; c->stim.tris = c->d_tris; c->stim.soup = c->d_soup; c->stim.n_tris = stim->n_tris;
    }
    // one allocation [tactile obs u8 | reward f32 | done u8]: what a rank ships to rank 0 per step is one contiguous byte range
    // (tg_get_packed_outputs).  The obs block is padded to 16 bytes so the reward block stays aligned.
    c->packed_obs_bytes = ((size_t)npix * n + 15) & ~(size_t)15;
    c->packed_bytes = c->packed_obs_bytes + (size_t)n * 4 + (size_t)n;
    const bool has_feature = cfg->env_kind == TG_ENV_OBJECT_PUSH || cfg->env_kind == TG_ENV_OBJECT_ROLL;
    if (has_feature) {   // extended_feature rides in the same message (config 4's tactile_and_feature observation)
        c->packed_feature_off = (c->packed_bytes + 3) & ~(size_t)3;
        c->packed_bytes = c->packed_feature_off + (size_t)n * 12 * 4;
    }
    TG_HIP(hipMalloc(&c->d_obs, c->packed_bytes)); TG_HIP(hipMemset(c->d_obs, 0, c->packed_bytes));
    s.reward = (float*)(c->d_obs + c->packed_obs_bytes);
    s.done = c->d_obs + c->packed_obs_bytes + (size_t)n * 4;
    if (has_feature) s.feature = (float*)(c->d_obs + c->packed_feature_off);
    TG_HIP(hipMalloc(&c->d_term, npix * n)); TG_HIP(hipMemset(c->d_term, 0, npix * n));
    TG_HIP(hipMalloc(&c->d_mask, n));
    TG_HIP(hipMalloc(&c->d_actions, (size_t)n * 6 * sizeof(float)));
    c->rp = make_raster_params(W, H, sensor->fov_deg, sensor->near_plane, sensor->far_plane, sensor->turn_off_border, sensor->nodef_dep);
    *out = c;
    return 0;
}

int tg_destroy(tg_ctx* c) {
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    drain_events(c);
    if (c->step_graph) (void)hipGraphExecDestroy(c->step_graph);
    State& s = c->st;
    void* ptrs[] = {c->d_robot, c->d_const, s.q, s.qd, s.qd_target, s.tcp_pos, s.tcp_rpy, s.edge_ang, s.embed, s.stim_xform, s.term_xform,
                    s.step_count, s.reset_ticks, s.licence, s.trig_sc, s.edge_sc, s.rng, s.dir, s.goal, s.heights, s.accum, s.surf_zoff, s.noise_seed, s.body_pos, s.body_rot, s.body_v, s.body_w, s.ext_pos, s.gravity, s.ext_pending, s.traj, s.obj_mass, s.goal_id, s.contact_code, s.term_feature, const_cast<void*>(s.tip_verts), c->d_nodef_dep, c->d_nodef_gray, c->d_border, c->d_verts, c->d_soup, c->d_tris,
                    c->d_obs, c->d_term, c->d_mask, c->d_actions};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

int tg_set_stream(tg_ctx* c, void* s) {
    if (!c) return fail(-1, "NULL ctx");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return 0;
}

int tg_seed(tg_ctx* c, const uint64_t* seeds, int32_t n) {
    if (!c || !seeds) return fail(-1, "tg_seed: NULL argument");
    if (n != c->cfg.num_envs) return fail(-1, "tg_seed: need one seed per env");
    std::vector<uint64_t> st(n);
    for (int i = 0; i < n; ++i) st[i] = mix64(seeds[i] + kGolden);
    TG_HIP(hipMemcpyAsync(c->st.rng, st.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_reset(tg_ctx* c, const uint8_t* host_mask) {
    if (!c) return fail(-1, "NULL ctx");
    const uint8_t* dmask = nullptr;
    if (host_mask) {
        TG_HIP(hipMemcpyAsync(c->d_mask, host_mask, c->cfg.num_envs, hipMemcpyHostToDevice, c->stream));
        dmask = c->d_mask;
    }
    reset_sequence(c, dmask);
    render(c, dmask, false);
    TG_HIP(hipGetLastError());
    return 0;
}

static void enqueue_step(tg_ctx* c, const float* d_act) {
    {
        Timer t(c, 0);
        if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE) {
            if (c->cfg.physics_dtype == TG_PHYSICS_F64) launch_step_body_t<double>(c, d_act);
            else launch_step_body_t<float>(c, d_act);
        } else if (c->cfg.env_kind == TG_ENV_OBJECT_ROLL) {
#define CALL(T, TOPO) launch_step_roll_t<T, TOPO>(c, d_act)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        } else if (c->cfg.env_kind == TG_ENV_OBJECT_PUSH) {
#define CALL(T, TOPO) launch_step_push_t<T, TOPO>(c, d_act)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        } else {
#define CALL(T, TOPO) launch_step_t<T, TOPO>(c, d_act)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        }
    }
    if (c->cfg.auto_reset && c->cfg.env_kind == TG_ENV_EDGE_FOLLOW) {
        reset_sequence(c, c->st.done); // k_reset keeps the terminal camera transform of the envs it resets
        render_fused(c);               // one launch draws the terminal and the post-reset observations
    } else if (c->cfg.auto_reset && c->aux_stream) {
        // object_balance: a pole falls somewhere in the batch on nearly every step, and its reset (rest pose, blocking move, settling: a
        // serial chain of 0.14 ms on a few wavefronts) depends on k_step only.  Fork: it runs on the second stream while this one draws the
        // step's observations from term_xform (k_step's second copy of the transforms, which the reset does not touch); join; then the
        // masked launch saves the terminal observations and draws the post-reset ones.  Inside the captured graph this is a fork/join.
        (void)hipEventRecord(c->ev_fork, c->stream);
        (void)hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0);
        {
            hipStream_t main_stream = c->stream;
            c->stream = c->aux_stream;
            reset_sequence(c, c->st.done);
            c->stream = main_stream;
        }
        (void)hipEventRecord(c->ev_join, c->aux_stream);
        {
            Timer t(c, 1);
            launch_render(c->rp, c->stim, c->st.term_xform, 1, c->cfg.num_envs, nullptr, c->d_nodef_dep, c->d_nodef_gray, c->d_border, c->d_obs,
                          nullptr, nullptr, nullptr, nullptr, c->stream);
        }
        (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
        render(c, c->st.done, true);
    } else {
        render(c, nullptr, false);
        if (c->cfg.auto_reset) {
            reset_sequence(c, c->st.done);
            render(c, c->st.done, true);   // terminal observation is saved, then the post-reset observation is drawn
        }
    }
}

int tg_step(tg_ctx* c, const float* actions, int32_t on_device) {
    if (!c || !actions) return fail(-1, "tg_step: NULL argument");
    const float* d_act = actions;
    if (!on_device) {
        TG_HIP(hipMemcpyAsync(c->d_actions, actions, (size_t)c->cfg.num_envs * c->act_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_act = c->d_actions;
    }
    // The launch sequence of a step is the same every step (all arguments are device pointers owned by the context, the action
    // buffer aside): capture it once per action pointer / stream and replay it as one graph launch.  Not while profiling (the
    // per-kernel events are host calls between the launches) and not for the push kernels (hipFuncSetAttribute on first launch).
    const bool want_graph = !c->profile && !c->graph_broken && c->cfg.env_kind != TG_ENV_OBJECT_PUSH && c->cfg.env_kind != TG_ENV_OBJECT_ROLL;
    if (want_graph) {
        if (c->step_graph && (c->step_graph_actions != d_act || c->step_graph_stream != c->stream)) {
            (void)hipGraphExecDestroy(c->step_graph);
            c->step_graph = nullptr;
        }
        if (!c->step_graph) {
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                enqueue_step(c, d_act);
                const hipError_t e1 = hipStreamEndCapture(c->stream, &g);
                if (e1 == hipSuccess && g && hipGraphInstantiate(&c->step_graph, g, nullptr, nullptr, 0) == hipSuccess) {
                    c->step_graph_actions = d_act; c->step_graph_stream = c->stream;
                } else {
                    c->step_graph = nullptr; c->graph_broken = true;
                }
                if (g) (void)hipGraphDestroy(g);
            } else {
                c->graph_broken = true;
            }
            (void)hipGetLastError();
        }
        if (c->step_graph) {
            TG_HIP(hipGraphLaunch(c->step_graph, c->stream));
            return 0;
        }
    }
    enqueue_step(c, d_act);
    TG_HIP(hipGetLastError());
    return 0;
}

int tg_sync(tg_ctx* c) {
    if (!c) return fail(-1, "NULL ctx");
    // A step is ~0.15 ms: the interrupt-driven wake-up of hipStreamSynchronize costs a noticeable fraction of it.  Poll for up to
    // ~2 ms (VecEnv.step_wait follows step_async immediately), then fall back to the blocking wait.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail(-2, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_get_obs_tactile(tg_ctx* c, void** p) { if (!c || !p) return fail(-1, "NULL argument"); *p = c->d_obs; return 0; }
int tg_get_terminal_obs(tg_ctx* c, void** p) { if (!c || !p) return fail(-1, "NULL argument"); *p = c->d_term; return 0; }
int tg_get_reward_done_dev(tg_ctx* c, void** r, void** d) {
    if (!c) return fail(-1, "NULL ctx");
    if (r) *r = c->st.reward;
    if (d) *d = c->st.done;
    return 0;
}
int tg_get_packed_outputs(tg_ctx* c, void** p, int64_t* obs_bytes, int64_t* total_bytes) {
    if (!c || !p) return fail(-1, "NULL argument");
    *p = c->d_obs;
    if (obs_bytes) *obs_bytes = (int64_t)c->packed_obs_bytes;
    if (total_bytes) *total_bytes = (int64_t)c->packed_bytes;
    return 0;
}
int tg_get_packed_feature(tg_ctx* c, int64_t* feature_off, int32_t* dim) {
    if (!c || !feature_off) return fail(-1, "NULL argument");
    const bool has = c->cfg.env_kind == TG_ENV_OBJECT_PUSH || c->cfg.env_kind == TG_ENV_OBJECT_ROLL;
    *feature_off = has ? (int64_t)c->packed_feature_off : -1;
    if (dim) *dim = has ? 12 : 0;
    return 0;
}
int tg_sample_actions(tg_ctx* c, uint64_t seed, uint64_t counter, float* dev_actions) {
    if (!c || !dev_actions) return fail(-1, "tg_sample_actions: NULL argument");
    const int total = c->cfg.num_envs * c->act_dim;
    hipLaunchKernelGGL(k_sample_actions, dim3((total + 255) / 256), dim3(256), 0, c->stream, total, seed, counter, (float)c->cfg.min_action,
                       (float)c->cfg.max_action, dev_actions);
    TG_HIP(hipGetLastError());
    return 0;
}
int tg_selftest_division(int64_t n, uint64_t seed, int64_t* mismatches) {
    if (!mismatches || n < 0) return fail(-1, "tg_selftest_division: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(-2, "no HIP device");
    long long m = 0;
    if (tg::selftest_division((long long)n, (unsigned long long)seed, &m) != 0) return fail(-3, "tg_selftest_division: launch failed");
    *mismatches = (int64_t)m;
    return 0;
}
int tg_get_obs_feature(tg_ctx* c, void** p, int32_t* dim, int32_t terminal) {
    if (!c || !p) return fail(-1, "NULL argument");
    if (c->cfg.env_kind != TG_ENV_OBJECT_PUSH && c->cfg.env_kind != TG_ENV_OBJECT_ROLL)
        return fail(-1, "tg_get_obs_feature: this env has no extended_feature observation");
    *p = terminal ? c->st.term_feature : c->st.feature;
    if (dim) *dim = 12;
    return 0;
}
int tg_copy_obs_feature(tg_ctx* c, float* dst, int32_t terminal) {
    if (!c || !dst) return fail(-1, "NULL argument");
    if (c->cfg.env_kind != TG_ENV_OBJECT_PUSH && c->cfg.env_kind != TG_ENV_OBJECT_ROLL)
        return fail(-1, "tg_copy_obs_feature: this env has no extended_feature observation");
    TG_HIP(hipMemcpyAsync(dst, terminal ? c->st.term_feature : c->st.feature, (size_t)c->cfg.num_envs * 12 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int tg_get_reward_done(tg_ctx* c, float* reward, uint8_t* done) {
    if (!c) return fail(-1, "NULL ctx");
    const int n = c->cfg.num_envs;
    if (reward) TG_HIP(hipMemcpyAsync(reward, c->st.reward, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (done) TG_HIP(hipMemcpyAsync(done, c->st.done, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int tg_copy_obs_tactile(tg_ctx* c, uint8_t* dst, int32_t terminal) {
    if (!c || !dst) return fail(-1, "NULL argument");
    TG_HIP(hipMemcpyAsync(dst, terminal ? c->d_term : c->d_obs, (size_t)c->cfg.num_envs * c->H * c->W, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_get_state(tg_ctx* c, const tg_state_view* v) {
    if (!c || !v) return fail(-1, "NULL argument");
    const int nd = c->robot.ndof;
    int rc = 0;
    if (v->q && (rc = fetch_soa(c, c->st.q, nd, v->q))) return rc;
    if (v->qd && (rc = fetch_soa(c, c->st.qd, nd, v->qd))) return rc;
    if (v->qd_target && (rc = fetch_soa(c, c->st.qd_target, nd, v->qd_target))) return rc;
    if (v->tcp_rpy && c->cfg.env_kind == TG_ENV_EDGE_FOLLOW) {   // the step kernel leaves this read-back to be recomputed on demand
#define CALL(T, TOPO) launch_refresh_t<T, TOPO>(c)
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
    if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE) {
        if (v->body_pos && (rc = fetch_soa(c, c->st.body_pos, 3, v->body_pos))) return rc;
        if (v->body_rot && (rc = fetch_soa(c, c->st.body_rot, 9, v->body_rot))) return rc;
        if (v->body_linvel && (rc = fetch_soa(c, c->st.body_v, 3, v->body_linvel))) return rc;
        if (v->body_angvel && (rc = fetch_soa(c, c->st.body_w, 3, v->body_angvel))) return rc;
        if (v->gravity_z && (rc = fetch_soa(c, c->st.gravity, 1, v->gravity_z))) return rc;
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
            int32_t ids[5] = {-1, -1, -1, -1, -1};
            for (int b = 0; b < 8 && cnt < 4; ++b) if ((code[i] >> b) & 1) ids[cnt++] = b;
            if ((code[i] >> 8) & 1) ids[cnt++] = 8 + (code[i] >> 9);
            if (v->contact_count) v->contact_count[i] = cnt;
            if (v->contact_ids) for (int k = 0; k < 5; ++k) v->contact_ids[(size_t)i * 5 + k] = ids[k];
        }
    }
    if (c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        if (v->goal_pos && (rc = fetch_soa(c, c->st.goal, 3, v->goal_pos))) return rc;
        if (v->direction && (rc = fetch_soa(c, c->st.dir, 2, v->direction))) return rc;
        if (v->surf_zoff && (rc = fetch_soa(c, c->st.surf_zoff, 1, v->surf_zoff))) return rc;
        if (v->heights) {
            TG_HIP(hipMemcpyAsync(v->heights, c->st.heights, (size_t)c->cfg.num_envs * c->cfg.surf_rows * c->cfg.surf_cols * 8, hipMemcpyDeviceToHost, c->stream));
            TG_HIP(hipStreamSynchronize(c->stream));
        }
    }
    return 0;
}

int tg_set_joint_state(tg_ctx* c, const double* q, const double* qd) {
    if (!c || !q || !qd) return fail(-1, "NULL argument");
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
    render(c, nullptr, false);
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_profile_enable(tg_ctx* c, int32_t enable) {
    if (!c) return fail(-1, "NULL ctx");
    (void)hipStreamSynchronize(c->stream);
    drain_events(c);
    c->profile = enable != 0;
    for (int k = 0; k < 4; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    return 0;
}
int tg_profile_get(tg_ctx* c, int32_t which, double* total_ms, int64_t* launches) {
    if (!c || which < 0 || which > 3) return fail(-1, "bad argument");
    (void)hipStreamSynchronize(c->stream);
    drain_events(c);
    if (total_ms) *total_ms = c->prof_ms[which];
    if (launches) *launches = c->prof_n[which];
    return 0;
}

// ---------------------------------------------------------------------------------------------------- function-level entry points
#define TG_FN_DISPATCH(robot, dtype, KERNEL, n, ...)                                                                              \
    do {                                                                                                                          \
        DevBuf rb;                                                                                                                \
        dim3 grid(((n) + 63) / 64), block(64);                                                                                    \
        if ((dtype) == TG_PHYSICS_F64) {                                                                                          \
            if (int rc = upload_robot<double>(robot, rb)) return rc;                                                              \
            if ((robot)->topology == 0) hipLaunchKernelGGL((KERNEL<double, 0>), grid, block, 0, 0, (const DevRobot<double>*)rb.p, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<double, 1>), grid, block, 0, 0, (const DevRobot<double>*)rb.p, __VA_ARGS__);          \
        } else {                                                                                                                  \
            if (int rc = upload_robot<float>(robot, rb)) return rc;                                                               \
            if ((robot)->topology == 0) hipLaunchKernelGGL((KERNEL<float, 0>), grid, block, 0, 0, (const DevRobot<float>*)rb.p, __VA_ARGS__);   \
            else hipLaunchKernelGGL((KERNEL<float, 1>), grid, block, 0, 0, (const DevRobot<float>*)rb.p, __VA_ARGS__);            \
        }                                                                                                                         \
        TG_HIP(hipDeviceSynchronize());                                                                                           \
    } while (0)

int tg_inverse_dynamics(const tg_robot* robot, int32_t dtype, int32_t n, const double* q, const double* qd, const double* qdd, double* tau) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const size_t bytes = (size_t)n * robot->ndof * 8;
    DevBuf a, b, c, d;
    if (a.alloc(bytes) || b.alloc(bytes) || c.alloc(bytes) || d.alloc(bytes)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, bytes, hipMemcpyHostToDevice)); TG_HIP(hipMemcpy(b.p, qd, bytes, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(c.p, qdd, bytes, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_inverse_dynamics, n, n, (const double*)a.p, (const double*)b.p, (const double*)c.p, (double*)d.p);
    TG_HIP(hipMemcpy(tau, d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int tg_mass_matrix(const tg_robot* robot, int32_t dtype, int32_t n, const double* q, double* M) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const int nd = robot->ndof;
    DevBuf a, b;
    if (a.alloc((size_t)n * nd * 8) || b.alloc((size_t)n * nd * nd * 8)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, (size_t)n * nd * 8, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_mass_matrix, n, n, (const double*)a.p, (double*)b.p);
    TG_HIP(hipMemcpy(M, b.p, (size_t)n * nd * nd * 8, hipMemcpyDeviceToHost));
    return 0;
}

int tg_jacobian_tcp(const tg_robot* robot, int32_t dtype, int32_t n, const double* q, double* J, double* pos, double* rot) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const int nd = robot->ndof;
    DevBuf a, b, c, d;
    if (a.alloc((size_t)n * nd * 8) || b.alloc((size_t)n * 6 * nd * 8) || c.alloc((size_t)n * 24) || d.alloc((size_t)n * 72)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, (size_t)n * nd * 8, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_jacobian, n, n, (const double*)a.p, (double*)b.p, (double*)c.p, (double*)d.p);
    if (J) TG_HIP(hipMemcpy(J, b.p, (size_t)n * 6 * nd * 8, hipMemcpyDeviceToHost));
    if (pos) TG_HIP(hipMemcpy(pos, c.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (rot) TG_HIP(hipMemcpy(rot, d.p, (size_t)n * 72, hipMemcpyDeviceToHost));
    return 0;
}

int tg_sim_ticks(const tg_robot* robot, int32_t dtype, int32_t n, int32_t n_ticks, int32_t iters, double dt, int32_t motor_mode,
                 const double* q_des, const double* qd_des, double max_force, double* q, double* qd) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const size_t bytes = (size_t)n * robot->ndof * 8;
    DevBuf a, b, c, d;
    if (a.alloc(bytes) || b.alloc(bytes) || c.alloc(bytes) || d.alloc(bytes)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q, bytes, hipMemcpyHostToDevice)); TG_HIP(hipMemcpy(b.p, qd, bytes, hipMemcpyHostToDevice));
    if (q_des) TG_HIP(hipMemcpy(c.p, q_des, bytes, hipMemcpyHostToDevice));
    if (qd_des) TG_HIP(hipMemcpy(d.p, qd_des, bytes, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_sim_ticks, n, n, n_ticks, iters, dt, motor_mode, q_des ? (const double*)c.p : (const double*)nullptr,
                   qd_des ? (const double*)d.p : (const double*)nullptr, max_force, (double*)a.p, (double*)b.p);
    TG_HIP(hipMemcpy(q, a.p, bytes, hipMemcpyDeviceToHost)); TG_HIP(hipMemcpy(qd, b.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int tg_inverse_kinematics(const tg_robot* robot, int32_t dtype, int32_t n, const double* q0, const double* target_pos, const double* target_rot,
                          int32_t max_iters, double threshold, double* q_out, int32_t* iters) {
    if (int rc = check_robot(robot)) return rc;
    if (int rc = need_device()) return rc;
    const int nd = robot->ndof;
    DevBuf a, b, c, d, e;
    if (a.alloc((size_t)n * nd * 8) || b.alloc((size_t)n * 24) || c.alloc((size_t)n * 72) || d.alloc((size_t)n * nd * 8) || e.alloc((size_t)n * 4))
        return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(a.p, q0, (size_t)n * nd * 8, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(b.p, target_pos, (size_t)n * 24, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(c.p, target_rot, (size_t)n * 72, hipMemcpyHostToDevice));
    TG_FN_DISPATCH(robot, dtype, k_ik, n, n, (const double*)a.p, (const double*)b.p, (const double*)c.p, max_iters, threshold, (double*)d.p,
                   (int32_t*)e.p);
    TG_HIP(hipMemcpy(q_out, d.p, (size_t)n * nd * 8, hipMemcpyDeviceToHost));
    if (iters) TG_HIP(hipMemcpy(iters, e.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int tg_render_tactile(const tg_sensor* sen, const tg_mesh* mesh, int32_t n, const float* xf, uint8_t* out) {
    if (!sen || !mesh || !xf || !out) return fail(-1, "NULL argument");
    if (int rc = need_device()) return rc;
    const int H = sen->image_h, W = sen->image_w;
    if (!((H % 128 == 0 && W % 128 == 0) || (H == 64 && W == 64))) return fail(-1, "image size must be 64x64 or a multiple of 128");
    const size_t npix = (size_t)H * W;
    DevBuf nd, ng, bm, vv, tt, xx, oo;
    if (nd.alloc(npix * 4) || ng.alloc(npix * 4) || bm.alloc(npix) || vv.alloc((size_t)mesh->n_verts * 12) || tt.alloc((size_t)mesh->n_tris * 12) ||
        xx.alloc((size_t)n * 48) || oo.alloc(npix * n))
        return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(nd.p, sen->nodef_dep, npix * 4, hipMemcpyHostToDevice));
    { std::vector<uint8_t> g8(npix); make_gray_u8(sen->nodef_gray, (int)npix, g8.data()); TG_HIP(hipMemcpy(ng.p, g8.data(), npix, hipMemcpyHostToDevice)); }
    TG_HIP(hipMemcpy(bm.p, sen->border_mask, npix, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(vv.p, mesh->verts, (size_t)mesh->n_verts * 12, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(tt.p, mesh->tris, (size_t)mesh->n_tris * 12, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(xx.p, xf, (size_t)n * 48, hipMemcpyHostToDevice));
    TG_HIP(hipMemset(oo.p, 0, npix * n));
    RasterParams P = make_raster_params(W, H, sen->fov_deg, sen->near_plane, sen->far_plane, sen->turn_off_border, sen->nodef_dep);
    Stimulus S{};
    DevBuf sp;
    {
        std::vector<float> soup((size_t)mesh->n_tris * 9);
        for (int t = 0; t < mesh->n_tris; ++t)
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 3; ++a) soup[(size_t)t * 9 + 3 * k + a] = mesh->verts[3 * (size_t)mesh->tris[3 * t + k] + a];
        if (sp.alloc(soup.size() * 4 + 4)) return fail(-2, "hipMalloc failed");
        TG_HIP(hipMemcpy(sp.p, soup.data(), soup.size() * 4, hipMemcpyHostToDevice));
    }
    S.kind = 0; S.verts = (const float*)vv.p; S.tris = (const int32_t*)tt.p; S.soup = (const float*)sp.p; S.n_tris = mesh->n_tris;
    launch_render(P, S, (const float*)xx.p, 0, n, nullptr, (const float*)nd.p, (const uint8_t*)ng.p, (const uint8_t*)bm.p, (uint8_t*)oo.p, nullptr, nullptr, nullptr, nullptr, 0);
    TG_HIP(hipDeviceSynchronize());
    TG_HIP(hipMemcpy(out, oo.p, npix * n, hipMemcpyDeviceToHost));
    return 0;
}

int tg_render_tactile_heightfield(const tg_sensor* sen, int32_t rows, int32_t cols, double grid_scale, int32_t n, const double* heights,
                                  const float* zoff, const float* xf, uint8_t* out) {
    if (!sen || !heights || !zoff || !xf || !out) return fail(-1, "NULL argument");
    if (int rc = need_device()) return rc;
    const int H = sen->image_h, W = sen->image_w;
    if (!((H % 128 == 0 && W % 128 == 0) || (H == 64 && W == 64))) return fail(-1, "image size must be 64x64 or a multiple of 128");
    const size_t npix = (size_t)H * W, cells = (size_t)rows * cols;
    DevBuf nd, ng, bm, hh, zz, xx, oo;
    if (nd.alloc(npix * 4) || ng.alloc(npix * 4) || bm.alloc(npix) || hh.alloc(cells * n * 8) || zz.alloc((size_t)n * 4) || xx.alloc((size_t)n * 48) ||
        oo.alloc(npix * n))
        return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(nd.p, sen->nodef_dep, npix * 4, hipMemcpyHostToDevice));
    { std::vector<uint8_t> g8(npix); make_gray_u8(sen->nodef_gray, (int)npix, g8.data()); TG_HIP(hipMemcpy(ng.p, g8.data(), npix, hipMemcpyHostToDevice)); }
    TG_HIP(hipMemcpy(bm.p, sen->border_mask, npix, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(hh.p, heights, cells * n * 8, hipMemcpyHostToDevice)); TG_HIP(hipMemcpy(zz.p, zoff, (size_t)n * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(xx.p, xf, (size_t)n * 48, hipMemcpyHostToDevice));
    TG_HIP(hipMemset(oo.p, 0, npix * n));
    RasterParams P = make_raster_params(W, H, sen->fov_deg, sen->near_plane, sen->far_plane, sen->turn_off_border, sen->nodef_dep);
    Stimulus S{};
    S.kind = 1; S.heights = (const double*)hh.p; S.zoff = (const float*)zz.p; S.rows = rows; S.cols = cols; S.scale = (float)grid_scale;
    S.n_tris = (rows - 1) * (cols - 1) * 2;
    launch_render(P, S, (const float*)xx.p, 0, n, nullptr, (const float*)nd.p, (const uint8_t*)ng.p, (const uint8_t*)bm.p, (uint8_t*)oo.p, nullptr, nullptr, nullptr, nullptr, 0);
    TG_HIP(hipDeviceSynchronize());
    TG_HIP(hipMemcpy(out, oo.p, npix * n, hipMemcpyDeviceToHost));
    return 0;
}

int tg_gen_heightfield(int32_t n, const int64_t* seeds, int32_t rows, int32_t cols, double interp, double range, double* heights, float* zoff) {
    if (!seeds || !heights) return fail(-1, "NULL argument");
    if (int rc = need_device()) return rc;
    const size_t cells = (size_t)rows * cols;
    DevBuf sd, hh, zz;
    if (sd.alloc((size_t)n * 8) || hh.alloc(cells * n * 8) || zz.alloc((size_t)n * 4)) return fail(-2, "hipMalloc failed");
    TG_HIP(hipMemcpy(sd.p, seeds, (size_t)n * 8, hipMemcpyHostToDevice));
    launch_gen_surface(n, nullptr, (const int64_t*)sd.p, rows, cols, interp, range, 1, 0, (double*)hh.p, (float*)zz.p, 0);
    TG_HIP(hipDeviceSynchronize());
    TG_HIP(hipMemcpy(heights, hh.p, cells * n * 8, hipMemcpyDeviceToHost));
    if (zoff) TG_HIP(hipMemcpy(zoff, zz.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
