// tg_api.hip — env-step / reset kernels and the C ABI of libtactile_gym_hip.so (see include/tactile_gym_hip.h).
//
// Per-env state lives in HBM as struct-of-arrays with the env index minor ([field][num_envs], doubles), so a
// wavefront of 64 consecutive envs reads or writes one 512-byte contiguous run per field: the whole dynamic state
// crosses HBM exactly once per env step (in) and once (out); the 24 sim ticks in between run out of registers.
#include "tg_ctx.hpp"

namespace tg {

int report_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }   // the other translation units' way to tg_last_error()
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
        if (cfg.balance_object != TG_BALANCE_POLE && cfg.balance_object != TG_BALANCE_BALL_ON_PLATE && cfg.balance_object != TG_BALANCE_SPINNING_PLATE)
            return fail(-1, "object_balance: unknown balance_object");
        c.spin.n_dish = 0; c.spin.n_spool = 0; c.spin.buffer_height = T(0); c.spin.embed0 = T(0);
        if (cfg.balance_object == TG_BALANCE_SPINNING_PLATE) {   // object_balance_env.py:198-239; csrc/tg_spin.hip
            if (!(cfg.spin_dish_mass > 0 && cfg.spin_buffer_height > 0 && cfg.spin_hull_margin >= 0 && cfg.spin_mu >= 0 && cfg.contact_erp > 0))
                return fail(-1, "object_balance spinning_plate: spin_dish_mass, spin_buffer_height, contact_erp must be positive");
            if (cfg.spin_n_dish <= 0 || cfg.spin_n_dish > 1152 || cfg.spin_n_spool <= 0 || cfg.spin_n_spool > 256 || !cfg.spin_dish_hull || !cfg.spin_spool_hull)
                return fail(-1, "object_balance spinning_plate: the hulls are missing or too large (dish <= 1152, spool <= 256 vertices)");
            SpinConst<T>& sp = c.spin;
            sp.mass = (T)cfg.spin_dish_mass;
            sp.com = {(T)cfg.spin_dish_com[0], (T)cfg.spin_dish_com[1], (T)cfg.spin_dish_com[2]};
            sp.inertia = {(T)cfg.spin_dish_inertia[0], (T)cfg.spin_dish_inertia[1], (T)cfg.spin_dish_inertia[2], (T)cfg.spin_dish_inertia[4],
                          (T)cfg.spin_dish_inertia[5], (T)cfg.spin_dish_inertia[8]};
            sp.margin = (T)cfg.spin_hull_margin; sp.breaking = (T)cfg.contact_breaking; sp.erp = (T)cfg.contact_erp; sp.mu = (T)cfg.spin_mu;
            sp.lin_damp = (T)cfg.obj_lin_damp; sp.ang_damp = (T)cfg.obj_ang_damp;
            sp.buffer_height = (T)cfg.spin_buffer_height; sp.embed0 = (T)cfg.embed_dist;
            sp.n_dish = cfg.spin_n_dish; sp.n_spool = cfg.spin_n_spool;
        }
        if (cfg.balance_object == TG_BALANCE_BALL_ON_PLATE) {
            if (!(cfg.ball_radius > 0 && cfg.ball_mass > 0 && cfg.ball_mu >= 0 && cfg.plate_radius > 0 && cfg.contact_erp > 0))
                return fail(-1, "object_balance ball_on_plate: ball_radius, ball_mass, plate_radius, contact_erp must be positive");
            c.ball.radius = (T)cfg.ball_radius; c.ball.mass = (T)cfg.ball_mass;
            c.ball.inertia = (T)(0.4 * cfg.ball_mass * cfg.ball_radius * cfg.ball_radius);
            c.ball.mu = (T)cfg.ball_mu; c.ball.plate_radius = (T)cfg.plate_radius; c.ball.plate_half_len = (T)(0.5 * cfg.obj_base_height);
            c.ball.breaking = (T)cfg.contact_breaking; c.ball.erp = (T)cfg.contact_erp;
            c.ball.lin_damp = (T)cfg.obj_lin_damp; c.ball.ang_damp = (T)cfg.obj_ang_damp;
        }
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
        ps.narrow = cfg.narrowphase;
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
    c.fused_reset = (cfg.auto_reset && (cfg.env_kind == TG_ENV_EDGE_FOLLOW || cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO)) ? 1 : 0;
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

namespace tg {
// Interior-only tactile payload (multi-GPU gather): the border ring of an image is a constant paste of the reference image
// (tactile_sensor.py:291-292), so a rank ships only the K pixels inside the border mask and rank 0 restores the ring.
// rank_of[p] = position of pixel p among the interior pixels (-1 on the ring), idx[k] = pixel of interior position k.
// The payload is made of the image's 4-pixel words that hold at least one interior pixel (Kd words; the few ring pixels they include are the
// sender's constants anyway): both directions are word copies through an index table, four words per lane, one 16-byte store each.
// idx[j] = word of payload position j (padded to a multiple of 4 with repeats of the last), rank_of[q] = payload position of word q (-1:
// a word of ring pixels only).
// env.reset() by the caller: the running return of the envs that start over is dropped (an auto-reset's is cleared by the step itself).
__global__ __launch_bounds__(256) void k_episode_clear(double* __restrict__ ep_return, const uint8_t* __restrict__ mask, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (mask == nullptr || mask[i])) ep_return[i] = 0.0;
}
__global__ __launch_bounds__(256) void k_pack_interior(const uint32_t* __restrict__ obs, const int32_t* __restrict__ idx, int Kd, int HWd, int n_img,
                                                       uint32_t* __restrict__ dst) {
    const int img = blockIdx.y;
    const uint32_t* __restrict__ o = obs + (size_t)img * HWd;
    for (int k = 4 * (blockIdx.x * blockDim.x + threadIdx.x); k < Kd; k += 4 * gridDim.x * blockDim.x) {
        const int4 i = *reinterpret_cast<const int4*>(idx + k);
        *reinterpret_cast<uint4*>(dst + (size_t)img * Kd + k) = make_uint4(o[i.x], o[i.y], o[i.z], o[i.w]);
    }
}
__global__ __launch_bounds__(256) void k_unpack_interior(const uint32_t* __restrict__ src, const int32_t* __restrict__ rank_of, const uint32_t* __restrict__ gray,
                                                         int Kd, int HWd, int n_img, uint32_t* __restrict__ dst) {
    const int img = blockIdx.y;
    const uint32_t* __restrict__ s = src + (size_t)img * Kd;
    for (int q = 4 * (blockIdx.x * blockDim.x + threadIdx.x); q < HWd; q += 4 * gridDim.x * blockDim.x) {
        const int4 r = *reinterpret_cast<const int4*>(rank_of + q);
        const uint4 g = *reinterpret_cast<const uint4*>(gray + q);
        *reinterpret_cast<uint4*>(dst + (size_t)img * HWd + q) =
            make_uint4(r.x < 0 ? g.x : s[r.x], r.y < 0 ? g.y : s[r.y], r.z < 0 ? g.z : s[r.z], r.w < 0 ? g.w : s[r.w]);
    }
}

}  // namespace tg


namespace tg {

// first wavefront start -> last wavefront end over the slots written since the last reduce, added to acc; the slots are cleared for the next scope
__global__ __launch_bounds__(1024) void k_kt_reduce(unsigned long long* __restrict__ slots, size_t n_slots, unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long lo[1024], hi[1024];
    unsigned long long a = ~0ull, b = 0ull;
    for (size_t i = threadIdx.x; i < n_slots; i += 1024) {
        const unsigned long long s0 = slots[2 * i], s1 = slots[2 * i + 1];
        if (s0 != ~0ull || s1 != 0ull) { slots[2 * i] = ~0ull; slots[2 * i + 1] = 0ull; }
        if (s0 != ~0ull) a = s0 < a ? s0 : a;
        if (s1 != 0ull) b = s1 > b ? s1 : b;
    }
    lo[threadIdx.x] = a; hi[threadIdx.x] = b;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            lo[threadIdx.x] = lo[threadIdx.x + k] < lo[threadIdx.x] ? lo[threadIdx.x + k] : lo[threadIdx.x];
            hi[threadIdx.x] = hi[threadIdx.x + k] > hi[threadIdx.x] ? hi[threadIdx.x + k] : hi[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && hi[0] > lo[0] && lo[0] != ~0ull) { acc[0] += hi[0] - lo[0]; acc[1] += 1ull; }
}

// A timed scope of profiling mode (tg_profile_enable): a HIP event pair around the launches, and - for the kernels that carry a KtScope - their
// span by their own clock.
struct Timer {
    tg_ctx* c; int which; hipEvent_t a = nullptr, b = nullptr;
    Timer(tg_ctx* ctx, int w) : c(ctx), which(w) {
        if ((c->profile || c->profile_clock) && c->d_kt && which < 5) { c->st.kt = c->d_kt; c->rp.kt = c->d_kt; }
        if (c->profile) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, c->stream); }
    }
    ~Timer() {
        if (c->profile) { (void)hipEventRecord(b, c->stream); c->events.push_back({a, b, which}); }
        if ((c->profile || c->profile_clock) && c->d_kt && which < 5) {
            hipLaunchKernelGGL(k_kt_reduce, dim3(1), dim3(1024), 0, c->stream, c->d_kt, c->kt_slots, c->d_kt_acc + 2 * which);
            c->st.kt = nullptr; c->rp.kt = nullptr;
        }
    }
};


// reset_phase >= 0: the finished envs' auto-reset (k_reset's body, phase `reset_phase`) runs inside the step's launch (k_step<T, TOPO, true>;
// the UR5 only - on the MG400 a step is 0.7 ms of full ticks and the launch it would save is noise); returns whether it did
template <typename T, int TOPO> static bool launch_step_t(tg_ctx* c, const float* d_actions, int reset_phase = -1) {
    const int n = c->cfg.num_envs;
    if (c->cfg.control_mode == TG_CONTROL_TCP_POSITION) {
        hipLaunchKernelGGL((k_step_pos<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
        return false;
    }
    if constexpr (TOPO == 0 && std::is_same<T, double>::value) {   // (f64 physics: the configuration every BASELINE config runs)
        if (reset_phase >= 0) {
            hipLaunchKernelGGL((k_step<T, TOPO, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                               (const EnvConst<T>*)c->d_const, c->st, d_actions, reset_phase, c->bank_mode != 0 ? (const BankDev*)c->d_bank : (const BankDev*)nullptr);
            return true;
        }
    }
    hipLaunchKernelGGL((k_step<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, d_actions, -1, (const BankDev*)nullptr);
    return false;
}
template <typename T, int TOPO> static void launch_reset_t(tg_ctx* c, const uint8_t* d_mask, int phase, bool bank = false) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_reset<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, d_mask, phase, bank ? (const BankDev*)c->d_bank : (const BankDev*)nullptr);
    if (c->st.kt) c->st.kt += 2 * (size_t)((n + 63) / 64);   // profiling: a second k_reset of the same scope (surface_follow's phase 2) stamps its own slots
}
template <typename T, int TOPO> static void launch_bank_refill_t(tg_ctx* c, int phase) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_bank_refill<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->bank_stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, c->bk, c->aux, phase);
}

template <typename T> static void launch_step_body_t(tg_ctx* c, const float* d_actions) {
    const int n = c->cfg.num_envs;
    if (c->cfg.balance_object == TG_BALANCE_BALL_ON_PLATE) {
        if (c->cfg.control_mode == TG_CONTROL_TCP_POSITION)
            hipLaunchKernelGGL((k_step_body<T, 0, true, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                               (const EnvConst<T>*)c->d_const, c->st, d_actions);
        else
            hipLaunchKernelGGL((k_step_body<T, 0, false, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                               (const EnvConst<T>*)c->d_const, c->st, d_actions);
        return;
    }
    if (c->cfg.control_mode == TG_CONTROL_TCP_POSITION)
        hipLaunchKernelGGL((k_step_body<T, 0, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
    else
        hipLaunchKernelGGL((k_step_body<T, 0, false>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_actions);
}
template <typename T> static void launch_reset_body_t(tg_ctx* c, const uint8_t* d_mask) {
    const int n = c->cfg.num_envs;
    if (c->cfg.balance_object == TG_BALANCE_SPINNING_PLATE) {   // (the template is always there in this mode: tg_create)
        if (c->tmpl_ready)
            hipLaunchKernelGGL((k_reset_body<T, 0, false, true, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                               (const EnvConst<T>*)c->d_const, c->st, d_mask);
        else
            hipLaunchKernelGGL((k_reset_body<T, 0, false, false, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                               (const EnvConst<T>*)c->d_const, c->st, d_mask);
        return;
    }
    if (c->tmpl_ready) {   // a reset that covered env 0 has been enqueued before this one: the template is there when this launch runs
        if (c->cfg.balance_object == TG_BALANCE_BALL_ON_PLATE)
            hipLaunchKernelGGL((k_reset_body<T, 0, true, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                               (const EnvConst<T>*)c->d_const, c->st, d_mask);
        else
            hipLaunchKernelGGL((k_reset_body<T, 0, false, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                               (const EnvConst<T>*)c->d_const, c->st, d_mask);
        return;
    }
    if (c->cfg.balance_object == TG_BALANCE_BALL_ON_PLATE) {
        hipLaunchKernelGGL((k_reset_body<T, 0, true>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                           (const EnvConst<T>*)c->d_const, c->st, d_mask);
        return;
    }
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

static void render(tg_ctx* c, const uint8_t* d_mask, bool save_prev) {
    Timer t(c, d_mask ? 3 : 1);
    launch_render(raster_params(c), c->stim, c->st.stim_xform, 1, c->cfg.num_envs, d_mask, c->d_nodef_dep, c->d_nodef_gray,
                  c->d_border, obs_buf(c), save_prev ? c->d_term : nullptr, nullptr, nullptr, nullptr, c->stream);
}
// fused auto-reset: every env's observation from stim_xform; for the envs flagged in `done` also the terminal observation from term_xform
static void render_fused(tg_ctx* c) {
    Timer t(c, 1);
    launch_render(raster_params(c), c->stim, c->st.stim_xform, 1, c->cfg.num_envs, nullptr, c->d_nodef_dep, c->d_nodef_gray,
                  c->d_border, obs_buf(c), nullptr, c->st.term_xform, c->st.done, c->d_term, c->stream);
}



// env.reset() for the masked envs: task randomisation, (surface generation), robot reset.
// tg_sample_actions: element i of draw `counter`: 24 random bits of splitmix64 over (seed, counter, i) -> lo + (hi - lo) u, u in [0, 1)
__global__ void k_sample_actions(int total, uint64_t seed, uint64_t counter, float lo, float hi, float* __restrict__ out, unsigned long long* tl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    TG_TL(tl, 0);
    if (i >= total) return;
    const uint64_t z = mix64(mix64(seed + kGolden * (counter + 1)) + kGolden * (uint64_t)(i + 1));
    const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
    out[i] = lo + (hi - lo) * u;
}

// tg_step_random's sampler: the draw counter and the seed live in device memory (a captured graph has no per-launch arguments); every thread
// reads the counter, the workgroup that finishes last moves it on - nobody can still be reading it then.  Draw k is tg_sample_actions' draw k.
__global__ void k_sample_actions_ctr(int total, unsigned long long* __restrict__ ctr, float lo, float hi, float* __restrict__ out, unsigned long long* tl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    TG_TL(tl, 0);
    const uint64_t counter = ctr[0] + 1, seed = ctr[1];
    if (i < total) {
        const uint64_t z = mix64(mix64(seed + kGolden * (counter + 1)) + kGolden * (uint64_t)(i + 1));
        const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
        out[i] = lo + (hi - lo) * u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ctr + 2, 1ull) == (unsigned long long)gridDim.x - 1) { ctr[2] = 0ull; ctr[0] = counter; }
    }
}

template <typename T, int TOPO> static void launch_oracle_obs_t(tg_ctx* c, int dim, float* dst) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_oracle_obs<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, dim, dst);
}
static int oracle_dim(const tg_ctx* c);
static void oracle_draw(tg_ctx* c, float* dst) {
    const int d = oracle_dim(c);
#define CALL(T, TOPO) launch_oracle_obs_t<T, TOPO>(c, d, dst)
    TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
}
static int oracle_dim(const tg_ctx* c) {
    switch (c->cfg.env_kind) {
        case TG_ENV_EDGE_FOLLOW: return 10;
        case TG_ENV_SURFACE_FOLLOW_AUTO: return 20;
        case TG_ENV_OBJECT_BALANCE: return 26;
        case TG_ENV_OBJECT_PUSH: return 30;
        default: return 34;
    }
}
template <typename T, int TOPO> static void launch_scene_xf_t(tg_ctx* c, const uint8_t* d_mask) {
    const int n = c->cfg.num_envs;
    hipLaunchKernelGGL((k_scene_xf<T, TOPO>), dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevRobot<T>*)c->d_robot,
                       (const EnvConst<T>*)c->d_const, c->st, c->scene_view, d_mask, c->d_scene_xf, c->d_scene_spheres, c->scene.n_spheres);
}
// get_visual_obs for the whole batch (or the masked envs; save_prev keeps their previous image as the terminal observation)
static void scene_draw(tg_ctx* c, const uint8_t* d_mask, bool save_prev) {
    Timer t(c, 4);
#define CALL(T, TOPO) launch_scene_xf_t<T, TOPO>(c, d_mask)
    TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    launch_scene(c->scene, c->d_scene_xf, c->cfg.num_envs, d_mask, c->d_vis, save_prev ? c->d_vis_term : nullptr, c->stream);
}

// Which mapping steps the contact envs (tg_config.contact_mapping).  One wavefront per env takes the whole register file of its SIMD (one
// wavefront per SIMD, 1024 per chip), so its rate is flat in the batch size - object_push: 0.52 M env-steps/s from 1024 envs up - while one
// lane per env scales with the batch until the chip is full: measured 1024 envs 2.0 ms (wave) against 7.4 ms (lane) per step, 2048 envs
// 3.9 against 7.5, 4096 envs 7.8 against 7.7, 8192 envs 15.5 against 7.9 (1.04 M env-steps/s).
static bool use_contact_wave(const tg_ctx* c) {
    if (c->cfg.env_kind != TG_ENV_OBJECT_PUSH && c->cfg.env_kind != TG_ENV_OBJECT_ROLL && c->cfg.env_kind != TG_ENV_OBJECT_BALANCE) return false;
    if (c->cfg.physics_dtype != TG_PHYSICS_F64) return false;
    if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE && c->cfg.balance_object == TG_BALANCE_BALL_ON_PLATE) return false;   // lane mapping only (tg_create refuses WAVE)
    if (c->cfg.contact_mapping == TG_CONTACT_MAP_LANE) return false;
    if (c->cfg.contact_mapping == TG_CONTACT_MAP_WAVE || c->cfg.narrowphase != TG_NARROW_CLOSED_FORM) return true;
    return c->cfg.num_envs < 4096;
}

// Contact-free arm tasks on the wave mapping (k_step_arm_wave: every tick a full tick on the env's own wavefront).  Measured on an MI355X at
// 1024 envs it does NOT beat the lane mapping: UR5 literal solver k_step 0.886 ms against 0.512 ms, MG400 (surface_follow-v2) 1.05 ms against
// ~0.95 ms - a wave64 instruction costs its 4 cycles whether 6 or 64 lanes do useful work, so the 150 sweeps (~190 issue cycles each on the
// wave mapping, 144 for 64 envs on the lane mapping) and the dynamics are a wash between 1024 half-empty wavefronts and 16 full ones
// (DESIGN.md 4.1g).  Kept for TG_CONTACT_MAP_WAVE only; AUTO stays on the lane mapping.
static bool use_arm_wave(const tg_ctx* c) {
    if (c->cfg.env_kind != TG_ENV_EDGE_FOLLOW && c->cfg.env_kind != TG_ENV_SURFACE_FOLLOW_AUTO) return false;
    if (c->cfg.physics_dtype != TG_PHYSICS_F64 || c->cfg.control_mode != TG_CONTROL_TCP_VELOCITY) return false;
    return c->cfg.contact_mapping == TG_CONTACT_MAP_WAVE;
}

// The env step as ONE launch (tg_fused.hip: k_step_render, the wavefront that steps an env draws it): edge_follow with the lane-mapped k_step
// and the block raster.  OPT-IN (tg_config.fused_step = TG_FUSED_ON / TG_FUSED_STEP=1): measured on an MI355X it is SLOWER than the three
// launches it replaces - 56.2 against 43.3 us per step at 1024 envs, 46.9 against 36.8 at 256, 108 against 75 at 4096
// (profiles/r5_exp_fused_step.txt; DESIGN.md 4.1k has the why: the step code needs the SIMD's whole register file, so the draw runs on one
// wavefront per SIMD without the occupancy the four-wavefront raster hides its latencies with, and four such wavefronts per CU slow each
// other down by 1.7x).  Byte-identical images / rewards / dones (tests/test_gpu_fused_step.py).
bool use_fused_step(const tg_ctx* c) {
    if (c->fused_pref <= 0) return false;   // TG_FUSED_AUTO: off (see above)
    if (c->cfg.env_kind != TG_ENV_EDGE_FOLLOW || c->cfg.physics_dtype != TG_PHYSICS_F64 || c->cfg.control_mode != TG_CONTROL_TCP_VELOCITY) return false;
    if (use_arm_wave(c) || c->scene_every_step || c->oracle_every_step) return false;   // (the scene / oracle draws sit between the step and the reset)
    if (c->stim.kind != 0 || c->stim.n_tris > 32 || c->stim.fills_view || c->rp.blockmax == nullptr || c->rp.tmpl == nullptr) return false;
    if (c->rp.W % 128 != 0 || c->rp.H % 128 != 0) return false;
    static const bool blocks_off = getenv("TG_NO_BLOCK_RASTER") != nullptr;
    if (blocks_off) return false;
    return true;
}

static int surf_gen_mode(const tg_ctx* c) {
    return c->cfg.noise_mode == TG_SNOISE_NONE ? TG_SURF_FLAT : c->cfg.noise_mode == TG_SNOISE_RANDOM ? TG_SURF_RANDOM
           : c->cfg.noise_mode == TG_SNOISE_VERTICAL_SIMPLEX ? TG_SURF_SIMPLEX_1D_VERT
           : (c->cfg.movement_mode == TG_SMOVE_YZ || c->cfg.movement_mode == TG_SMOVE_YZRX) ? TG_SURF_SIMPLEX_1D : TG_SURF_SIMPLEX_2D;
}

// bank: the auto-reset of tg_step with the reset bank on (k_reset mode 1): finished envs take their precomputed state, the rest (aux.late)
// are reset by the launches that follow
static void reset_sequence(tg_ctx* c, const uint8_t* d_mask, bool bank = false, bool phase1_done = false /* surface_follow: k_step<.., true> has run phase 1 */) {
    Timer t(c, 2);
    if (bank && c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        if (!phase1_done) {
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 1, true)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        }
        launch_gen_surface(c->cfg.num_envs, c->aux.late, c->st.noise_seed, c->cfg.surf_rows, c->cfg.surf_cols, c->cfg.surf_interp,
                           c->cfg.surf_height_range, c->cfg.surf_center_z, surf_gen_mode(c), c->st.heights, c->st.surf_zoff, c->stream,
                           c->aux.swapped, c->st.hsel);
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, c->aux.late, 2, true)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        return;
    }
    if (bank && c->cfg.env_kind == TG_ENV_EDGE_FOLLOW) {
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 0, true)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        return;
    }
    if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE) {
        if (c->cfg.physics_dtype == TG_PHYSICS_F64) launch_reset_body_t<double>(c, d_mask);
        else launch_reset_body_t<float>(c, d_mask);
    } else if (c->cfg.env_kind == TG_ENV_OBJECT_ROLL) {
        if (!(use_contact_wave(c) && launch_reset_contact_wave(c->cfg.env_kind, c->cfg.physics_dtype, c->robot.topology, c->cfg.cone_friction, c->cfg.num_envs,
                                                               c->cfg.n_tip_verts, c->stream, c->d_robot, c->d_const, c->st, d_mask) == 0)) {
#define CALL(T, TOPO) launch_reset_roll_t<T, TOPO>(c, d_mask)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        }
    } else if (c->cfg.env_kind == TG_ENV_OBJECT_PUSH) {
        if (!(use_contact_wave(c) && launch_reset_contact_wave(c->cfg.env_kind, c->cfg.physics_dtype, c->robot.topology, c->cfg.cone_friction, c->cfg.num_envs,
                                                               c->cfg.n_tip_verts, c->stream, c->d_robot, c->d_const, c->st, d_mask, c->cfg.narrowphase) == 0)) {
#define CALL(T, TOPO) launch_reset_push_t<T, TOPO>(c, d_mask)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        }
        if (c->cfg.traj_type == TG_TRAJ_SIMPLEX)
            launch_gen_traj(c->cfg.num_envs, d_mask, c->st.noise_seed, c->cfg.traj_n_points, c->cfg.traj_spacing, c->cfg.traj_max_perturb,
                            c->cfg.traj_init_offset, c->cfg.reset_goal_id, c->st.traj, c->st.feature, c->stream);
    } else if (c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
        if (!phase1_done) {
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 1)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        }
        launch_gen_surface(c->cfg.num_envs, d_mask, c->st.noise_seed, c->cfg.surf_rows, c->cfg.surf_cols, c->cfg.surf_interp,
                           c->cfg.surf_height_range, c->cfg.surf_center_z, surf_gen_mode(c), c->st.heights, c->st.surf_zoff, c->stream, nullptr, c->st.hsel);
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 2)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    } else {
#define CALL(T, TOPO) launch_reset_t<T, TOPO>(c, d_mask, 0)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    }
}

// The refill of the reset bank: outside the step graph, on the bank's low-priority stream, paced by a marker on the step stream (so the host
// running ahead of the device cannot spend all its refills before the episodes they are for have ended).  It never blocks the stream the
// steps run on; the only data-path ordering between the two streams is the tag / RNG acquire-release pair (tg_kernels.hpp).
static void bank_refill(tg_ctx* c) {
    if (c->bank_mode == 0) return;
    if ((c->bank_steps++ % c->bank_every) != 0 && c->bank_mode != 2) return;
    if (c->bank_mode == 2) {
        (void)hipEventRecord(c->ev_bank, c->stream);
        (void)hipStreamWaitEvent(c->bank_stream, c->ev_bank, 0);
    } else {
        // Pacing (round 5): no cross-stream wait.  Every visit leaves a marker on the step stream, waits - on the host - for the marker of
        // kBankLag visits ago, and launches the refill at once: it finds the device between 8 kBankLag and 8 (kBankLag + 1) steps behind the
        // host, i.e. the refills stay spread over the rollout however fast the host enqueues, and the host is never further ahead than that.
        // (Until round 5 the refill waited for its marker on the bank stream.  A host that runs ahead - any rollout that does not read a result
        // every step - then keeps a blocked barrier packet in a second hardware queue all the time, and every dispatch of the step stream was
        // ~1 us slower for it: 45.9 instead of 42.1 us per step on the headline, whatever `bank_every` and whatever the queue's priority.
        // TG_BANK_PACE=0 is that scheme.)
        static const bool by_barrier = getenv("TG_BANK_PACE") != nullptr && atoi(getenv("TG_BANK_PACE")) == 0;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(c->stream, &cap);
        if (by_barrier || cap != hipStreamCaptureStatusNone) {
            (void)hipEventRecord(c->ev_bank, c->stream);
            (void)hipStreamWaitEvent(c->bank_stream, c->ev_bank, 0);
        } else {
            // (the ring's events exist since tg_create.  If another thread of the process is capturing in global mode - torch.cuda.graph's default -
            //  an event record / synchronise from here is illegal: the call fails with hipErrorStreamCapture*; the error is cleared and this visit
            //  falls back to the cross-stream wait, which is legal under any capture mode - ADVICE r5)
            const unsigned long long k = c->bank_visits++;
            hipEvent_t ev = c->ev_bank_ring[k % tg_ctx::kBankRing];
            bool paced = ev != nullptr && hipEventRecord(ev, c->stream) == hipSuccess;
            if (paced && k >= tg_ctx::kBankLag) {
                hipEvent_t old = c->ev_bank_ring[(k - tg_ctx::kBankLag) % tg_ctx::kBankRing];
                paced = old != nullptr && hipEventSynchronize(old) == hipSuccess;
            }
            if (!paced) {
                (void)hipGetLastError();
                if (hipEventRecord(c->ev_bank, c->stream) != hipSuccess || hipStreamWaitEvent(c->bank_stream, c->ev_bank, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    return;                     // no ordering to be had this visit: no refill (finished envs reset on the spot, same results)
                }
            }
        }
    }
    if (c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
#define CALL(T, TOPO) launch_bank_refill_t<T, TOPO>(c, 1)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        launch_gen_surface(c->cfg.num_envs, c->aux.need, c->bk.noise_seed, c->cfg.surf_rows, c->cfg.surf_cols, c->cfg.surf_interp,
                           c->cfg.surf_height_range, c->cfg.surf_center_z, surf_gen_mode(c), c->bk.heights, c->bk.surf_zoff, c->bank_stream, nullptr, c->bk.hsel);
#define CALL(T, TOPO) launch_bank_refill_t<T, TOPO>(c, 2)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    } else {
#define CALL(T, TOPO) launch_bank_refill_t<T, TOPO>(c, 0)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    }
    if (c->bank_mode == 2) {   // tests: every finished env finds its entry ready (the step stream waits for the refill)
        (void)hipEventRecord(c->ev_bank_done, c->bank_stream);
        (void)hipStreamWaitEvent(c->stream, c->ev_bank_done, 0);
    }
}

}  // namespace tg

using namespace tg;

extern "C" {

const char* tg_last_error(void) { return g_err.c_str(); }
int tg_abi_version(void) { return TG_ABI_VERSION; }

static int create_impl(const tg_config* cfg, const tg_robot* robot, const tg_sensor* sensor, const tg_mesh* stim, tg_ctx** out);
int tg_create(const tg_config* cfg, const tg_robot* robot, const tg_sensor* sensor, const tg_mesh* stim, tg_ctx** out) {
    if (!out) return fail(-1, "tg_create: NULL argument");
    *out = nullptr;
    const int rc = create_impl(cfg, robot, sensor, stim, out);
    if (rc != 0 && *out) {               // a failure half way: release the context and every device allocation made so far
        const std::string keep = g_err;
        tg_destroy(*out);
        *out = nullptr;
        g_err = keep;
    }
    return rc;
}
static int create_impl(const tg_config* cfg, const tg_robot* robot, const tg_sensor* sensor, const tg_mesh* stim, tg_ctx** out) {
    if (!cfg || !robot || !sensor || !out) return fail(-1, "tg_create: NULL argument");
    if (cfg->abi_version != TG_ABI_VERSION) return fail(-1, "tg_create: ABI version mismatch");
    if (cfg->env_kind != TG_ENV_EDGE_FOLLOW && cfg->env_kind != TG_ENV_SURFACE_FOLLOW_AUTO && cfg->env_kind != TG_ENV_OBJECT_BALANCE &&
        cfg->env_kind != TG_ENV_OBJECT_PUSH && cfg->env_kind != TG_ENV_OBJECT_ROLL)
        return fail(-1, "tg_create: unknown env_kind");
    if (cfg->env_kind != TG_ENV_SURFACE_FOLLOW_AUTO && !stim) return fail(-1, "tg_create: this env needs a stimulus mesh");
    if (cfg->env_kind == TG_ENV_OBJECT_BALANCE && robot->topology != 0) return fail(-1, "tg_create: object_balance is built for the UR5 chain");
    if (cfg->num_envs <= 0) return fail(-1, "tg_create: num_envs must be positive");
    if (cfg->num_envs > 65535) return fail(-1, "tg_create: at most 65535 envs per context (the render launch carries the env index in grid.y); shard over contexts / GPUs");
    if (int rc = check_robot(robot)) return rc;
    const int H = sensor->image_h, W = sensor->image_w;
    if (!((H % 128 == 0 && W % 128 == 0) || (H == 64 && W == 64))) return fail(-1, "tg_create: image size must be 64x64 or a multiple of 128");
    if (!sensor->nodef_dep || !sensor->nodef_gray || !sensor->border_mask) return fail(-1, "tg_create: sensor reference images missing");
    if (cfg->physics_dtype != TG_PHYSICS_F64 && cfg->physics_dtype != TG_PHYSICS_F32) return fail(-1, "tg_create: bad physics_dtype");
    if (cfg->narrowphase < 0 || cfg->narrowphase > TG_NARROW_GJK_SINGLE) return fail(-1, "tg_create: bad narrowphase");
    if (cfg->env_kind == TG_ENV_OBJECT_BALANCE && cfg->balance_object == TG_BALANCE_SPINNING_PLATE) {
        if (cfg->physics_dtype != TG_PHYSICS_F64) return fail(-1, "tg_create: object_balance spinning_plate is built for f64 physics");
        if (!cfg->cone_friction) return fail(-1, "tg_create: object_balance spinning_plate is built with cone friction (enableConeFriction = 1)");
        if (cfg->solver_residual_threshold > 0.0) return fail(-1, "tg_create: object_balance spinning_plate does not run in threshold mode (its reset takes the arm's template)");
    }
    if (cfg->env_kind == TG_ENV_OBJECT_BALANCE && cfg->balance_object == TG_BALANCE_BALL_ON_PLATE) {
        if (cfg->contact_mapping == TG_CONTACT_MAP_WAVE) return fail(-1, "tg_create: object_balance ball_on_plate runs on the lane mapping (contact_mapping auto or lane)");
        if (!cfg->cone_friction) return fail(-1, "tg_create: object_balance ball_on_plate is built with cone friction (enableConeFriction = 1)");
    }
    if (cfg->narrowphase != TG_NARROW_CLOSED_FORM) {
        if (cfg->env_kind != TG_ENV_OBJECT_PUSH) return fail(-1, "tg_create: narrowphase GJK / EPA is built for object_push (the tip core - cube pair)");
        if (cfg->physics_dtype != TG_PHYSICS_F64 || !cfg->cone_friction || cfg->contact_mapping == TG_CONTACT_MAP_LANE)
            return fail(-1, "tg_create: narrowphase GJK / EPA runs on the wave mapping (f64, cone friction, contact_mapping auto or wave)");
        if (cfg->n_tip_verts > 1152) return fail(-1, "tg_create: narrowphase GJK / EPA holds at most 1152 hull vertices");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(-3, "tg_create: no HIP device visible — the tactile-env step has no CPU fallback");
    TG_HIP(hipSetDevice(cfg->device));
    tg_ctx* c = new tg_ctx();
    *out = c;                            // owned by tg_create's guard from here on
    c->cfg = *cfg; c->robot = *robot; c->H = H; c->W = W;
    TG_HIP(hipStreamCreate(&c->own_stream));
    TG_HIP(hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking));   // non-blocking: a capture must not drag the legacy default stream (torch's) into its rules
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
        dr.res_thr = cfg->solver_residual_threshold > 0.0 ? cfg->solver_residual_threshold : 0.0;
        if (int rc = build_env_const(*cfg, *sensor, *robot, ec)) { delete c; return rc; }
        c->act_dim = ec.act_dim;
        TG_HIP(hipMalloc(&c->d_robot, sizeof dr)); TG_HIP(hipMemcpy(c->d_robot, &dr, sizeof dr, hipMemcpyHostToDevice));
        TG_HIP(hipMalloc(&c->d_const, sizeof ec)); TG_HIP(hipMemcpy(c->d_const, &ec, sizeof ec, hipMemcpyHostToDevice));
    } else {
        DevRobot<float> dr; EnvConst<float> ec;
        build_dev_robot(*robot, dr);
        dr.res_thr = cfg->solver_residual_threshold > 0.0 ? (float)cfg->solver_residual_threshold : 0.0f;
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
    TG_HIP(hipMalloc(&s.sweeps, n * 4)); TG_HIP(hipMemset(s.sweeps, 0, n * 4));
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
    c->cfg_turn_off_border = sensor->turn_off_border != 0;
    {   // what a tile is compared with (tg_pack_tiles): t_s_camera's output for an untouched sensor (tactile_sensor.py:261-294)
        std::vector<uint8_t> g8(npix), tmpl(npix, 0);
        make_gray_u8(sensor->nodef_gray, (int)npix, g8.data());
        if (!c->cfg_turn_off_border)
            for (size_t p = 0; p < npix; ++p) tmpl[p] = sensor->border_mask[p] == 1 ? g8[p] : 0;
        TG_HIP(hipMalloc(&c->d_tile_tmpl, npix)); TG_HIP(hipMemcpy(c->d_tile_tmpl, tmpl.data(), npix, hipMemcpyHostToDevice));
    }
    {   // interior word tables (tg_pack_interior / tg_unpack_interior): the 4-pixel words that hold at least one interior pixel
        const size_t nw = (size_t)npix / 4;
        std::vector<int32_t> idx, rank_of(nw, -1);
        for (size_t q = 0; q < nw; ++q) {
            bool any = false;
            for (int e = 0; e < 4; ++e) any = any || sensor->border_mask[4 * q + e] != 1;
            if (any) { rank_of[q] = (int32_t)idx.size(); idx.push_back((int32_t)q); }
        }
        while (idx.size() % 4 != 0 && !idx.empty()) idx.push_back(idx.back());   // payload rows are padded to a multiple of 4 words
        c->n_interior = (int)idx.size();                                         // in words
        TG_HIP(hipMalloc(&c->d_int_idx, std::max<size_t>(idx.size(), 1) * 4)); TG_HIP(hipMalloc(&c->d_int_rank, (size_t)npix));
        TG_HIP(hipMemcpy(c->d_int_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
        TG_HIP(hipMemcpy(c->d_int_rank, rank_of.data(), (size_t)npix, hipMemcpyHostToDevice));
    }
    if (cfg->env_kind == TG_ENV_OBJECT_PUSH) {
        // the arm's post-reset state + the tip's path (k_reset_contact_wave's reset template); off with reset_bank = TG_BANK_OFF / TG_RESET_BANK=0 and in
        // threshold mode (the truncated solve couples the arm to the object's rows)
        bool tmpl = cfg->reset_bank != TG_BANK_OFF && !(cfg->solver_residual_threshold > 0.0);
        if (const char* e = getenv("TG_RESET_BANK")) tmpl = tmpl && e[0] != '0';
        const size_t tb = (size_t)(2 * TG_MAX_DOF + 4 + 3 * 64) * 8;
        if (tmpl) { TG_HIP(hipMalloc(&s.reset_tmpl, tb)); TG_HIP(hipMemset(s.reset_tmpl, 0, tb)); }
        TG_HIP(hipMalloc(&s.tmpl_stats, 16)); TG_HIP(hipMemset(s.tmpl_stats, 0, 16));
    }
    if (cfg->env_kind == TG_ENV_OBJECT_BALANCE) {
        TG_HIP(hipMalloc(&s.body_pos, 3 * n * 8)); TG_HIP(hipMalloc(&s.body_rot, 9 * n * 8)); TG_HIP(hipMalloc(&s.body_v, 3 * n * 8));
        TG_HIP(hipMalloc(&s.body_w, 3 * n * 8)); TG_HIP(hipMalloc(&s.ext_pos, 3 * n * 8)); TG_HIP(hipMalloc(&s.gravity, n * 8));
        TG_HIP(hipMalloc(&s.ext_pending, n));
        TG_HIP(hipMemset(s.body_v, 0, 3 * n * 8)); TG_HIP(hipMemset(s.body_w, 0, 3 * n * 8)); TG_HIP(hipMemset(s.ext_pos, 0, 3 * n * 8));
        TG_HIP(hipMemset(s.ext_pending, 0, n));
        {   // the arm's post-reset state, computed once (k_reset_body); off with reset_bank = TG_BANK_OFF / TG_RESET_BANK=0
            bool tmpl = cfg->reset_bank != TG_BANK_OFF;
            if (const char* e = getenv("TG_RESET_BANK")) tmpl = e[0] != '0';
            if (cfg->solver_residual_threshold > 0.0) tmpl = false;   // threshold mode: the reset tick's truncated solve sees the fallen object (1e-6 rad): every reset is recomputed
            if (cfg->balance_object == TG_BALANCE_SPINNING_PLATE) tmpl = true;   // (its literal reset moves the arm with the spool only: the template is that state)
            if (tmpl) { TG_HIP(hipMalloc(&s.reset_tmpl, (2 * TG_MAX_DOF + 2) * 8)); TG_HIP(hipMemset(s.reset_tmpl, 0, (2 * TG_MAX_DOF + 2) * 8)); }
        }
        if (cfg->balance_object == TG_BALANCE_SPINNING_PLATE) {
            const size_t hw = (size_t)3 * (cfg->spin_n_dish + cfg->spin_n_spool);
            double* dh = nullptr;
            TG_HIP(hipMalloc(&dh, hw * 8));
            TG_HIP(hipMemcpy(dh, cfg->spin_dish_hull, (size_t)3 * cfg->spin_n_dish * 8, hipMemcpyHostToDevice));
            TG_HIP(hipMemcpy(dh + (size_t)3 * cfg->spin_n_dish, cfg->spin_spool_hull, (size_t)3 * cfg->spin_n_spool * 8, hipMemcpyHostToDevice));
            s.spin_hulls = dh;
            c->cfg.spin_dish_hull = nullptr; c->cfg.spin_spool_hull = nullptr;   // the host pointers are not kept
            TG_HIP(hipMalloc(&s.dish, (size_t)20 * n * 8)); TG_HIP(hipMemset(s.dish, 0, (size_t)20 * n * 8));
            TG_HIP(hipMalloc(&s.mani, (size_t)37 * n * 8)); TG_HIP(hipMemset(s.mani, 0, (size_t)37 * n * 8));
        }
        if (cfg->balance_object == TG_BALANCE_BALL_ON_PLATE) {   // load_ball (:241-260): at workframe + (0, 0, radius), at rest
            TG_HIP(hipMalloc(&s.ball, (size_t)13 * n * 8));
            std::vector<double> bl((size_t)13 * n, 0.0);
            for (int i = 0; i < n; ++i) {
                bl[(size_t)0 * n + i] = cfg->workframe_pos[0]; bl[(size_t)1 * n + i] = cfg->workframe_pos[1];
                bl[(size_t)2 * n + i] = cfg->workframe_pos[2] + cfg->ball_radius;
            }
            TG_HIP(hipMemcpy(s.ball, bl.data(), bl.size() * 8, hipMemcpyHostToDevice));
        }
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
        if (cfg->balance_object == TG_BALANCE_SPINNING_PLATE) {   // the dish where the pole would be, on top of the spool (:215-219); the spool at init_buffer_pos (:228-233)
            std::vector<double> ds((size_t)20 * n, 0.0);
            for (int i = 0; i < n; ++i) {
                for (int a = 0; a < 3; ++a) ds[(size_t)a * n + i] = p0[a] + (a == 2 ? cfg->spin_buffer_height : 0.0);
                for (int a = 0; a < 9; ++a) ds[(size_t)(3 + a) * n + i] = oR[a];
                bp[(size_t)0 * n + i] = cfg->workframe_pos[0]; bp[(size_t)1 * n + i] = cfg->workframe_pos[1];
                bp[(size_t)2 * n + i] = cfg->workframe_pos[2] + cfg->spin_buffer_height / 2;
                for (int a = 0; a < 9; ++a) br[(size_t)a * n + i] = (a % 4 == 0) ? 1.0 : 0.0;
            }
            TG_HIP(hipMemcpy(s.dish, ds.data(), ds.size() * 8, hipMemcpyHostToDevice));
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
        if (cfg->narrowphase != TG_NARROW_CLOSED_FORM) { TG_HIP(hipMalloc(&s.mani, (size_t)37 * n * 8)); TG_HIP(hipMemset(s.mani, 0, (size_t)37 * n * 8)); }
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
        // (three surfaces per env - State::hsel: live 0, last 2, spare 1 to begin with)
        TG_HIP(hipMalloc(&s.dir, 2 * n * 8)); TG_HIP(hipMalloc(&s.goal, 3 * n * 8)); TG_HIP(hipMalloc(&s.heights, 3 * cells * n * 8));
        TG_HIP(hipMalloc(&s.hsel, n)); TG_HIP(hipMemset(s.hsel, 2 << 2, n));
        TG_HIP(hipMalloc(&s.surf_zoff, 3 * n * 4)); TG_HIP(hipMalloc(&s.noise_seed, n * 8)); TG_HIP(hipMalloc(&s.accum, n * 8));
        TG_HIP(hipMemset(s.dir, 0, 2 * n * 8)); TG_HIP(hipMemset(s.goal, 0, 3 * n * 8)); TG_HIP(hipMemset(s.heights, 0, 3 * cells * n * 8));
        TG_HIP(hipMemset(s.surf_zoff, 0, 3 * n * 4)); TG_HIP(hipMemset(s.noise_seed, 0, n * 8)); TG_HIP(hipMemset(s.accum, 0, n * 8));
        TG_HIP(hipMalloc(&s.term_feature, (size_t)12 * n * 4)); TG_HIP(hipMemset(s.term_feature, 0, (size_t)12 * n * 4));
        c->stim.kind = 1; c->stim.heights = s.heights; c->stim.zoff = s.surf_zoff; c->stim.hsel = s.hsel;
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
        // The per-quad reject of k_render_small is off for every shared mesh: with the back faces culled at set-up the records are few, and
        // without it the kernel needs 98 instead of 126 VGPRs - five workgroups per CU instead of four (edge_follow render 34.5 -> 31.7 us,
        // 16 384 envs 0.340 -> 0.289 ms; object_push's cube unchanged; object_balance's plate never used it).
        c->stim.skip_quad_reject = 1;
        c->stim.fills_view = cfg->env_kind == TG_ENV_OBJECT_BALANCE ? 1 : 0;
        c->stim.closed_outward = (mesh_closed_outward(stim) && getenv("TG_NO_BACKFACE_CULL") == nullptr) ? 1 : 0;   // env var: A/B measurements only
        c->stim.kind = 0; c->stim.verts = c->d_verts; c->stim.tris = c->d_tris; c->stim.soup = c->d_soup; c->stim.n_tris = stim->n_tris;
    }
    // one allocation [tactile obs u8 | reward f32 | done u8]: what a rank ships to rank 0 per step is one contiguous byte range
    // (tg_get_packed_outputs).  The obs block is padded to 16 bytes so the reward block stays aligned.
    c->packed_obs_bytes = ((size_t)npix * n + 15) & ~(size_t)15;
    c->packed_bytes = c->packed_obs_bytes + (size_t)n * 4 + (size_t)n;
    const bool has_feature = env_has_feature(cfg->env_kind);
    if (has_feature) {   // extended_feature rides in the same message (config 4's tactile_and_feature observation)
        c->packed_feature_off = (c->packed_bytes + 3) & ~(size_t)3;
        c->packed_bytes = c->packed_feature_off + (size_t)n * 12 * 4;
    }
    TG_HIP(hipMalloc(&c->d_obs, c->packed_bytes)); TG_HIP(hipMemset(c->d_obs, 0, c->packed_bytes));
    s.reward = (float*)(c->d_obs + c->packed_obs_bytes);
    s.done = c->d_obs + c->packed_obs_bytes + (size_t)n * 4;
    if (has_feature) s.feature = (float*)(c->d_obs + c->packed_feature_off);
    TG_HIP(hipMalloc(&c->d_term, npix * n)); TG_HIP(hipMemset(c->d_term, 0, npix * n));
    TG_HIP(hipMalloc(&c->d_episode, (size_t)n * 16)); TG_HIP(hipMemset(c->d_episode, 0, (size_t)n * 16));
    s.ep_return = (double*)c->d_episode;
    s.ep_final_return = (float*)(c->d_episode + (size_t)n * 8);
    s.ep_final_len = (int32_t*)(c->d_episode + (size_t)n * 12);
    TG_HIP(hipMalloc(&c->d_mask, n));
#ifdef TG_TL_STAMPS
    TG_HIP(hipMalloc(&s.tl, (4 * 8192 + 4) * 8)); TG_HIP(hipMemset(s.tl, 0, (4 * 8192 + 4) * 8));
#endif
    TG_HIP(hipMalloc(&c->d_actions, (size_t)n * 6 * sizeof(float)));
    c->rp = make_raster_params(W, H, sensor->fov_deg, sensor->near_plane, sensor->far_plane, sensor->turn_off_border, sensor->nodef_dep);
    if (make_block_tables(c->rp, sensor->nodef_dep, sensor->nodef_gray, sensor->border_mask, n, &c->d_block_tables)) return fail(-2, "hipMalloc failed (raster block tables)");
    c->fused_pref = cfg->fused_step == TG_FUSED_OFF ? -1 : cfg->fused_step == TG_FUSED_ON ? 1 : 0;
    if (const char* e = getenv("TG_FUSED_STEP")) c->fused_pref = e[0] == '0' ? -1 : 1;   // A/B switch (tests, measurements)
#ifdef TG_TL_STAMPS
    c->rp.tl = s.tl;
#endif
    c->bk = s;
    {   // reset bank: edge_follow / surface_follow with auto_reset on the lane mapping (tg_config.reset_bank, TG_RESET_BANK)
        // TG_BANK_AUTO: on (round 5; until then only for the MG400, whose reset is 0.9 ms).  With every env finishing in the same step - a random
        // rollout from a common start - a UR5 reset on the spot costs 0.1 ms once per episode and the bank buys nothing; with episodes that end
        // in different steps - any RL run - some env finishes in nearly every step and the step waits for that reset every time: 6.7 against
        // 18.3 M env-steps/s at 1024 envs (tools/desync_rate.py), while the aligned rollout is unchanged since bank_refill stopped waiting across
        // streams.
        int want = cfg->reset_bank == TG_BANK_OFF ? 0 : cfg->reset_bank == TG_BANK_SYNC ? 2 : 1;
        if (const char* e = getenv("TG_RESET_BANK")) want = (e[0] == '0') ? 0 : (e[0] == 's') ? 2 : 1;
        if (const char* e = getenv("TG_RESET_BANK_EVERY")) { const int k = atoi(e); if (k >= 1) c->bank_every = k; }
        const bool kind_ok = cfg->env_kind == TG_ENV_EDGE_FOLLOW || cfg->env_kind == TG_ENV_SURFACE_FOLLOW_AUTO;
        if (want && kind_ok && cfg->auto_reset && !use_arm_wave(c)) {
            State& b = c->bk;
            auto grab = [&](auto*& ptr, size_t bytes) -> int {
                void* p_ = nullptr;
                if (hipMalloc(&p_, bytes) != hipSuccess) return -1;
                (void)hipMemset(p_, 0, bytes);
                c->bank_allocs.push_back(p_);
                ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(p_);
                return 0;
            };
            int bad = 0;
            bad |= grab(b.q, nd * 8); bad |= grab(b.qd, nd * 8); bad |= grab(b.qd_target, nd * 8);
            bad |= grab(b.tcp_pos, (size_t)3 * n * 8); bad |= grab(b.tcp_rpy, (size_t)3 * n * 8);
            bad |= grab(b.edge_ang, (size_t)n * 8); bad |= grab(b.embed, (size_t)n * 8); bad |= grab(b.edge_sc, (size_t)2 * n * 8);
            bad |= grab(b.stim_xform, (size_t)12 * n * 4);
            bad |= grab(b.step_count, (size_t)n * 4); bad |= grab(b.reset_ticks, (size_t)n * 4); bad |= grab(b.licence, (size_t)n * 4);
            bad |= grab(b.trig_sc, (size_t)16 * n * 8);
            bad |= grab(b.rng, (size_t)n * 8);
            if (cfg->env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) {
                bad |= grab(b.dir, (size_t)2 * n * 8); bad |= grab(b.goal, (size_t)3 * n * 8); bad |= grab(b.hsel, (size_t)n);   // (heights / surf_zoff: the env state's own, the entry's surface goes to the spare third)
                bad |= grab(b.accum, (size_t)n * 8); bad |= grab(b.noise_seed, (size_t)n * 8);
                if (s.feature) bad |= grab(b.feature, (size_t)12 * n * 4);
            }
            bad |= grab(c->aux.tag, (size_t)n * 8); bad |= grab(c->aux.rng_in, (size_t)n * 8);
            bad |= grab(c->aux.stats, 16);
            bad |= grab(c->aux.need, (size_t)n); bad |= grab(c->aux.late, (size_t)n); bad |= grab(c->aux.swapped, (size_t)n);
            if (bad) return fail(-2, "hipMalloc failed (reset bank)");
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = the least priority
            if (getenv("TG_BANK_PRIO0")) lo = 0;
            TG_HIP(hipStreamCreateWithPriority(&c->bank_stream, hipStreamNonBlocking, lo));
            TG_HIP(hipEventCreateWithFlags(&c->ev_bank, hipEventDisableTiming)); TG_HIP(hipEventCreateWithFlags(&c->ev_bank_done, hipEventDisableTiming));
            for (int k = 0; k < tg_ctx::kBankRing; ++k) TG_HIP(hipEventCreateWithFlags(&c->ev_bank_ring[k], hipEventDisableTiming));   // the refill's pacing markers (bank_refill)
            c->aux.enabled = 1;
            c->bank_mode = want;
            BankDev hb{c->bk, c->aux};
            bad |= grab(c->d_bank, sizeof hb);
            if (bad) return fail(-2, "hipMalloc failed (reset bank)");
            TG_HIP(hipMemcpy(c->d_bank, &hb, sizeof hb, hipMemcpyHostToDevice));
        }
    }
    return 0;
}


}  // extern "C"
void drop_step_graphs(tg_ctx* c) {   // every captured step graph of every render target (they are captured again on the next step)
    for (int t = 0; t < 3; ++t) {
        for (int k = 0; k < 2; ++k) if (c->step_graph_t[t][k]) { (void)hipGraphExecDestroy(c->step_graph_t[t][k]); c->step_graph_t[t][k] = nullptr; }
        if (c->random_graph_t[t]) { (void)hipGraphExecDestroy(c->random_graph_t[t]); c->random_graph_t[t] = nullptr; }
    }
}
extern "C" {
int tg_destroy(tg_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->cfg.device);
    (void)hipStreamSynchronize(c->stream);
    drain_events(c);
    if (c->scene_on) scene_debug_stats();
    raster_debug_stats();
#ifdef TG_TL_STAMPS
    if (c->st.tl) {
        std::vector<unsigned long long> h(4 * 8192 + 4);
        if (hipMemcpy(h.data(), c->st.tl, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            const unsigned long long cnt = h[4 * 8192 + 1];   // k_step launches
            if (cnt > 600 && h[4 * 8192 + 0] == cnt) {
                // the last 500 steps: mean interval sampler -> k_step -> k_reset -> render -> next sampler (10 ns ticks)
                double d[4] = {0, 0, 0, 0}; int m = 0;
                const bool has_reset = h[4 * 8192 + 2] >= cnt;
                const unsigned long long roff = h[4 * 8192 + 2] - cnt, qoff = h[4 * 8192 + 3] - cnt;   // resets / renders issued before the first step
                for (unsigned long long s_ = cnt - 501; s_ + 1 < cnt; ++s_) {
                    const unsigned long long a = h[0 * 8192 + (s_ & 8191)], b = h[1 * 8192 + (s_ & 8191)], r = has_reset ? h[2 * 8192 + ((s_ + roff) & 8191)] : 0,
                                             q = h[3 * 8192 + ((s_ + qoff) & 8191)], a2 = h[0 * 8192 + ((s_ + 1) & 8191)];
                    d[0] += (double)(b - a); d[1] += has_reset ? (double)(r - b) : 0.0; d[2] += (double)(q - (has_reset ? r : b)); d[3] += (double)(a2 - q); ++m;
                }
                fprintf(stderr, "TL (%d steps, starts): sampler->k_step %.2f us, k_step->k_reset %.2f us, ->render %.2f us, render->next sampler %.2f us, sum %.2f us\n", m,
                        d[0] / m / 100.0, d[1] / m / 100.0, d[2] / m / 100.0, d[3] / m / 100.0, (d[0] + d[1] + d[2] + d[3]) / m / 100.0);
            }
        }
    }
#endif
#ifdef TG_KSTEP_STAMPS
    { unsigned long long h[16]; if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_kstep_stamps), sizeof h) == hipSuccess) { fprintf(stderr, "k_step stamps (cycles from start, full=%llu):", h[15]); for (int i = 1; i < 13; ++i) fprintf(stderr, " [%d] %lld", i, (long long)(h[i] - h[0])); fprintf(stderr, "\n"); } }
#endif
    // nothing of this context may still be running when its arrays go (the refill stream reads st.rng and writes the bank)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->bank_stream) (void)hipStreamSynchronize(c->bank_stream);
    drop_step_graphs(c);
    for (int k = 0; k < 2; ++k) if (c->drawn_ext[k]) (void)hipFree(c->drawn_ext[k]);
    if (c->d_draw) (void)hipFree(c->d_draw);
    if (c->d_bp) (void)hipFree(c->d_bp);
    if (c->d_bp_hull) (void)hipFree(c->d_bp_hull);
    if (c->d_bp_out) (void)hipFree(c->d_bp_out);
    if (c->d_bp_tot) (void)hipFree(c->d_bp_tot);
    if (c->d_kt) (void)hipFree(c->d_kt);
    if (c->d_kt_acc) (void)hipFree(c->d_kt_acc);
    State& s = c->st;
    void* ptrs[] = {c->d_robot, c->d_const, s.q, s.qd, s.qd_target, s.tcp_pos, s.tcp_rpy, s.edge_ang, s.embed, s.stim_xform, s.term_xform,
                    s.step_count, s.reset_ticks, s.licence, s.sweeps, s.tmpl_stats, s.trig_sc, s.edge_sc, s.rng, s.dir, s.goal, s.heights, s.hsel, s.accum, s.surf_zoff, s.noise_seed, s.body_pos, s.body_rot, s.body_v, s.body_w, s.ext_pos, s.gravity, s.ext_pending, s.ball, s.dish, const_cast<double*>(s.spin_hulls), s.reset_tmpl, s.traj, s.obj_mass, s.goal_id, s.contact_code, s.term_feature, s.mani, const_cast<void*>(s.tip_verts), c->d_nodef_dep, c->d_nodef_gray, c->d_border, c->d_verts, c->d_soup, c->d_tris,
                    c->d_obs, c->d_term, c->d_mask, c->d_actions, c->d_scene_verts, c->d_scene_xf, c->d_scene_spheres, c->d_scene_tris, c->d_scene_attr, c->d_scene_local, c->d_scene_static, c->d_scene_chunks, c->d_vis, c->d_vis_term, c->d_oracle, c->d_oracle_term, c->d_int_idx, c->d_int_rank, c->d_tile_tmpl, c->d_episode, c->d_block_tables};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->bank_stream) { (void)hipStreamSynchronize(c->bank_stream); (void)hipStreamDestroy(c->bank_stream); }
    if (c->ev_bank) (void)hipEventDestroy(c->ev_bank);
    if (c->ev_bank_done) (void)hipEventDestroy(c->ev_bank_done);
    for (hipEvent_t& e : c->ev_bank_ring) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (c->h_rows) { (void)hipHostFree(c->h_rows); c->h_rows = nullptr; }
    for (void* p_ : c->bank_allocs) (void)hipFree(p_);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->capture_stream) (void)hipStreamDestroy(c->capture_stream);
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
    TG_ENTER(c);
    if (n != c->cfg.num_envs) return fail(-1, "tg_seed: need one seed per env");
    std::vector<uint64_t> st(n);
    for (int i = 0; i < n; ++i) st[i] = mix64(seeds[i] + kGolden);
    if (c->bank_stream) TG_HIP(hipStreamSynchronize(c->bank_stream));   // a refill in flight reads the RNG states this call replaces
    TG_HIP(hipMemcpyAsync(c->st.rng, st.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_reset(tg_ctx* c, const uint8_t* host_mask) {
    if (!c) return fail(-1, "NULL ctx");
    TG_ENTER(c);
    const uint8_t* dmask = nullptr;
    if (host_mask) {
        TG_HIP(hipMemcpyAsync(c->d_mask, host_mask, c->cfg.num_envs, hipMemcpyHostToDevice, c->stream));
        dmask = c->d_mask;
    }
    reset_sequence(c, dmask);
    if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE && c->st.reset_tmpl != nullptr && (host_mask == nullptr || host_mask[0] != 0))
        c->tmpl_ready = true;   // env 0 went through the full reset just enqueued: later launches (stream order) take the template-only kernel
    hipLaunchKernelGGL(tg::k_episode_clear, dim3((c->cfg.num_envs + 255) / 256), dim3(256), 0, c->stream, c->st.ep_return, dmask, c->cfg.num_envs);
    render(c, dmask, false);
    if (c->scene_every_step) scene_draw(c, dmask, false);
    if (c->oracle_every_step) oracle_draw(c, c->d_oracle);
    TG_HIP(hipGetLastError());
    return 0;
}

// Whether a step is replayed as a captured graph or enqueued launch by launch.  Until round 6 every step was ONE graph launch; measured on an
// MI355X with ROCm 7.2 (tools/dev/graph_floor.hip: 15-us kernels back to back, the device never idle) a graph launch costs ~6.6 us before its
// first kernel starts and ~1.7 us per further dependent node, launches on the stream ~2.0 us each - and on the host a graph launch is ~9 us
// against ~2.8 us per kernel launch.  The steps here are 2 - 4 launches (the policy draw and the auto-reset live inside k_step), so the stream
// wins up to ~16 launches per step: headline 42.2 -> 37.2 us per step (24.3 -> 27.5 M env-steps/s), episodes out of phase 47.8 -> 42.8 us,
// surface_follow-v0 100.4 -> 95.8 us, object_balance 117 -> 112.5 us (tools/desync_rate.py, same box, alternating).  Default: the stream.
// TG_STEP_GRAPH=1 keeps the graphs (A/B measurements; a host that cannot keep two launches per step ahead of the device).
static bool step_as_graph(const tg_ctx* c) {
    static const int forced = getenv("TG_STEP_GRAPH") != nullptr ? atoi(getenv("TG_STEP_GRAPH")) : -1;
    if (forced >= 0) return forced != 0;
    (void)c;
    return false;
}

static void enqueue_step(tg_ctx* c, const float* d_act) {
    if (c->profile) { Timer t(c, 5); }   // an empty event pair: what every per-kernel figure of this mode carries on top of its kernel
    bool reset_inlined = false;          // object_balance: k_step_body_wave has reset its finished envs itself
    if (use_fused_step(c)) {
        Timer t(c, 1);
        const int rc = launch_step_render(c->robot.topology, c->cfg.num_envs, c->stream, c->d_robot, c->d_const, c->st, d_act, c->cfg.auto_reset,
                                          c->bank_mode != 0 ? c->d_bank : nullptr, raster_params(c), c->stim, c->d_nodef_dep, c->d_nodef_gray, c->d_border, obs_buf(c),
                                          c->d_term);
        if (rc == 0) return;
    }
    {
        Timer t(c, 0);
        if (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE) {
            // the reset of finished envs inside the step's launch: auto-reset with a valid template, the pole, no scene / oracle draw between
            // the step and the reset (they show the pre-reset state)
            const int inline_reset = (c->cfg.auto_reset && c->st.reset_tmpl != nullptr && c->tmpl_ready && c->cfg.balance_object == TG_BALANCE_POLE &&
                                      !c->scene_every_step && !c->oracle_every_step && getenv("TG_NO_INLINE_RESET") == nullptr) ? 1 : 0;
            if (c->cfg.balance_object == TG_BALANCE_SPINNING_PLATE) {
                (void)launch_step_spin(c->cfg.physics_dtype, c->robot.topology, c->cfg.control_mode, c->cfg.num_envs, c->cfg.spin_n_dish, c->stream, c->d_robot,
                                       c->d_const, c->st, d_act);   // one wavefront per env (tg_spin.hip); tg_create has checked the combination
            } else if (use_contact_wave(c) && launch_step_body_wave(c->cfg.physics_dtype, c->robot.topology, c->cfg.control_mode, c->cfg.num_envs, c->stream, c->d_robot,
                                                             c->d_const, c->st, d_act, inline_reset) == 0) {
                reset_inlined = inline_reset != 0;
                // one wavefront per env: the env's own licence, full ticks on the wave mapping (tg_contact_wave.hip)
            } else if (c->cfg.physics_dtype == TG_PHYSICS_F64) launch_step_body_t<double>(c, d_act);
            else launch_step_body_t<float>(c, d_act);
        } else if (use_contact_wave(c) && launch_step_contact_wave(c->cfg.env_kind, c->cfg.physics_dtype, c->robot.topology, c->cfg.control_mode, c->cfg.cone_friction,
                                                                  c->cfg.num_envs, c->cfg.n_tip_verts, c->stream, c->d_robot, c->d_const, c->st, d_act, c->cfg.narrowphase) == 0) {
            // object_push / object_roll with one wavefront per env (tg_contact_wave.hip)
        } else if (c->cfg.env_kind == TG_ENV_OBJECT_ROLL) {
#define CALL(T, TOPO) launch_step_roll_t<T, TOPO>(c, d_act)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        } else if (c->cfg.env_kind == TG_ENV_OBJECT_PUSH) {
#define CALL(T, TOPO) launch_step_push_t<T, TOPO>(c, d_act)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        } else if (use_arm_wave(c) && launch_step_arm_wave(c->cfg.physics_dtype, c->robot.topology, c->cfg.control_mode, c->cfg.num_envs, c->stream, c->d_robot,
                                                           c->d_const, c->st, d_act) == 0) {
            // edge_follow / surface_follow with one wavefront per env (tg_contact_wave.hip: k_step_arm_wave)
        } else {
            // edge_follow / surface_follow: the auto-reset's first launch (edge_follow: its only one) inside the step kernel, unless something
            // is drawn or checked between the step and the reset (scene / oracle observations, the broadphase guard: the pre-reset state)
            static const bool no_inline = getenv("TG_NO_INLINE_RESET") != nullptr;
            const bool fused_render_path = c->cfg.env_kind == TG_ENV_EDGE_FOLLOW || c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO;
            // Only with the reset bank on: the swap-in is a copy, identical wherever it runs; a reset computed on the spot (IK, blocking move) inside
            // k_step<.., true> is another instantiation of reset_env than k_reset's, and the compiler contracts each one's f64 expressions into FMAs
            // on its own - measured: one build in which a bank-off rollout differed from the k_reset launch's in the last bits of q (12 grey levels
            // of one image sum).  With the bank off the reset is 90 us on the spot anyway, and it stays the k_reset launch.  (With the bank on, an env
            // that finds no entry - none in any measured rollout - is reset on the spot in here, as k_step_render does.)
            const int reset_phase = (c->cfg.auto_reset && fused_render_path && c->bank_mode != 0 && !no_inline && !c->scene_every_step && !c->oracle_every_step && !(c->d_bp && c->bp_every_step))
                                        ? (c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO ? 1 : 0) : -1;
#define CALL(T, TOPO) reset_inlined = launch_step_t<T, TOPO>(c, d_act, reset_phase)
            TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
        }
    }
    // broadphase guard: the state the step kernel left, before any reset teleports a finished env (tg_set_broadphase, every_step)
    if (c->d_bp && c->bp_every_step) (void)launch_broadphase(c->cfg.physics_dtype, c->robot.topology, c->cfg.num_envs, c->stream, c->d_robot, c->d_bp, c->st, c->d_bp_out, c->d_bp_tot);
    // visual observation modes: the step's image of every env before any reset touches the state; the envs that finished are redrawn
    // after their reset below (their step image moves to the terminal buffer)
    if (c->scene_every_step) scene_draw(c, nullptr, false);
    if (c->oracle_every_step) oracle_draw(c, c->cfg.auto_reset ? c->d_oracle_term : c->d_oracle);   // the step's own vectors, before any reset
    if (c->cfg.auto_reset && (c->cfg.env_kind == TG_ENV_EDGE_FOLLOW || c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO || (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE && c->st.reset_tmpl != nullptr))) {
        // (object_balance with the reset template: k_reset_body is a few microseconds - teleport, draws, one forward kinematics - so it runs
        //  in line like edge_follow's, without the fork / join of the branch below and without the masked second render: 236 -> 20x us per step)
        if (!reset_inlined) reset_sequence(c, c->st.done, c->bank_mode != 0); // k_reset keeps the terminal camera transform of the envs it resets
        else if (c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) reset_sequence(c, c->st.done, c->bank_mode != 0, true);   // phase 1 ran inside k_step
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
            launch_render(raster_params(c), c->stim, c->st.term_xform, 1, c->cfg.num_envs, nullptr, c->d_nodef_dep, c->d_nodef_gray, c->d_border, obs_buf(c),
                          nullptr, nullptr, nullptr, nullptr, c->stream);
        }
        (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
        render(c, c->st.done, true);
    } else {
        render(c, nullptr, false);
        if (c->cfg.auto_reset) {
            reset_sequence(c, c->st.done, c->bank_mode != 0);
            render(c, c->st.done, true);   // terminal observation is saved, then the post-reset observation is drawn
        }
    }
    if (c->scene_every_step && c->cfg.auto_reset) scene_draw(c, c->st.done, true);
    if (c->oracle_every_step && c->cfg.auto_reset) oracle_draw(c, c->d_oracle);                       // after the resets: what the next step starts from
}

int tg_step(tg_ctx* c, const float* actions, int32_t on_device) {
    if (!c || !actions) return fail(-1, "tg_step: NULL argument");
    TG_ENTER(c);
    const float* d_act = actions;
    if (!on_device) {
        TG_HIP(hipMemcpyAsync(c->d_actions, actions, (size_t)c->cfg.num_envs * c->act_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_act = c->d_actions;
    }
    // The launch sequence of a step is the same every step (all arguments are device pointers owned by the context, the action
    // buffer aside), so it is captured once and replayed as one graph launch.  Two graphs, neither ever re-captured:
    //   slot 0  reads the context's own action buffer: host actions are uploaded into it, and so are device actions that arrive in a
    //           buffer other than the pinned one below - a policy that hands over a fresh tensor every step costs one 8 KB device
    //           copy per step, not a graph instantiation;
    //   slot 1  reads the FIRST caller-owned device buffer seen, in place (a rollout that reuses one action tensor, e.g. bench.py).
    // Not while profiling (the per-kernel events are host calls between the launches) and not for the lane-per-env push kernels
    // (hipFuncSetAttribute on first launch).
    const bool want_graph = !c->profile && !c->graph_broken && c->cfg.env_kind != TG_ENV_OBJECT_PUSH && c->cfg.env_kind != TG_ENV_OBJECT_ROLL && step_as_graph(c);
    if (want_graph) {
        int slot = 0;
        if (on_device) {
            if (c->step_graph_actions[1] == nullptr && c->step_graph[1] == nullptr) c->step_graph_actions[1] = d_act;   // pin the first one
            if (c->step_graph_actions[1] == d_act) slot = 1;
            else {
                TG_HIP(hipMemcpyAsync(c->d_actions, actions, (size_t)c->cfg.num_envs * c->act_dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
                d_act = c->d_actions;
            }
        }
        if (!c->step_graph[slot]) {
            // Captured on a stream of its own, never on the stream the work runs on: while a stream is capturing, hipEventQuery on an event
            // that was recorded on it BEFORE the capture began fails with hipErrorCapturedEvent, and other threads do query such events - the
            // watchdog thread of torch's RCCL process group polls the end events of collectives that ran on the caller's stream, and a poll
            // that fell into the few hundred microseconds of a capture aborted the process (seen in 2 of ~100 one-rank bench runs).  A graph
            // is not tied to the stream it was captured on, so tg_set_stream needs no re-capture either.
            hipGraph_t g = nullptr;
            const hipStream_t run_stream = c->stream;
            c->stream = c->capture_stream;
            if (hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                enqueue_step(c, d_act);
                const hipError_t e1 = hipStreamEndCapture(c->capture_stream, &g);
                if (e1 == hipSuccess && g && hipGraphInstantiate(&c->step_graph[slot], g, nullptr, nullptr, 0) == hipSuccess) {
                    c->step_graph_stream[slot] = run_stream;
                } else {
                    c->step_graph[slot] = nullptr; c->graph_broken = true;
                }
                if (g) (void)hipGraphDestroy(g);
            } else {
                c->graph_broken = true;
            }
            c->stream = run_stream;
            (void)hipGetLastError();
        }
        if (c->step_graph[slot]) {
            TG_HIP(hipGraphLaunch(c->step_graph[slot], c->stream));
            bank_refill(c);
            return 0;
        }
    }
    enqueue_step(c, d_act);
    bank_refill(c);
    TG_HIP(hipGetLastError());
    return 0;
}

int tg_step_random(tg_ctx* c, uint64_t seed, uint64_t first_draw, int32_t restart) {
    if (!c) return fail(-1, "tg_step_random: NULL argument");
    TG_ENTER(c);
    if (!c->d_draw) { TG_HIP(hipMalloc(&c->d_draw, tg::kDrawWords * 8)); TG_HIP(hipMemset(c->d_draw, 0, tg::kDrawWords * 8)); restart = 1; }
    if (restart || seed != c->random_seed) {
        const unsigned long long h[4] = {first_draw, seed, 0ull, 0ull};    // the next step uses draw first_draw + 1
        TG_HIP(hipMemcpyAsync(c->d_draw, h, 32, hipMemcpyHostToDevice, c->stream));
        TG_HIP(hipStreamSynchronize(c->stream));
        c->random_seed = seed;
    }
    const int total = c->cfg.num_envs * c->act_dim;
    auto sample = [&]() {
        hipLaunchKernelGGL(k_sample_actions_ctr, dim3((total + 255) / 256), dim3(256), 0, c->stream, total, c->d_draw, (float)c->cfg.min_action,
                           (float)c->cfg.max_action, c->d_actions,
#ifdef TG_TL_STAMPS
                           c->st.tl
#else
                           (unsigned long long*)nullptr
#endif
                           );
    };
    const bool want_graph = !c->profile && !c->graph_broken && c->cfg.env_kind != TG_ENV_OBJECT_PUSH && c->cfg.env_kind != TG_ENV_OBJECT_ROLL && step_as_graph(c);
    // the lane-mapped k_step (edge_follow / surface_follow, TCP_velocity_control) draws its own actions: no sampler node at all - a dependent
    // kernel in this graph costs its ~6 us dispatch floor whatever it computes (profiles/r4_exp_reset_launch.txt)
    // (round 5: so does object_balance's k_step_body_wave - the conditions of launch_step_body_wave)
    const bool in_kernel = ((c->cfg.env_kind == TG_ENV_EDGE_FOLLOW || c->cfg.env_kind == TG_ENV_SURFACE_FOLLOW_AUTO) &&
                            c->cfg.control_mode == TG_CONTROL_TCP_VELOCITY && !use_arm_wave(c)) ||
                           (c->cfg.env_kind == TG_ENV_OBJECT_BALANCE && use_contact_wave(c) && c->cfg.physics_dtype == TG_PHYSICS_F64 &&
                            c->robot.topology == 0 && c->cfg.control_mode == TG_CONTROL_TCP_VELOCITY && !use_fused_step(c));
    struct DrawScope {                   // c->st carries the counter only while this call enqueues / captures
        tg_ctx* c; bool on;
        DrawScope(tg_ctx* c_, bool on_) : c(c_), on(on_) { if (on) { c->st.draw = c->d_draw; c->st.act_out = c->d_actions; } }
        ~DrawScope() { if (on) { c->st.draw = nullptr; c->st.act_out = nullptr; } }
    } scope(c, in_kernel);
    if (want_graph && !c->random_graph_t[c->obs_sel]) {
        hipGraph_t g = nullptr;
        const hipStream_t run_stream = c->stream;
        c->stream = c->capture_stream;
        if (hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            if (!in_kernel) sample();
            enqueue_step(c, c->d_actions);
            const hipError_t e1 = hipStreamEndCapture(c->capture_stream, &g);
            if (!(e1 == hipSuccess && g && hipGraphInstantiate(&c->random_graph_t[c->obs_sel], g, nullptr, nullptr, 0) == hipSuccess)) { c->random_graph_t[c->obs_sel] = nullptr; c->graph_broken = true; }
            if (g) (void)hipGraphDestroy(g);
        } else c->graph_broken = true;
        c->stream = run_stream;
        (void)hipGetLastError();
    }
    if (want_graph && c->random_graph_t[c->obs_sel]) {
        TG_HIP(hipGraphLaunch(c->random_graph_t[c->obs_sel], c->stream));
        bank_refill(c);
        return 0;
    }
    if (!in_kernel) sample();
    enqueue_step(c, c->d_actions);
    bank_refill(c);
    TG_HIP(hipGetLastError());
    return 0;
}

int tg_set_obs_targets(tg_ctx* c, int32_t count, void* const* dev_ptrs) {
    if (!c || count < 0 || count > 2 || (count > 0 && !dev_ptrs)) return fail(-1, "tg_set_obs_targets: bad argument");
    for (int k = 0; k < count; ++k) if (!dev_ptrs[k]) return fail(-1, "tg_set_obs_targets: NULL target");   // (every argument is checked before anything is changed)
    TG_ENTER(c);
    TG_HIP(hipStreamSynchronize(c->stream));
    for (int t = 1; t < 3; ++t) {        // the old targets' graphs name their buffers: gone with them
        for (int k = 0; k < 2; ++k) if (c->step_graph_t[t][k]) { (void)hipGraphExecDestroy(c->step_graph_t[t][k]); c->step_graph_t[t][k] = nullptr; }
        if (c->random_graph_t[t]) { (void)hipGraphExecDestroy(c->random_graph_t[t]); c->random_graph_t[t] = nullptr; }
    }
    c->obs_sel = 0; c->step_graph = c->step_graph_t[0];
    const size_t regions = (c->rp.W % 128 == 0 && c->rp.H % 128 == 0) ? (size_t)(c->rp.W / 128) * (c->rp.H / 128) : 0;
    for (int k = 0; k < 2; ++k) {
        c->obs_ext[k] = k < count ? (uint8_t*)dev_ptrs[k] : nullptr;
        if (k < count && c->rp.drawn != nullptr && regions == 0) return fail(-3, "tg_set_obs_targets: a changed-block record without 128-pixel regions (internal)");
        if (k < count && c->rp.drawn != nullptr) {   // a changed-block record of its own: nothing in that buffer is known to hold the untouched-sensor image
            if (!c->drawn_ext[k]) TG_HIP(hipMalloc(&c->drawn_ext[k], (size_t)c->cfg.num_envs * regions * 8));
            TG_HIP(hipMemset(c->drawn_ext[k], 0xFF, (size_t)c->cfg.num_envs * regions * 8));
        }
    }
    return 0;
}
int tg_select_obs_target(tg_ctx* c, int32_t index) {
    if (!c || index < 0 || index > 2 || (index > 0 && c->obs_ext[index - 1] == nullptr)) return fail(-1, "tg_select_obs_target: no such target");
    c->obs_sel = index;
    c->step_graph = c->step_graph_t[index];
    return 0;
}

int tg_get_tile_template(tg_ctx* c, void** p) {
    if (!c || !p) return fail(-1, "NULL argument");
    *p = c->d_tile_tmpl;
    return 0;
}
int tg_pack_interior(tg_ctx* c, void* dst_dev) {
    if (!c || !dst_dev) return fail(-1, "NULL argument");
    TG_ENTER(c);
    if (c->cfg_turn_off_border || c->n_interior == 0) return fail(-1, "tg_pack_interior: no constant border ring");
    const int K = c->n_interior, HW = c->H * c->W / 4;     // in 4-pixel words
    hipLaunchKernelGGL(k_pack_interior, dim3((K / 4 + 255) / 256, c->cfg.num_envs), dim3(256), 0, c->stream, (const uint32_t*)obs_buf(c), c->d_int_idx, K, HW,
                       c->cfg.num_envs, (uint32_t*)dst_dev);
    TG_HIP(hipGetLastError());
    return 0;
}
int tg_unpack_interior(tg_ctx* c, const void* src_dev, int32_t n_images, void* dst_dev) {
    if (!c || !src_dev || !dst_dev || n_images <= 0) return fail(-1, "bad argument");
    TG_ENTER(c);
    if (c->cfg_turn_off_border || c->n_interior == 0) return fail(-1, "tg_unpack_interior: no constant border ring");
    if (n_images > 65535) return fail(-1, "tg_unpack_interior: at most 65535 images per call");
    const int K = c->n_interior, HW = c->H * c->W / 4;     // in 4-pixel words
    hipLaunchKernelGGL(k_unpack_interior, dim3((HW / 4 + 255) / 256, n_images), dim3(256), 0, c->stream, (const uint32_t*)src_dev, c->d_int_rank,
                       (const uint32_t*)c->d_nodef_gray, K, HW, n_images, (uint32_t*)dst_dev);
    TG_HIP(hipGetLastError());
    return 0;
}
int tg_enable_oracle_obs(tg_ctx* c) {
    if (!c) return fail(-1, "NULL ctx");
    TG_ENTER(c);
    if (c->step_graph[0] || c->step_graph[1]) return fail(-1, "tg_enable_oracle_obs: call before the first tg_step");
    if (!c->d_oracle) TG_HIP(hipMalloc(&c->d_oracle, (size_t)c->cfg.num_envs * 34 * sizeof(float)));
    if (!c->d_oracle_term) TG_HIP(hipMalloc(&c->d_oracle_term, (size_t)c->cfg.num_envs * 34 * sizeof(float)));
    TG_HIP(hipMemset(c->d_oracle, 0, (size_t)c->cfg.num_envs * 34 * sizeof(float)));
    TG_HIP(hipMemset(c->d_oracle_term, 0, (size_t)c->cfg.num_envs * 34 * sizeof(float)));
    c->oracle_every_step = true;
    return 0;
}
int tg_get_obs_oracle_terminal(tg_ctx* c, void** p) {
    if (!c || !p) return fail(-1, "NULL argument");
    if (!c->oracle_every_step) return fail(-1, "tg_get_obs_oracle_terminal: needs tg_enable_oracle_obs");
    *p = c->d_oracle_term;
    return 0;
}
int tg_copy_obs_oracle_terminal(tg_ctx* c, float* dst) {
    if (!c || !dst) return fail(-1, "NULL argument");
    TG_ENTER(c);
    if (!c->oracle_every_step) return fail(-1, "tg_copy_obs_oracle_terminal: needs tg_enable_oracle_obs");
    TG_HIP(hipMemcpyAsync(dst, c->d_oracle_term, (size_t)c->cfg.num_envs * oracle_dim(c) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int tg_get_obs_oracle(tg_ctx* c, void** p, int32_t* dim) {
    if (!c || !p) return fail(-1, "NULL argument");
    TG_ENTER(c);
    const int d = oracle_dim(c);
    if (!c->d_oracle) TG_HIP(hipMalloc(&c->d_oracle, (size_t)c->cfg.num_envs * 34 * sizeof(float)));
    if (!c->oracle_every_step) {     // (enabled: tg_step / tg_reset have already written it)
#define CALL(T, TOPO) launch_oracle_obs_t<T, TOPO>(c, d, c->d_oracle)
        TG_DISPATCH(c->cfg.physics_dtype, c->robot.topology, CALL);
#undef CALL
    }
    TG_HIP(hipGetLastError());
    *p = c->d_oracle;
    if (dim) *dim = d;
    return 0;
}
int tg_copy_obs_oracle(tg_ctx* c, float* dst) {
    if (!c || !dst) return fail(-1, "NULL argument");
    void* p = nullptr; int32_t d = 0;
    if (int rc = tg_get_obs_oracle(c, &p, &d)) return rc;
    TG_HIP(hipMemcpyAsync(dst, p, (size_t)c->cfg.num_envs * d * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

static int set_scene_impl(tg_ctx* c, const tg_scene* sc) {
    if (!c || !sc) return fail(-1, "tg_set_scene: NULL argument");
    TG_ENTER(c);
    if (c->scene_on) return fail(-1, "tg_set_scene: the scene is already set");
    if (sc->every_step && (c->step_graph[0] || c->step_graph[1])) return fail(-1, "tg_set_scene: every_step needs the call before the first tg_step");
    if (!sc->verts || !sc->tris || !sc->tri_frame || !sc->tri_rgb || sc->n_tris <= 0 || sc->n_verts <= 0) return fail(-1, "tg_set_scene: empty scene");
    const int W = sc->image_w, H = sc->image_h;
    if (W <= 0 || H <= 0 || (W > 128 && W % 128) || (H > 128 && H % 128)) return fail(-1, "tg_set_scene: image sides must be <= 128 or multiples of 128");
    if (!(sc->near_plane > 0 && sc->far_plane > sc->near_plane && sc->fov_deg > 0 && sc->fov_deg < 180)) return fail(-1, "tg_set_scene: bad projection");
    const int n_frames = c->robot.ndof + 2;
    if (n_frames > 16) return fail(-1, "tg_set_scene: too many frames");
    std::vector<uint32_t> attr(sc->n_tris);
    for (int t = 0; t < sc->n_tris; ++t) {
        if (sc->tri_frame[t] >= n_frames) return fail(-1, "tg_set_scene: tri_frame out of range");
        for (int k = 0; k < 3; ++k)
            if (sc->tris[3 * t + k] < 0 || sc->tris[3 * t + k] >= sc->n_verts) return fail(-1, "tg_set_scene: vertex index out of range");
        attr[t] = ((uint32_t)sc->tri_frame[t] << 24) | ((uint32_t)sc->tri_rgb[3 * t] << 16) | ((uint32_t)sc->tri_rgb[3 * t + 1] << 8) | sc->tri_rgb[3 * t + 2];
    }
    // computeViewMatrixFromYawPitchRoll(target, distance, yaw, pitch, roll = 0, upAxisIndex = 2) [A31]: eye = target + Rz(yaw) Rx(pitch)
    // (0, -distance, 0), up = Rz(yaw) Rx(pitch) (0, 0, 1), then the look-at matrix
    const double d2r = 3.14159265358979323846 / 180.0, cy = cos(sc->cam_yaw_deg * d2r), sy = sin(sc->cam_yaw_deg * d2r),
                 cp = cos(sc->cam_pitch_deg * d2r), sp = sin(sc->cam_pitch_deg * d2r);
    // Rz Rx = [[cy, -sy cp, sy sp], [sy, cy cp, -cy sp], [0, sp, cp]]
    const double E1[3] = {-sy * cp, cy * cp, sp}, E2[3] = {sy * sp, -cy * sp, cp};
    double eye[3], f[3], up[3] = {E2[0], E2[1], E2[2]};
    for (int k = 0; k < 3; ++k) { eye[k] = sc->cam_target[k] - sc->cam_dist * E1[k]; f[k] = sc->cam_target[k] - eye[k]; }
    auto normalize = [](double (&v)[3]) { const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (double& x : v) x /= n; };
    normalize(f);
    double s_[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    normalize(s_);
    const double u[3] = {s_[1] * f[2] - s_[2] * f[1], s_[2] * f[0] - s_[0] * f[2], s_[0] * f[1] - s_[1] * f[0]};
    SceneView& V = c->scene_view;
    for (int k = 0; k < 3; ++k) { V.R[k] = s_[k]; V.R[3 + k] = u[k]; V.R[6 + k] = -f[k]; }
    for (int r = 0; r < 3; ++r) V.t[r] = -(V.R[3 * r] * eye[0] + V.R[3 * r + 1] * eye[1] + V.R[3 * r + 2] * eye[2]);
    SceneParams P = make_scene_params(W, H, sc->fov_deg, sc->near_plane, sc->far_plane);
    double L[3] = {sc->light_dir[0], sc->light_dir[1], sc->light_dir[2]};
    normalize(L);
    for (int r = 0; r < 3; ++r) P.light_eye[r] = (float)(V.R[3 * r] * L[0] + V.R[3 * r + 1] * L[1] + V.R[3 * r + 2] * L[2]);
    for (int k = 0; k < 3; ++k) P.background[k] = sc->background[k];
    P.n_tris = sc->n_tris; P.n_frames = n_frames;
    if (sc->body_heightfield) {
        if (c->cfg.env_kind != TG_ENV_SURFACE_FOLLOW_AUTO) return fail(-1, "tg_set_scene: body_heightfield needs a surface env");
        P.hf_heights = c->st.heights; P.hf_zoff = c->st.surf_zoff; P.hf_sel = c->st.hsel; P.hf_n = c->cfg.num_envs; P.hf_rows = c->cfg.surf_rows; P.hf_cols = c->cfg.surf_cols;
        P.hf_scale = (float)c->cfg.surf_grid_scale;
        P.hf_rgb = ((uint32_t)sc->body_rgb[0] << 16) | ((uint32_t)sc->body_rgb[1] << 8) | sc->body_rgb[2];
    }
    const size_t n = (size_t)c->cfg.num_envs, img = (size_t)W * H * 3;
    std::vector<int32_t> tris(sc->tris, sc->tris + (size_t)sc->n_tris * 3);
    std::vector<SceneChunk> chunks;
    std::vector<float> cverts;            // chunk-ordered vertex copies (tg_scene.h)
    std::vector<uint32_t> tri_local;
    build_scene_chunks(sc->verts, tris.data(), attr.data(), sc->n_tris, chunks, cverts, tri_local);
    if (chunks.size() > 8192) return fail(-1, "tg_set_scene: too many triangle chunks");
    TG_HIP(hipMalloc(&c->d_scene_chunks, chunks.size() * sizeof(SceneChunk)));
    TG_HIP(hipMemcpy(c->d_scene_chunks, chunks.data(), chunks.size() * sizeof(SceneChunk), hipMemcpyHostToDevice));
    P.chunks = c->d_scene_chunks; P.n_chunks = (int)chunks.size();
    TG_HIP(hipMalloc(&c->d_scene_verts, cverts.size() * 4 + 16)); TG_HIP(hipMalloc(&c->d_scene_tris, (size_t)sc->n_tris * 12));
    TG_HIP(hipMalloc(&c->d_scene_local, (size_t)sc->n_tris * 4));
    TG_HIP(hipMalloc(&c->d_scene_attr, (size_t)sc->n_tris * 4)); TG_HIP(hipMalloc(&c->d_scene_xf, n * n_frames * 12 * 4));
    TG_HIP(hipMalloc(&c->d_vis, n * img)); TG_HIP(hipMalloc(&c->d_vis_term, n * img));
    TG_HIP(hipMemcpy(c->d_scene_verts, cverts.data(), cverts.size() * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(c->d_scene_local, tri_local.data(), (size_t)sc->n_tris * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(c->d_scene_tris, tris.data(), (size_t)sc->n_tris * 12, hipMemcpyHostToDevice));
    TG_HIP(hipMemcpy(c->d_scene_attr, attr.data(), (size_t)sc->n_tris * 4, hipMemcpyHostToDevice));
    TG_HIP(hipMemset(c->d_vis, 0, n * img)); TG_HIP(hipMemset(c->d_vis_term, 0, n * img));
    P.verts = c->d_scene_verts; P.tris = c->d_scene_tris; P.tri_attr = c->d_scene_attr; P.tri_local = c->d_scene_local;
    {   // the env's translucent visuals (goal indicator / trajectory markers), one slot list per env, filled by k_scene_xf
        const int kind = c->cfg.env_kind;
        P.n_spheres = 1 + (kind == TG_ENV_OBJECT_PUSH ? c->cfg.traj_n_points : (kind == TG_ENV_OBJECT_BALANCE ? 0 : 1));   // slot 0: the arm's TCP marker
        if (P.n_spheres > 16) return fail(-1, "tg_set_scene: more than 16 trajectory markers");
        if (P.n_spheres > 0) {
            TG_HIP(hipMalloc(&c->d_scene_spheres, n * (size_t)P.n_spheres * 8 * 4));
            TG_HIP(hipMemset(c->d_scene_spheres, 0, n * (size_t)P.n_spheres * 8 * 4));
        }
        P.spheres = c->d_scene_spheres;
    }
    if (scene_prepare(P) != 0) return fail(-1, "tg_set_scene: the scene's chunk list does not fit the workgroup's LDS (or hipFuncSetAttribute failed)");
    {   // the world frame once: frame 0 of every env is the view matrix itself (k_scene_xf: put(0, I, 0)), rounded to float the same way
        std::vector<float> xf0((size_t)n_frames * 12, 0.0f);
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc) xf0[3 * r + cc] = (float)(V.R[3 * r + 0] * I[cc] + V.R[3 * r + 1] * I[3 + cc] + V.R[3 * r + 2] * I[6 + cc]);
            xf0[9 + r] = (float)(V.R[3 * r + 0] * z[0] + V.R[3 * r + 1] * z[1] + V.R[3 * r + 2] * z[2] + V.t[r]);
        }
        TG_HIP(hipMalloc(&c->d_scene_static, (size_t)W * H * 8));
        TG_HIP(hipMemcpy(c->d_scene_xf, xf0.data(), xf0.size() * 4, hipMemcpyHostToDevice));
        launch_scene_static(P, c->d_scene_xf, c->d_scene_static, c->stream);
        TG_HIP(hipStreamSynchronize(c->stream));
        P.static_keys = c->d_scene_static;
    }
    c->scene = P;
    c->scene_on = true;
    c->scene_every_step = sc->every_step != 0;
    return 0;
}


int tg_set_scene(tg_ctx* c, const tg_scene* sc) {
    const int rc = set_scene_impl(c, sc);
    if (rc != 0 && c && !c->scene_on) {     // a failed set-up leaves nothing behind: a retry starts from null pointers, nothing leaks
        const std::string keep = tg_last_error();
        (void)hipSetDevice(c->cfg.device);
        void** ptrs[] = {(void**)&c->d_scene_chunks, (void**)&c->d_scene_verts, (void**)&c->d_scene_tris, (void**)&c->d_scene_local, (void**)&c->d_scene_attr,
                         (void**)&c->d_scene_xf, (void**)&c->d_vis, (void**)&c->d_vis_term, (void**)&c->d_scene_static, (void**)&c->d_scene_spheres};
        for (void** p : ptrs) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        (void)hipGetLastError();
        tg::report_error(rc, keep.c_str());
    }
    return rc;
}

int tg_render_scene(tg_ctx* c) {
    if (!c) return fail(-1, "NULL ctx");
    TG_ENTER(c);
    if (!c->scene_on) return fail(-1, "tg_render_scene: no scene (tg_set_scene)");
    scene_draw(c, nullptr, false);
    TG_HIP(hipGetLastError());
    return 0;
}

int tg_get_obs_visual(tg_ctx* c, void** p, int32_t terminal) {
    if (!c || !p) return fail(-1, "NULL argument");
    if (!c->scene_on) return fail(-1, "tg_get_obs_visual: no scene (tg_set_scene)");
    *p = terminal ? c->d_vis_term : c->d_vis;
    return 0;
}

int tg_copy_obs_visual(tg_ctx* c, uint8_t* dst, int32_t terminal) {
    if (!c || !dst) return fail(-1, "NULL argument");
    TG_ENTER(c);
    if (!c->scene_on) return fail(-1, "tg_copy_obs_visual: no scene (tg_set_scene)");
    TG_HIP(hipMemcpyAsync(dst, terminal ? c->d_vis_term : c->d_vis, (size_t)c->cfg.num_envs * c->scene.W * c->scene.H * 3, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_sync(tg_ctx* c) {
    if (!c) return fail(-1, "NULL ctx");
    TG_ENTER(c);
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

int tg_get_obs_tactile(tg_ctx* c, void** p) { if (!c || !p) return fail(-1, "NULL argument"); *p = obs_buf(c); return 0; }
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
    const bool has = env_has_feature(c->cfg.env_kind);
    *feature_off = has ? (int64_t)c->packed_feature_off : -1;
    if (dim) *dim = has ? 12 : 0;
    return 0;
}
int tg_sample_actions(tg_ctx* c, uint64_t seed, uint64_t counter, float* dev_actions) {
    if (!c || !dev_actions) return fail(-1, "tg_sample_actions: NULL argument");
    const int total = c->cfg.num_envs * c->act_dim;
    hipLaunchKernelGGL(k_sample_actions, dim3((total + 255) / 256), dim3(256), 0, c->stream, total, seed, counter, (float)c->cfg.min_action,
                       (float)c->cfg.max_action, dev_actions,
#ifdef TG_TL_STAMPS
                       c->st.tl
#else
                       (unsigned long long*)nullptr
#endif
                       );
    TG_HIP(hipGetLastError());
    return 0;
}
int tg_get_obs_feature(tg_ctx* c, void** p, int32_t* dim, int32_t terminal) {
    if (!c || !p) return fail(-1, "NULL argument");
    if (!env_has_feature(c->cfg.env_kind))
        return fail(-1, "tg_get_obs_feature: this env has no extended_feature observation");
    *p = terminal ? c->st.term_feature : c->st.feature;
    if (dim) *dim = 12;
    return 0;
}
int tg_copy_obs_feature(tg_ctx* c, float* dst, int32_t terminal) {
    if (!c || !dst) return fail(-1, "NULL argument");
    TG_ENTER(c);
    if (!env_has_feature(c->cfg.env_kind))
        return fail(-1, "tg_copy_obs_feature: this env has no extended_feature observation");
    TG_HIP(hipMemcpyAsync(dst, terminal ? c->st.term_feature : c->st.feature, (size_t)c->cfg.num_envs * 12 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int tg_get_reward_done(tg_ctx* c, float* reward, uint8_t* done) {
    if (!c) return fail(-1, "NULL ctx");
    TG_ENTER(c);
    const int n = c->cfg.num_envs;
    if (reward) TG_HIP(hipMemcpyAsync(reward, c->st.reward, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (done) TG_HIP(hipMemcpyAsync(done, c->st.done, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int tg_copy_obs_tactile(tg_ctx* c, uint8_t* dst, int32_t terminal) {
    if (!c || !dst) return fail(-1, "NULL argument");
    TG_ENTER(c);
    TG_HIP(hipMemcpyAsync(dst, terminal ? c->d_term : obs_buf(c), (size_t)c->cfg.num_envs * c->H * c->W, hipMemcpyDeviceToHost, c->stream));
    TG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tg_copy_obs_rows(tg_ctx* c, int32_t visual, int32_t terminal, const int32_t* env_ids, int32_t count, uint8_t* dst) {
    if (!c || count < 0 || (count > 0 && (!env_ids || !dst))) return fail(-1, "tg_copy_obs_rows: bad argument");
    TG_ENTER(c);
    if (visual && !c->scene_on) return fail(-1, "tg_copy_obs_rows: no scene (tg_set_scene)");
    const size_t img = visual ? (size_t)c->scene.W * c->scene.H * 3 : (size_t)c->H * c->W;
    const uint8_t* src = visual ? (terminal ? c->d_vis_term : c->d_vis) : (terminal ? c->d_term : obs_buf(c));
    for (int32_t k = 0; k < count; ++k)
        if (env_ids[k] < 0 || env_ids[k] >= c->cfg.num_envs) return fail(-1, "tg_copy_obs_rows: env id out of range");
    // through a pinned staging block (64 images at a time): a device -> pageable copy of 16 KB is a staged, blocking copy of its own each time
    constexpr int kChunk = 64;
    if (c->h_rows_bytes < img * kChunk) {
        if (c->h_rows) { (void)hipHostFree(c->h_rows); c->h_rows = nullptr; c->h_rows_bytes = 0; }
        TG_HIP(hipHostMalloc((void**)&c->h_rows, img * kChunk, hipHostMallocDefault));
        c->h_rows_bytes = img * kChunk;
    }
    for (int32_t k0 = 0; k0 < count; k0 += kChunk) {
        const int32_t m = count - k0 < kChunk ? count - k0 : kChunk;
        for (int32_t k = 0; k < m; ++k)
            TG_HIP(hipMemcpyAsync(c->h_rows + (size_t)k * img, src + (size_t)env_ids[k0 + k] * img, img, hipMemcpyDeviceToHost, c->stream));
        TG_HIP(hipStreamSynchronize(c->stream));
        memcpy(dst + (size_t)k0 * img, c->h_rows, (size_t)m * img);
    }
    return 0;
}

}  // extern "C"
