/*
 * tactile_gym_hip.h — C ABI of libtactile_gym_hip.so: the MI355X-native vectorised tactile-env step.
 *
 * The reference (ac-93/tactile_gym) has no FFI layer of its own: its hot path is Python calling the PyBullet C
 * extension.  This header is therefore the boundary a maintainer would bind *instead of* those PyBullet calls;
 * every entry point cites the reference call sites it replaces (paths relative to tactile_gym/).  All arguments
 * are plain pointers and sizes; no torch or HIP types appear in signatures (streams and device pointers travel as
 * `void*`).  Every function returns 0 on success or a negative error code; tg_last_error() gives the message
 * (the reference calls sys.exit(msg) on bad mode strings — base_tactile_env.py:264, robot.py:65,174 — the Python
 * host layer turns a non-zero status into an exception).
 *
 * Threading: one context per (process, GPU); calls on one context are not re-entrant (the reference is single
 * threaded per env with one PyBullet client per process, base_tactile_env.py:51,59).  tg_step() only enqueues work
 * on the context's stream; tg_sync() waits — they map to SB3's VecEnv.step_async()/step_wait().
 */
#ifndef TACTILE_GYM_HIP_H
#define TACTILE_GYM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_MAX_DOF 8
#define TG_MAX_BODIES_PER_LINK 4
#define TG_ABI_VERSION 14
#define TG_MAX_TRAJ_POINTS 16

/* ---- robot description: the flattened URDF (replaces loadURDF, robots/arms/robot.py:95-112) --------------------- */
typedef struct {
    int32_t ndof;
    int32_t topology;                       /* 0: serial chain (UR5), 1: MG400 tree {-1,0,1,2,3,0,5,6} */
    double joint_pos[TG_MAX_DOF][3];        /* joint origin in parent link frame */
    double joint_rot[TG_MAX_DOF][9];        /* joint frame in parent link frame, row-major */
    double joint_axis[TG_MAX_DOF][3];
    /* rigid bodies welded to each moving link (URDF links incl. fixed children); mass 0 = empty slot */
    double body_mass[TG_MAX_DOF][TG_MAX_BODIES_PER_LINK];
    double body_com[TG_MAX_DOF][TG_MAX_BODIES_PER_LINK][3];
    double body_rot[TG_MAX_DOF][TG_MAX_BODIES_PER_LINK][9];
    double body_inertia[TG_MAX_DOF][TG_MAX_BODIES_PER_LINK][3];
    /* frames the reference reads with getLinkState (PyBullet inertial-frame convention) */
    int32_t tcp_link;     double tcp_pos[3];    double tcp_rot[9];     /* base_robot_arm.py:136-151 */
    int32_t sensor_link;  double sensor_pos[3]; double sensor_rot[9];  /* tactile_sensor.py:153-155 */
    double gravity[3];                      /* base_tactile_env.py:126 */
    double linear_damping, angular_damping; /* base_robot_arm.py:24 */
    double joint_damping;                   /* base_robot_arm.py:25 */
    double max_force, pos_gain, vel_gain;   /* ur5.py:19-21, mg400.py:27-29 */
    double rest_q[TG_MAX_DOF];              /* rest_poses.py (movable joints) */
} tg_robot;

/* ---- tactile sensor: camera + reference images (sensors/tactile_sensor.py:63-80,127-187) ------------------------- */
typedef struct {
    int32_t image_h, image_w;
    double cam_pos[3];                      /* camera mount in the sensor-body frame */
    double cam_rpy[3];
    double fov_deg, near_plane, far_plane;
    int32_t turn_off_border;
    const float*   nodef_dep;               /* host pointers, [h*w]; copied at tg_create */
    const float*   nodef_gray;
    const uint8_t* border_mask;
} tg_sensor;

/* ---- stimulus mesh seen by the tactile camera (edge_follow_env.py:218-235 loadURDF of the edge) ------------------ */
typedef struct {
    int32_t n_verts, n_tris;
    const float*   verts;                   /* host, [n_verts][3], object frame */
    const int32_t* tris;                    /* host, [n_tris][3] */
} tg_mesh;

enum { TG_ENV_EDGE_FOLLOW = 0, TG_ENV_SURFACE_FOLLOW_AUTO = 1, TG_ENV_OBJECT_BALANCE = 2, TG_ENV_OBJECT_PUSH = 3, TG_ENV_OBJECT_ROLL = 4 };
enum { TG_MOVE_XY = 0, TG_MOVE_XYZ = 1, TG_MOVE_XYRZ = 2, TG_MOVE_XYZRZ = 3 };            /* edge_follow_env.py:345-369 */
enum { TG_SMOVE_YZ = 0, TG_SMOVE_XYZ = 1, TG_SMOVE_YZRX = 2, TG_SMOVE_XYZRXRY = 3, TG_SMOVE_XRZ = 4 };   /* surface_follow_auto_env.py:27-57; xRz: surface_follow_vert_env.py:29-48 */
enum { TG_BMOVE_XY = 0, TG_BMOVE_XYZ = 1, TG_BMOVE_RXRY = 2, TG_BMOVE_XYRXRY = 3 };         /* object_balance_env.py:398-424 */
enum { TG_PMOVE_Y = 0, TG_PMOVE_YRZ = 1, TG_PMOVE_XYRZ = 2, TG_PMOVE_TYRZ = 3, TG_PMOVE_TXTYRZ = 4 }; /* object_push_env.py:372-454 */
enum { TG_TRAJ_SIMPLEX = 0, TG_TRAJ_STRAIGHT = 1 };                                          /* object_push_env.py:248-313 */
enum { TG_NOISE_FIXED_HEIGHT = 0, TG_NOISE_RAND_HEIGHT = 1 };                               /* edge_follow noise_mode */
enum { TG_SNOISE_SIMPLEX = 0, TG_SNOISE_NONE = 1, TG_SNOISE_RANDOM = 2, TG_SNOISE_VERTICAL_SIMPLEX = 3 };                   /* surface_follow noise_mode (base_surface_env.py:448-471) */
enum { TG_REWARD_DENSE = 0, TG_REWARD_SPARSE = 1 };
enum { TG_PHYSICS_F64 = 0, TG_PHYSICS_F32 = 1 };
enum { TG_CONTACT_MAP_AUTO = 0, TG_CONTACT_MAP_LANE = 1, TG_CONTACT_MAP_WAVE = 2 };
enum { TG_CONTROL_TCP_VELOCITY = 0, TG_CONTROL_TCP_POSITION = 1 };                             /* robot.py:156-186 apply_action */

/* ---- task + engine configuration (env ctor kwargs / env_modes, edge_follow_env.py:23-134) ------------------------- */
typedef struct {
    int32_t abi_version;                    /* TG_ABI_VERSION */
    int32_t env_kind;                       /* TG_ENV_* */
    int32_t num_envs;
    int32_t max_steps;                      /* episode length cap */
    int32_t movement_mode, noise_mode, reward_mode;
    int32_t physics_dtype;                  /* TG_PHYSICS_* */
    int32_t action_repeat;                  /* sim ticks per env step: 24 (edge_follow_env.py:35-37) */
    int32_t solver_iterations;              /* 150 (base_tactile_env.py:128-130) */
    int32_t auto_reset;                     /* VecEnv semantics: reset finished envs inside tg_step */
    int32_t device;                         /* HIP device ordinal */
    double sim_dt;                          /* 1/240 */
    double min_action, max_action;          /* edge_follow_env.py:140 */
    double act_lo[6], act_hi[6];            /* per-dimension physical ranges, :143-166 */
    double tcp_lims[6][2];                  /* :75-90 */
    double workframe_pos[3], workframe_rpy[3]; /* :106-107 */
    double stim_pos[3];                     /* edge_pos :201 */
    double edge_height, edge_len;           /* :203-207 */
    double termination_dist;                /* :67 */
    double embed_dist, embed_lo, embed_hi;  /* :94-99, :291-298 */
    /* surface_follow (base_surface_env.py:234-282): heightfield stimulus generated per episode from OpenSimplex noise.
     * stim_pos = surface_pos; rows x cols grid of pitch surf_grid_scale; height = noise2(i*interp, j*interp) * range. */
    int32_t surf_rows, surf_cols;           /* 64, 64 (:240-243) */
    int32_t surf_center_z;                  /* 1: Bullet's heightfield shape is centred on (min+max)/2 [PARITY_ASSUMPTIONS A15] */
    int32_t pgs_full_sweeps;                /* 0 (default): the motor-row PGS leaves the loop once no impulse can change in its last bit
                                             * (Bullet's zero-residual exit, PARITY_ASSUMPTIONS A7b); 1: always run solver_iterations sweeps */
    double surf_grid_scale;                 /* 0.006 (:238) */
    double surf_height_range;               /* 0.025 (:239) */
    double surf_interp;                     /* 0.05  (:244) */
    double surf_xy_extent;                  /* 0.15  (:245) goal distance and TCP xy limits */
    double auto_action_scale;               /* 1.0 tactip / 0.9 digitac / 0.7 digit (surface_follow_auto_env.py:33-41) */
    /* object_balance (object_balance_env.py): a free rigid body (stimulus mesh = its visual triangles, in the body's base
     * inertial frame) tied to the TCP link by a point-to-point constraint. embed_lo/hi: rand_embed_dist range (:308-316). */
    int32_t rand_gravity, rand_embed;       /* env_modes flags (:39-41) */
    double gravity_lo, gravity_hi, gravity_default;   /* U(-1.0, -0.1) or -0.1 (:301-306) */
    double obj_mass;
    double obj_com[3];                      /* composite centre of mass in the base inertial frame */
    double obj_inertia[9];                  /* about obj_com, base axes, row major */
    double obj_root_inertial_pos[3];        /* root link's inertial origin in its link frame (loadURDF places the link frame) */
    double obj_base_width, obj_base_height; /* 0.1, 0.0025 (:158-159) */
    double obj_init_rpy[3];                 /* (0, 0, -pi/2) (:190) */
    double ext_force[3];                    /* apply_random_force_base: (0, 0, -0.1) (:360-381) */
    double term_deg, term_pos;              /* 35 deg, 0.1 m (:50-51) */
    double p2p_erp, p2p_max_impulse;        /* 0.2, 500 [PARITY_ASSUMPTIONS A18-A19] */
    /* object_push (object_push_env.py): a free cube (obj_mass / obj_com / obj_inertia, stimulus mesh = its visual triangles)
     * on the table, pushed by the collision core of the sensor tip along a per-episode trajectory of goals.
     * termination_dist = termination_pos_dist (:57); obj_init_rpy = (-pi, 0, pi/2) (:158). */
    int32_t traj_type;                      /* TG_TRAJ_* (env_modes["traj_type"]) */
    int32_t traj_n_points;                  /* 10 (:229), <= TG_MAX_TRAJ_POINTS */
    int32_t rand_init_orn, rand_obj_mass;   /* env_modes flags (:168-192) */
    int32_t tip_link, n_tip_verts;          /* moving link carrying the tip's collision core; its convex-hull vertices */
    int32_t cone_friction;                  /* enableConeFriction=1 (base_tactile_env.py:128-130) */
    int32_t surf_goal_variant;              /* surface_follow: 0 = -v0 auto-drive (surface_follow_auto_env.py), 1 = -v1: every action dimension from the
                                             * agent, dense reward -(goal_xy + 10 surf + w_norm cos) (surface_follow_goal_env.py:27-90) */
    const double* tip_verts;                /* host, [n_tip_verts][3], link frame; copied at tg_create */
    double obj_half[3];                     /* cube half extents 0.04 (:45-46) */
    double obj_init_pos[3];                 /* (:160) */
    double table_z;                         /* top of the table: 0 */
    double mu_table, mu_tip;                /* combined friction: cube 0.065 x table 1.0 / x tip lateral friction (:50-56, :216-225) */
    double margin_cube, margin_tip;         /* collision margins [PARITY_ASSUMPTIONS A24] */
    double contact_breaking, contact_erp;   /* 1e-4 [A24], 0.2 */
    double tip_stiffness, tip_damping;      /* contactStiffness / contactDamping of the tip core (:50-56) */
    double obj_lin_damp, obj_ang_damp;      /* Bullet defaults 0.04 */
    double traj_spacing, traj_max_perturb, traj_init_offset;   /* 0.025, 0.1, obj_width/2 + spacing (:229-262) */
    double mass_lo, mass_hi;                /* rand_obj_mass U(0.4, 0.8) (:190-192) */
    double init_orn_range, traj_ang_range;  /* pi/32 (:170), pi/8 (:283) */
    /* control mode (env_modes["control_mode"], robot.py:156-186).  TG_CONTROL_TCP_POSITION: the action is a work-frame pose delta
     * (act_lo/hi = +-1 mm, +-1 deg, e.g. edge_follow_env.py:143-153); target = clip(current + delta, tcp_lims), inverse kinematics,
     * POSITION_CONTROL motors (base_robot_arm.py:228-279, mg400.py:131-190), then blocking_move(max_steps = max_blocking_steps,
     * constant_vel = None) (robot.py:188-260). */
    int32_t control_mode;                   /* TG_CONTROL_* */
    int32_t max_blocking_steps;             /* _max_blocking_pos_move_steps = 10 (edge_follow_env.py:38) */
    /* object_push: goal index right after reset.  reset() ends with get_step_data() (base_object_env.py:183-185), whose termination()
     * advances the goal when the cube is within termination_pos_dist of it (:520-537); the first goal sits exactly that far from the
     * cube's start position, so whether index 0 or 1 comes out is decided by double rounding of the work-frame constants.  The host
     * evaluates that comparison once, the way the reference would (PARITY_ASSUMPTIONS A29). */
    int32_t reset_goal_id;
    /* surface_follow-v2 (noise_mode vertical_simplex, movement xRz; the `vertical_simplex` branches of base_surface_env.py and
     * surface_follow_vert_env.py): the heightfield stands upright, rotated by rpy (0, -pi/2, 0) about stim_pos = surface_pos
     * (x, y, 0.15 + range); the bins, the goal and the reward follow the reference's flipped surface_array (:476-516, :735-758);
     * reward -(10 surf_dist + 3 cos_dist) (surface_follow_vert_env.py:66-81). */
    int32_t surf_vertical;
    /* object_roll (object_roll_env.py): a marble (sphere.urdf: obj_mass, roll_radius; stimulus mesh = its tessellation at that radius)
     * on the table under the flat TacTip, whose tip collides as a URDF cylinder; goal given in the TCP frame.  Shares with object_push:
     * obj_mass, mu_table, mu_tip, contact_breaking / erp, tip_stiffness / damping, obj_lin / ang_damp, tip_link, cone_friction, table_z,
     * termination_dist (:60).  workframe_pos z = 2 roll_radius - embed_dist is the default; per episode it is 2 r - embed with
     * r = roll_radius x U(1, 2) (rand_obj_size) and embed = U(embed_lo, embed_hi) (rand_embed) (:181-201).  Movement "xy" only.
     * State view: obj_mass = the episode's radius, goal_pos = the goal in the TCP frame. [PARITY_ASSUMPTIONS A30] */
    int32_t roll_rand_init_pos, roll_rand_size, roll_rand_embed;   /* env_modes flags (:41-43) */
    double roll_radius;                     /* default_obj_radius 0.0025 (:161) */
    double roll_init_range;                 /* 0.009 (:210-214) */
    double roll_goal_lo, roll_goal_hi;      /* goal distance U(0.005 | 0 with rand_init_pos, 0.015) (:256-259) */
    double tip_cyl_pos[3], tip_cyl_rot[9];  /* tip cylinder frame (axis z) in the frame of tip_link */
    double tip_cyl_half_len, tip_cyl_radius;
    /* object_push / object_roll: how the contact solve of stepSimulation (robot.py:141; 150 PGS sweeps, base_tactile_env.py:127-130) is
     * mapped onto the GPU.  TG_CONTACT_MAP_WAVE: one 64-lane wavefront per env, one solver row per lane (fills the chip at ~1024 envs);
     * TG_CONTACT_MAP_LANE: one lane per env (fewer instructions per env, better once every SIMD is busy); AUTO picks by num_envs.
     * Same solver, same row order; results agree to rounding (f64), contact sets exactly. */
    int32_t contact_mapping;                /* TG_CONTACT_MAP_* */
    /* edge_follow / surface_follow with auto_reset: the reset of these envs (edge_follow_env.py:311-336, base_surface_env.py:616-662,
     * robot.py:114-125) is a pure function of the env's RNG stream, so every env's NEXT post-reset state is computed ahead of time on a
     * second low-priority stream ("reset bank") and a finished env takes it inside the step instead of stalling the batch for its blocking
     * move.  Results are identical with the bank on, off, or not ready in time (the same reset code either way) - for edge_follow and
     * surface_follow.  object_balance has a reset TEMPLATE instead (the arm's post-reset state computed once, by env 0's first reset): against
     * TG_BANK_OFF, which recomputes every reset with the fallen object still on the constraint, the arm differs by the last-bit residue of that
     * tick's Gauss-Seidel (joints within 1e-13 rad, frames within 3 pixels: tests/test_gpu_reset_bank.py) - not bit-identical.
     * TG_BANK_AUTO = TG_BANK_ON (round 5; until then on for the MG400 only - its blocking move stalls a 1024-env batch for 9-11 ms per
     * full-batch reset, 0.126 -> 1.13 M env-steps/s on surface_follow-v2).  The UR5's reset is 0.1 ms: nothing when every env finishes in the
     * same step (a random rollout from a common start), but with episodes ending in different steps - any RL run - some env finishes in nearly
     * every step and the whole batch waits for it each time: 6.7 -> 18.3 M env-steps/s at 1024 envs (tools/desync_rate.py); the aligned
     * rollout costs the same either way.  With the bank on, a context's step calls keep the host at most ~70 steps ahead of the device (the
     * refill's pacing: one host-side wait for a marker every 8 steps).  TG_BANK_OFF: every reset on the spot; TG_BANK_SYNC: on, and the refill
     * is waited for after every step (tests: the bank is always ready).  The environment variable TG_RESET_BANK (0 / 1 / sync) overrides. */
    int32_t reset_bank;                     /* TG_BANK_* */
    /* object_push: narrowphase of the tip-cube pair (stepSimulation's collision detection, robot.py:141).  TG_NARROW_CLOSED_FORM: the
     * deepest point of the tip's convex hull against the cube's faces in closed form, ONE contact point per tick (PARITY A24);
     * TG_NARROW_GJK_MANIFOLD: support-mapping GJK distance + EPA penetration on the hull and the box, fed into a persistent manifold of up
     * to 4 points with Bullet's add / replace / break rules (PARITY A35-A38); TG_NARROW_GJK_SINGLE: the same GJK / EPA, the tick's point only
     * (no cache): the closed form's special case - equal to it wherever the closest features are a hull vertex and a box face (tests).
     * Both run on the wave mapping (f64, cone friction). */
    int32_t narrowphase;                    /* TG_NARROW_* */
    /* object_balance, object_mode "ball_on_plate" (object_balance_env.py:187-199, 241-260): the free body is the round plate (obj_mass /
     * obj_com / obj_inertia, obj_base_width 0.2, tied to the TCP as the pole is) and a ball rolls on it: sphere.urdf x globalScaling 7.5
     * (mass 0.05 unscaled, radius 0.0025 x 7.5), lateralFriction 10 x the plate's default 0.5, reset to workframe + (0, 0, radius) with a
     * one-shot torque 0.001 x (U(-1,1), U(-1,1), 0) (:350-353, 393-401).  Contact: ball against the plate's solid cylinder (plate_radius,
     * half length obj_base_height / 2), contact_breaking / contact_erp / obj_lin_damp / obj_ang_damp as for object_push, cone friction.
     * Reward, termination and the observations ignore the ball (:426-497).  Lane mapping only.  [PARITY_ASSUMPTIONS A39] */
    int32_t balance_object;                 /* TG_BALANCE_* */
    double ball_radius, ball_mass, ball_mu, plate_radius;
    /* edge_follow, TCP_velocity_control, f64, 128-multiple images: the env step as ONE launch (csrc/tg_fused.hip) - the wavefront that steps an
     * env (base_tactile_env.py:166-185: apply_action .. get_observation is one chain per env) also resets it when its episode ended and draws
     * its tactile image(s) - instead of the three dependent launches k_step -> k_reset -> render.  Same step / reset code, the block raster's
     * arithmetic on another lane mapping: images, rewards, dones byte-identical, joint angles to the last bits (tests/test_gpu_fused_step.py).
     * Measured SLOWER on an MI355X (56 against 43 us per step at 1024 envs), hence TG_FUSED_AUTO = off; TG_FUSED_ON opts in;
     * the environment variable TG_FUSED_STEP (0 / 1) overrides.  tg_get_step_mode reports what runs. */
    int32_t fused_step;                     /* TG_FUSED_* */
    /* btContactSolverInfo::m_leastSquaresResidualThreshold of stepSimulation() (robot.py:141).  The reference's setPhysicsEngineParameter
     * call (base_tactile_env.py:127-130) passes fixedTimeStep, numSolverIterations, enableConeFriction and contactBreakingThreshold - NOT
     * solverResidualThreshold, so whatever PyBullet's physics server installs applies (PARITY_ASSUMPTIONS A7b: believed to be 1e-7; Bullet's
     * own library default is 0).  0 (default): the Gauss-Seidel loop leaves only at last-bit convergence, every mode of this library that
     * rests on that (the licensed analytic fixed point, the composed sweeps, object_balance's reset template) is available.  > 0 ("threshold
     * mode"): Bullet's rule - the loop leaves after the sweep whose largest squared row velocity change deltaImpulse / jacDiagABInv (a cone
     * friction pair counts once, with the sum of its two; A7c) is <= this, never before the first sweep, at most solver_iterations - evaluated
     * per env after every sweep on every mapping; every tick is a full tick, no licence, no composed sweeps, no reset template
     * (tg_state_view.solver_sweeps reports the sweeps run).  pgs_full_sweeps is ignored in this mode.  (ABI v13) */
    double solver_residual_threshold;
    /* object_balance, object_mode "spinning_plate" (balance_object = TG_BALANCE_SPINNING_PLATE; object_balance_env.py:107-108, 198-239, 267-269,
     * 355-358; ABI v14): the body on the TCP's constraint is the spool (plate_buffer.urdf: obj_mass / obj_com / obj_inertia describe IT, the
     * stimulus mesh is its mesh, its pivot sits at (0, 0, -spin_buffer_height / 2 + embed_dist) and is never updated) and the env's object - what
     * termination, reward and the oracle observation look at - is the dish (spinning_plate.urdf: spin_dish_*) standing on the spool's spindle.
     * Both collide as the convex hull of their mesh (margin spin_hull_margin each): the pair goes through the wave-mapped GJK / EPA and a
     * persistent manifold of up to four points (csrc/tg_spin.hip, PARITY_ASSUMPTIONS A35-A38, A41), friction spin_mu under the cone,
     * contact_breaking / contact_erp as for the other contact envs; the spool keeps obj_lin_damp / obj_ang_damp, the dish has none (:338-345).
     * Reset: dish to init_obj_pos = workframe + (0, 0, spin_buffer_height + obj_base_height / 2 - embed) - without the buffer height when
     * rand_embed is on, as upstream's reset_task (:317-321) -, spool to workframe + (0, 0, spin_buffer_height / 2), a one-tick torque (0, 0, -1)
     * in the dish's frame and ext_force at a random point of the dish (:355-358).  f64, UR5 chain, one wavefront per env; not with
     * solver_residual_threshold > 0.  Hull arrays are copied at tg_create. */
    double spin_dish_mass, spin_dish_com[3], spin_dish_inertia[9];
    double spin_buffer_height, spin_hull_margin, spin_mu;
    int32_t spin_n_dish, spin_n_spool;      /* hull vertex counts: n_dish <= 1152, n_spool <= 256 */
    const double* spin_dish_hull;           /* [spin_n_dish][3] in the dish's base frame */
    const double* spin_spool_hull;          /* [spin_n_spool][3] in the spool's base frame */
} tg_config;

enum { TG_BANK_AUTO = 0, TG_BANK_OFF = 1, TG_BANK_SYNC = 2, TG_BANK_ON = 3 };
enum { TG_NARROW_CLOSED_FORM = 0, TG_NARROW_GJK_MANIFOLD = 1, TG_NARROW_GJK_SINGLE = 2 };
enum { TG_BALANCE_POLE = 0, TG_BALANCE_BALL_ON_PLATE = 1, TG_BALANCE_SPINNING_PLATE = 2 };
enum { TG_FUSED_AUTO = 0, TG_FUSED_OFF = 1, TG_FUSED_ON = 2 };

typedef struct tg_ctx tg_ctx;

const char* tg_last_error(void);
int tg_abi_version(void);

/* gym.make(...) / env ctor: uploads constants, allocates per-env SoA state for num_envs envs, seeds env i with i. */
int tg_create(const tg_config* cfg, const tg_robot* robot, const tg_sensor* sensor, const tg_mesh* stimulus, tg_ctx** out);
int tg_destroy(tg_ctx* ctx);                                  /* env.close(), base_tactile_env.py:69-74 */

/* Use an existing HIP stream (e.g. torch's current stream) for all work; NULL = the context's own stream. */
int tg_set_stream(tg_ctx* ctx, void* hip_stream);

/* env.seed(s): per-env 64-bit seeds (SB3 convention seed+i), base_tactile_env.py:61-64. */
int tg_seed(tg_ctx* ctx, const uint64_t* seeds, int32_t n);

/* env.reset() for the envs whose mask byte is non-zero (NULL = all): task randomisation, rest pose, IK,
 * blocking move, first observation.  edge_follow_env.py:311-336, robot.py:114-125,188-260. Asynchronous. */
int tg_reset(tg_ctx* ctx, const uint8_t* host_mask);

/* env.step(a): actions float32[num_envs][act_dim] (device pointer if on_device, else host). Asynchronous.
 * base_tactile_env.py:166-185 -> robot.py:156-186 -> 24 x robot.py:131-141 -> tactile_sensor.py:261-294. */
int tg_step(tg_ctx* ctx, const float* actions, int32_t on_device);

int tg_sync(tg_ctx* ctx);                                     /* VecEnv.step_wait(): wait for enqueued work */

/* Device-resident results of the last step/reset (valid after tg_sync or on the context's stream).
 * The tactile observation buffer is READ-ONLY for the caller: for the edge / cube stimuli a render launch rewrites only the 16 x 16 pixel blocks
 * whose content changes (those it draws, and those an earlier launch drew that now show the untouched sensor again), so bytes a caller wrote into
 * it would survive into later observations.  TG_RASTER_REWRITE_ALL=1 in the environment makes every launch rewrite every block. */
int tg_get_obs_tactile(tg_ctx* ctx, void** dev_ptr);           /* uint8 [num_envs][H][W][1] */
int tg_get_terminal_obs(tg_ctx* ctx, void** dev_ptr);          /* uint8 [num_envs][H][W][1], rows valid where done */
int tg_get_reward_done_dev(tg_ctx* ctx, void** reward_f32, void** done_u8);
/* The three per-step outputs live in ONE device allocation: [tactile obs u8[N*H*W] | pad to 16 B | reward f32[N] | done u8[N]]
 * (tg_get_obs_tactile / tg_get_reward_done_dev point into it).  A rank ships this byte range to rank 0 as one message per step
 * (replaces SubprocVecEnv's per-env pickled pipes, sb3_helpers/rl_utils.py:17-30).  obs_bytes = offset of the reward. */
int tg_get_packed_outputs(tg_ctx* ctx, void** dev_ptr, int64_t* obs_bytes, int64_t* total_bytes);
/* Interior-only tactile payload for that message: the border ring of an image is a constant paste of the reference image
 * (tactile_sensor.py:291-292; 40 % of a 128 x 128 TacTip image), so a rank may ship only the 4-pixel words that hold a pixel inside the border
 * mask; *k = bytes per image (a multiple of 16; the pad repeats the last word).
 * tg_pack_interior: this context's current observations -> dst uint8 [num_envs][k] (device).  tg_unpack_interior (on the receiver; the
 * context only supplies the sensor constants): src uint8 [n_images][k] -> dst uint8 [n_images][H*W] with the ring restored.  Both are
 * enqueued on the context's stream.  *k = -1 with turn_off_border (the ring then carries rendered values). */
int tg_get_interior_count(tg_ctx* ctx, int32_t* k);
/* Reset bank (tg_config.reset_bank): how many auto-resets so far took a precomputed entry (*swapped) and how many were done on the spot because
 * the entry was not ready (*late); *mode = 0 bank off, 1 on, 2 on and waited for.  object_push (round 6): *mode = 3 - resets that took the reset
 * TEMPLATE (the arm's post-reset state is a function of constants while the tip stays clear of the previous episode's cube) / resets that ran their
 * blocking move; 0 with the template off.  Synchronises the context's stream. */
int tg_get_bank_stats(tg_ctx* ctx, int64_t* swapped, int64_t* late, int32_t* mode);
/* Render targets (multi-GPU; replaces the copy of rank 0's own observations into the gathered batch that SubprocVecEnv's parent does
 * per env, sb3_helpers/rl_utils.py:17-30): tg_set_obs_targets names up to two caller-owned device buffers uint8 [num_envs][H][W] - rank 0's
 * blocks of its two alternating gathered batches - and tg_select_obs_target picks where the NEXT steps / resets draw the tactile observations:
 * 0 = the context's own buffer (tg_get_packed_outputs), 1 / 2 = the caller's.  Each target has its own changed-block record (the block raster
 * rewrites only what changes, so a buffer must see every launch that is meant for it... or none) and its own captured step graphs.  Reward /
 * done / feature stay in the context's packed block; tg_get_obs_tactile returns the selected target.  count = 0 forgets the targets. */
int tg_set_obs_targets(tg_ctx* ctx, int32_t count, void* const* dev_ptrs);
int tg_select_obs_target(tg_ctx* ctx, int32_t index);
/* How tg_step runs on this context: *mode = 1 one launch per step (tg_config.fused_step; csrc/tg_fused.hip), 0 separate step / reset / render
 * launches; *envs_per_wavefront = envs one wavefront steps and draws in the one-launch form (0 otherwise). */
int tg_get_step_mode(tg_ctx* ctx, int32_t* mode, int32_t* envs_per_wavefront);
int tg_pack_interior(tg_ctx* ctx, void* dst_dev);
int tg_unpack_interior(tg_ctx* ctx, const void* src_dev, int32_t n_images, void* dst_dev);
/* ---- tile-sparse tactile payload and direct stores into rank 0's memory (csrc/tg_exchange.hip) ------------------------------------
 * A tactile image is zero away from the contact patch and a constant paste on the border ring (tactile_sensor.py:261-294).  Cut into
 * 16 x 16 tiles, only the tiles that differ from that constant template travel (lossless; 9 % of the bytes of an edge_follow batch, 22 % of
 * object_balance at 256 x 256, 48 % surface_follow, 73 % object_push).  Message: one 16-byte header {u32 count, n_images, tiles per image,
 * magic} and `count` records of 272 bytes {u32 tile id = image * tiles_per_image + tile, 12 bytes pad, 16 rows x 16 pixels}; records are in
 * no particular order.  These entry points are context free (raw device pointers + a HIP stream; NULL = the default stream).
 * tg_get_tile_template: the template of a context's sensor, uint8 [H][W] (device).
 * tg_pack_tiles: obs uint8 [n_images][h][w] -> dst (tg_tiles_capacity bytes; may be another GPU's memory opened with tg_ipc_open);
 *   counters_dev = two zeroed uint32 in LOCAL device memory (left zero again by every launch); tail_src_dev (may be NULL): tail_bytes (<= 1 MiB)
 *   copied to dst_dev + tail_offset by the same launch - the small block that rides behind the images (reward | done | feature).
 * tg_unpack_tiles: src message -> dst uint8 [n_images][h][w]: the template everywhere, then the records. */
int tg_get_tile_template(tg_ctx* ctx, void** dev_ptr);
int tg_tiles_capacity(int32_t n_images, int32_t h, int32_t w, int64_t* bytes);
int tg_pack_tiles(void* hip_stream, const void* obs_dev, const void* template_dev, int32_t n_images, int32_t h, int32_t w, void* dst_dev,
                  void* counters_dev, const void* tail_src_dev, int64_t tail_bytes, int64_t tail_offset);
int tg_unpack_tiles(void* hip_stream, const void* src_dev, const void* template_dev, int32_t n_images, int32_t h, int32_t w, void* dst_dev);
/* The same for the messages of n_ranks ranks in two launches: message r at src_dev + r * src_stride (bytes, a multiple of 16) ->
 * dst uint8 [n_ranks][n_images][h][w]; rank skip_rank (-1: none) is left alone (rank 0 copies its own images instead of packing them).
 * prev_ids_dev (may be NULL = fill everything): uint32 [n_ranks][1 + n_images * tiles] zeroed by the caller, owned by this destination
 * buffer: the list of tiles its last message had live; with it only those tiles get the template back before the new records land (dst
 * must hold the template to begin with). */
int tg_unpack_tiles_multi(void* hip_stream, const void* src_dev, int64_t src_stride, int32_t n_ranks, int32_t skip_rank, const void* template_dev,
                          int32_t n_images, int32_t h, int32_t w, void* dst_dev, void* prev_ids_dev);
/* Receive slots that peers store into directly (one process per GPU; the reference's counterpart is the pipe of each SubprocVecEnv worker,
 * sb3_helpers/rl_utils.py:17-30).  tg_ipc_alloc: zeroed device memory on the current device + its 64-byte IPC handle; tg_ipc_open /
 * tg_ipc_close: map / unmap it in another process (its GPU then reaches the memory over xGMI).  HSA_ENABLE_IPC_MODE_LEGACY=0 is required. */
int tg_ipc_alloc(int64_t bytes, void** dev_ptr, uint8_t* handle64);
int tg_ipc_alloc_was_uncached(void);   /* 1: the last tg_ipc_alloc of this process got uncached device memory; 0: the runtime refused and it is plain hipMalloc memory */
int tg_ipc_free(void* dev_ptr);
int tg_ipc_open(const uint8_t* handle64, void** dev_ptr);
int tg_ipc_close(void* dev_ptr);
/* Device copy on a stream whose destination (or source) may be memory opened with tg_ipc_open; both pointers 16-byte aligned. */
int tg_copy_bytes(void* hip_stream, void* dst_dev, const void* src_dev, int64_t bytes);
int tg_copy_bytes2(void* hip_stream, void* dst1_dev, const void* src1_dev, int64_t bytes1, void* dst2_dev, const void* src2_dev, int64_t bytes2);   /* two ranges, one launch */
/* tg_copy_bytes2 whose launch also raises n_flags flags (flags_dev[i * stride_words] = value, system-scope release; NULL: none). */
int tg_copy_bytes2_flag(void* stream, void* dst1_dev, const void* src1_dev, int64_t bytes1, void* dst2_dev, const void* src2_dev, int64_t bytes2,
                        void* flags_dev, int32_t n_flags, int32_t stride_words, uint32_t value);
/* Stream-ordered flags (uint32, monotone step counters) in such memory.  tg_flag_set: after everything enqueued before it on the stream has
 * finished, flags[i * stride_words] = value for i < n (release, system scope).  tg_flag_wait: the stream goes on once every
 * flags[i * stride_words] has reached value (compared modulo 2^32); after timeout_ms of waiting it goes on anyway and ORs bit (i & 31) into
 * *err_dev (uint32 in local device memory, may be NULL); a wait that finds *err_dev already non-zero goes on at once (one timeout has shown the
 * exchange to be broken: the later waits must not sit theirs out as well).  n <= 64. */
int tg_flag_set(void* hip_stream, void* flags_dev, int32_t n, int32_t stride_words, uint32_t value);
int tg_flag_wait(void* hip_stream, const void* flags_dev, int32_t n, int32_t stride_words, uint32_t value, void* err_dev, int32_t timeout_ms);
/* Envs with an "extended_feature" observation (object_push, object_roll) append it to the same allocation, so that config 4's
 * tactile_and_feature observation still travels as one message: [... | done u8[N] | pad to 4 B | feature f32[N][dim]]
 * (object_push_env.py:611-629).  *feature_off = byte offset of the feature block (-1: this env has none). */
int tg_get_packed_feature(tg_ctx* ctx, int64_t* feature_off, int32_t* dim);
/* "extended_feature" observation (object_push_env.py:611-629): float32 [num_envs][*dim]: TCP pos, rpy and current goal
 * pos, rpy in the work frame; terminal != 0: the copy taken at the last step (rows valid where done). */
int tg_get_obs_feature(tg_ctx* ctx, void** dev_ptr, int32_t* dim, int32_t terminal);
/* observation_mode "oracle" (get_oracle_obs: edge_follow_env.py:454-476, base_surface_env.py:789-819, object_balance_env.py:528-563,
 * object_push_env.py:571-609, object_roll_env.py:367-407): float32 [num_envs][*dim] computed on the device from the current state
 * (dim 10 / 20 / 26 / 30 / 34 by env kind).  Enqueued on the context's stream; the pointer stays valid for the context's lifetime. */
int tg_get_obs_oracle(tg_ctx* ctx, void** dev_ptr, int32_t* dim);
int tg_copy_obs_oracle(tg_ctx* ctx, float* host_dst);          /* synchronises */
/* observation_mode "oracle" as a per-step output: tg_step / tg_reset then also write the vectors (tg_get_obs_oracle returns them without a
 * launch of its own), and the step's own vectors - taken before the auto-reset - are kept: rows of the envs that finished are their
 * terminal observation (VecEnv info["terminal_observation"]).  Call before the first tg_step. */
int tg_enable_oracle_obs(tg_ctx* ctx);
int tg_get_obs_oracle_terminal(tg_ctx* ctx, void** dev_ptr);   /* float32 [num_envs][dim] */
int tg_copy_obs_oracle_terminal(tg_ctx* ctx, float* host_dst); /* synchronises */
/* Per-env episode statistics: what the reference's callers read from the Monitor wrapper they put around every env
 * (sb3_helpers/rl_utils.py:17-30, 59: info["episode"] = {"r", "l", "t"}).  The step kernels add up, in double, the float32 rewards they hand
 * out; when an env reports done its return and length (in env steps) are kept here until its next episode ends.  Rows valid where done. */
int tg_get_episode_stats(tg_ctx* ctx, void** final_return_f32, void** final_len_i32);      /* device: float32 [num_envs], int32 [num_envs] */
int tg_copy_episode_stats(tg_ctx* ctx, float* final_return, int32_t* final_len);           /* synchronises */
/* Host copies (synchronise). */
int tg_get_reward_done(tg_ctx* ctx, float* reward, uint8_t* done);
int tg_copy_obs_tactile(tg_ctx* ctx, uint8_t* host_dst, int32_t terminal);
/* The images of `count` chosen envs only: env_ids[k]'s tactile image (visual != 0: its scene-camera image, after tg_set_scene) goes to
 * host_dst + k * image_bytes, from the current observation buffer or (terminal != 0) from the terminal one.  What a VecEnv's step_wait reads
 * for info["terminal_observation"] (stable_baselines3's VecEnv contract; sb3_helpers/rl_utils.py:17-37 wraps the envs in one): with the
 * episodes out of phase a handful of envs finish in nearly every step, and copying the whole terminal batch for them (16.8 MB at 1024 x
 * 128 x 128) was two thirds of the numpy step's time.  ABI v12. */
int tg_copy_obs_rows(tg_ctx* ctx, int32_t visual, int32_t terminal, const int32_t* env_ids, int32_t count, uint8_t* host_dst);
/* The same for the envs that finished in the last step, without the host naming them (ABI v13, round 6): ONE launch on the context's stream compacts
 * the done flags and stores, straight into device-visible pinned host memory `dst_pinned` (tg_done_rows_bytes(cap) bytes, 16-byte aligned), a 16-byte
 * header {u32 count = envs done, u32 cap, u32 num_envs, u32 magic 0x74674452}, then int32 ids[cap] (ascending), float32 episode returns[cap], int32
 * episode lengths[cap] (the Monitor statistics of tg_get_episode_stats), then, 16-byte aligned, uint8 rows[cap][H * W]: the TERMINAL tactile images of
 * the first min(count, cap) finished envs.  What info["terminal_observation"] / info["episode"] need arrives under the synchronisation the
 * observation fetch makes anyway (sb3_helpers/rl_utils.py:17-37: SubprocVecEnv's workers send obs, reward, done and info in one message).
 * count > cap: the caller falls back to tg_copy_obs_rows / tg_copy_episode_stats for that step. */
int tg_done_rows_bytes(tg_ctx* ctx, int32_t cap, int64_t* bytes);
int tg_pack_done_rows(tg_ctx* ctx, void* dst_pinned, int32_t cap);
int tg_copy_obs_feature(tg_ctx* ctx, float* host_dst, int32_t terminal);   /* float32 [num_envs][12] */

/* Parity / inspection view of the per-env state, host arrays sized by the caller ([num_envs][...]), any may be NULL. */
typedef struct {
    double*  q;              /* [num_envs][ndof] */
    double*  qd;             /* [num_envs][ndof] */
    double*  qd_target;      /* [num_envs][ndof] joint velocity targets of the last controller call */
    double*  tcp_pos;        /* [num_envs][3] world, inertial-frame convention */
    double*  tcp_rpy;        /* [num_envs][3] */
    double*  edge_ang;       /* [num_envs] */
    double*  embed_dist;     /* [num_envs] */
    float*   stim_xform;     /* [num_envs][12] camera<-stimulus transform used by the last render */
    int32_t* step_count;     /* [num_envs] */
    int32_t* reset_ticks;    /* [num_envs] sim ticks used by the last reset's blocking move */
    uint64_t* rng_state;     /* [num_envs] */
    double*  goal_pos;       /* [num_envs][3] world (surface_follow) */
    double*  direction;      /* [num_envs][2] work-frame auto-drive direction (surface_follow) */
    double*  heights;        /* [num_envs][rows*cols] heightfield_data[row][col] (surface_follow) */
    float*   surf_zoff;      /* [num_envs] vertical centring offset applied to the rendered heightfield */
    double*  body_pos;       /* [num_envs][3] free object base frame (object_balance) */
    double*  body_rot;       /* [num_envs][9] row major */
    double*  body_linvel;    /* [num_envs][3] velocity of the composite centre of mass */
    double*  body_angvel;    /* [num_envs][3] */
    double*  gravity_z;      /* [num_envs] */
    double*  traj;           /* [num_envs][3][TG_MAX_TRAJ_POINTS] work-frame x, y, yaw of the goal trajectory (object_push) */
    int32_t* goal_id;        /* [num_envs] targ_traj_list_id (object_push) */
    double*  obj_mass;       /* [num_envs] (object_push) */
    /* Contact pairs of the last stepSimulation tick (robot.py:141), in solver row order (object_push / object_roll; 0 elsewhere:
     * the other envs filter every sensor contact out, tactile_sensor.py:46-57, base_surface_env.py:432).  An id names the pair by its
     * feature: 0-7 = cube vertex (index 4 ix + 2 iy + iz of the +-half extents) against the table, the marble's table contact is 0;
     * 8 + k = hull vertex k of the sensor tip's collision core against the cube (8 for the marble against the tip's cylinder).
     * Unused slots are -1.  Integer data: compared bit-exactly with the oracle. */
    int32_t* contact_count;  /* [num_envs] */
    int32_t* contact_ids;    /* [num_envs][8]: solver row order, -1 beyond contact_count (4 table + up to 4 tip slots) */
    double*  ball_pos;       /* [num_envs][3] object_balance ball_on_plate: the ball's centre, world */
    double*  ball_linvel;    /* [num_envs][3] */
    double*  ball_angvel;    /* [num_envs][3] */
    double*  ball_impulse;   /* [num_envs] normal impulse of the ball - plate contact in the last sim tick (0: not touching) */
    double*  dish_state;     /* [num_envs][20] spinning_plate: the dish's base position (0-2), orientation (3-11, row major), linear (12-14) and angular
                                (15-17) velocity, the last tick's summed normal impulse (18) and number of manifold points (19); body_* is the spool */
    int32_t* broadphase_pairs;  /* [num_envs] last broadphase check (tg_set_broadphase): unexpected pairs whose world AABBs overlap (stage 1) */
    int32_t* broadphase_hits;   /* [num_envs] ... of which the oriented boxes / the hull also overlap (stages 2, 3): 0 = no unmodelled contact possible */
    int32_t* broadphase_mask;   /* [num_envs] bit k: slot k is part of a hit */
    int32_t* solver_sweeps;  /* [num_envs] threshold mode (tg_config.solver_residual_threshold > 0): PGS sweeps the sim ticks of the env's last
                              * tg_step ran, summed over the ticks (a reset's ticks are not counted); zeros otherwise.  (ABI v13) */
} tg_state_view;
int tg_get_state(tg_ctx* ctx, const tg_state_view* view);
/* Overwrite joint state (tests): q, qd [num_envs][ndof]; re-evaluates the cached TCP pose. */
int tg_set_joint_state(tg_ctx* ctx, const double* q, const double* qd);

/* Per-kernel timing (bench.py's roofline leg).  enable = 1: HIP event pairs around every launch class on the launch stream, the step's launches
 * issued one by one (no graph); every figure carries what an EMPTY event pair measures (3 - 5 us).  enable = 2: only the kernels' own clock
 * (csrc/tg_kt.hpp: every wavefront stamps its start and end, wall_clock64; first start -> last end per class), the step's launches stay as the
 * rollout issues them (since round 6: on the stream; one replayed hipGraph with TG_STEP_GRAPH=1) - the figures of the rollout itself.  enable = 0: off.
 * tg_profile_get which: 0 step kernel, 1 render of all envs (the one launch of a fused step), 2 reset sequence, 3 masked render (reset /
 * auto-reset envs only), 4 scene camera, 5 an empty event pair - by HIP events; 8 + k (k = 0 .. 3): class k by the kernels' own clock. */
int tg_profile_enable(tg_ctx* ctx, int32_t enable);
int tg_profile_get(tg_ctx* ctx, int32_t which, double* total_ms, int64_t* launches);

/* ---- scene camera: the "visual" / "visuotactile" observations and render() (base_tactile_env.py:212-245, 284-320) -------------------
 * get_visual_obs draws the whole scene (plane, table, robot, stimulus) from a fixed world camera with getCameraImage and keeps rgb.
 * The triangle set is shared by all envs; each triangle is rigid in one frame: 0 the world, 1 + i moving link i (tg_robot numbering),
 * ndof + 1 the task's stimulus / free body (same frame as tg_mesh).  Upstream's pixels depend on the GL driver: PARITY_ASSUMPTIONS A31-A33. */
typedef struct {
    int32_t image_h, image_w;               /* rgb_image_size (= image_size upstream); <= 128 or a multiple of 128 */
    int32_t n_verts, n_tris;
    const float*   verts;                   /* host [n_verts][3] */
    const int32_t* tris;                    /* host [n_tris][3] */
    const uint8_t* tri_frame;               /* host [n_tris] */
    const uint8_t* tri_rgb;                 /* host [n_tris][3], the <material> colour */
    double cam_target[3], cam_dist, cam_yaw_deg, cam_pitch_deg;   /* rgb_cam_pos / _dist / _yaw / _pitch, e.g. edge_follow_env.py:176-195 */
    double fov_deg, near_plane, far_plane;  /* rgb_fov / rgb_near_val / rgb_far_val */
    double light_dir[3];                    /* world, towards the light */
    uint8_t background[3];
    uint8_t body_rgb[3];                    /* surface envs: colour of the per-env heightfield (changeVisualShape rgbaColor, base_surface_env.py:431) */
    int32_t body_heightfield;               /* != 0: the task's body is the env's heightfield (frame ndof + 1), not part of the triangle set */
    int32_t every_step;                     /* != 0: tg_step / tg_reset also draw the scene (observation modes with "visual" / "visuo") */
} tg_scene;
/* Uploads the scene; call before the first tg_step. */
int tg_set_scene(tg_ctx* ctx, const tg_scene* scene);

/* ---- broadphase guard (ABI v13; csrc/tg_broadphase.hip, oracle/broadphase.py) --------------------------------------------------------------------
 * PyBullet's stepSimulation (robots/arms/robot.py:141) runs Bullet's broadphase over the world AABBs of every collision object and hands every
 * overlapping pair of different bodies (not both static, neither filtered out) to the narrowphase.  This library's contact sets are FIXED per env
 * from the reference's collision filters (sensors/tactile_sensor.py:46-57, robots/arms/mg400/mg400.py:68-72, base_surface_env.py:432,
 * object_push_env.py:249): the guard checks per env step that no OTHER pair can touch - one oriented box per URDF link with <collision> geometry
 * (the robot's, the table, the plane, the stimulus, the free objects), world AABBs sorted and swept on x in the env's wavefront (stage 1: what
 * Bullet's broadphase would report), the pairs found narrowed by an oriented-box separating-axis test (stage 2) and, for a robot link against the
 * table, by the link's convex hull against the table top (stage 3).  Results per env in tg_state_view.broadphase_pairs / _hits / _mask; a
 * non-zero hit count means PyBullet may generate a contact this library has no solver row for.
 * Slots: 0-15 the robot's boxes in URDF link order, 16 table, 17 plane, 18 edge stimulus, 19 / 20 the object's boxes, 21 the ball of
 * ball_on_plate.  src: where a slot's pose comes from - TG_BP_LINK the robot's moving link `link` (-1: the fixed base), TG_BP_WORLD none (center is
 * in the world), TG_BP_EDGE the episode's edge (stim_pos, yaw = edge angle), TG_BP_BODY the free body's pose, TG_BP_SPHERE its position only with the
 * box scaled to the episode's radius (object_roll), TG_BP_BALL the ball of ball_on_plate. */
enum { TG_BP_NONE = 0, TG_BP_LINK = 1, TG_BP_WORLD = 2, TG_BP_EDGE = 3, TG_BP_BODY = 4, TG_BP_SPHERE = 5, TG_BP_BALL = 6 };
#define TG_BP_SLOTS 22
typedef struct {
    double center[3], rot[9], half[3];      /* the box in the frame `src` names */
    int32_t src;                            /* TG_BP_* (TG_BP_NONE: empty slot, or a link the reference filters out) */
    int32_t link;                           /* TG_BP_LINK: moving link index, -1 = base */
    int32_t body;                           /* boxes of one body are never paired */
    int32_t is_static;                      /* two static boxes are never paired (robot base, table, plane, edge) */
    int32_t hull_off, hull_n;               /* TG_BP_LINK: the link's convex-hull vertices (moving-link frame) in `hull_verts`, for stage 3 */
    uint32_t expected;                      /* bit k: the pair (this slot, slot k) is one the solver has rows for */
    int32_t conj;                           /* -1, or the slot of a second box bounding the SAME shape (a disc: its square and the square turned 45
                                             * degrees): a pair with this slot is a hit only if the other box overlaps the second one as well */
} tg_bp_box;
typedef struct {
    tg_bp_box box[TG_BP_SLOTS];
    double margin;                          /* added to every half extent: what a box travels inside one env step + contactBreakingThreshold */
    double hull_margin;                     /* Bullet inflates a URDF convex hull by gUrdfDefaultCollisionMargin = 0.001 */
    double sphere_half;                     /* half extent of the TG_BP_SPHERE / TG_BP_BALL asset boxes (= the asset's radius) */
    double ball_radius;                     /* TG_BP_BALL: the ball's radius in the world */
    int32_t n_hull_verts;
    int32_t every_step;                     /* 1: the check is a node of every tg_step / tg_step_random (after the step kernel, before any reset) */
    const double* hull_verts;               /* host, [n_hull_verts][3] */
} tg_broadphase;
int tg_set_broadphase(tg_ctx* ctx, const tg_broadphase* guard);        /* NULL: forget the guard */
int tg_check_broadphase(tg_ctx* ctx);                                   /* enqueue one check of the current state on the context's stream */
int tg_get_broadphase_totals(tg_ctx* ctx, int64_t* checks, int64_t* pairs, int64_t* hits);   /* env-checks run, stage-1 pairs and hits found since tg_set_broadphase (synchronises) */
/* Draws the current state of every env now (render() in the other observation modes). Asynchronous. */
int tg_render_scene(tg_ctx* ctx);
/* uint8 [num_envs][H][W][3]; terminal != 0: the image of the last step of the envs that finished (rows valid where done). */
int tg_get_obs_visual(tg_ctx* ctx, void** dev_ptr, int32_t terminal);
int tg_copy_obs_visual(tg_ctx* ctx, uint8_t* host_dst, int32_t terminal);

/* ---- function-level entry points (parity tests call the device implementations through these) --------------------- */
/* calculateInverseDynamics (base_robot_arm.py:176-178), batch of n states; physics_dtype as in tg_config. */
int tg_inverse_dynamics(const tg_robot* robot, int32_t physics_dtype, int32_t n, const double* q, const double* qd,
                        const double* qdd, double* tau);
/* joint-space inertia matrix, [n][ndof][ndof]. */
int tg_mass_matrix(const tg_robot* robot, int32_t physics_dtype, int32_t n, const double* q, double* M);
/* calculateJacobian at the TCP frame (base_robot_arm.py:300-307): J [n][6][ndof]; also TCP pose [n][3], [n][9]. */
int tg_jacobian_tcp(const tg_robot* robot, int32_t physics_dtype, int32_t n, const double* q, double* J, double* pos,
                    double* rot);
/* n_ticks x stepSimulation (robot.py:131-141) with gravity compensation and velocity (mode 1) or position (mode 2)
 * motors; q, qd updated in place ([n][ndof]). */
int tg_sim_ticks(const tg_robot* robot, int32_t physics_dtype, int32_t n, int32_t n_ticks, int32_t solver_iterations,
                 double dt, int32_t motor_mode, const double* q_des, const double* qd_des, double max_force, double* q,
                 double* qd);
/* calculateInverseKinematics at the TCP frame (base_robot_arm.py:201-209, maxNumIterations=100, residualThreshold=1e-8):
 * q0 [n][ndof] start, target_pos [n][3], target_rot [n][9] row-major -> q_out [n][ndof], iters [n] (nullable). */
int tg_inverse_kinematics(const tg_robot* robot, int32_t physics_dtype, int32_t n, const double* q0, const double* target_pos,
                          const double* target_rot, int32_t max_iters, double threshold, double* q_out, int32_t* iters);
/* getCameraImage depth + t_s_camera (tactile_sensor.py:239-294) for n transforms [n][12] -> uint8 [n][h][w]. */
int tg_render_tactile(const tg_sensor* sensor, const tg_mesh* mesh, int32_t n, const float* cam_from_obj, uint8_t* out);
/* Same for a per-image heightfield stimulus (createCollisionShape(GEOM_HEIGHTFIELD), base_surface_env.py:402-432):
 * heights [n][rows*cols] (double, heightfield_data[row][col]), zoff [n]. */
int tg_render_tactile_heightfield(const tg_sensor* sensor, int32_t rows, int32_t cols, double grid_scale, int32_t n,
                                  const double* heights, const float* zoff, const float* cam_from_obj, uint8_t* out);
/* gen_heigtfield_simplex_2d (base_surface_env.py:319-337) for n seeds: heights [n][rows*cols], zoff [n] (nullable). */
int tg_gen_heightfield(int32_t n, const int64_t* seeds, int32_t rows, int32_t cols, double interp, double range, double* heights,
                       float* zoff);

/* action_space.sample() for the whole batch (the reference's demo loops draw env.action_space.sample() per env and step,
 * examples/demo_rl_env_base.py:34; Box(min_action, max_action) float32, edge_follow_env.py:169-174): dev_actions [num_envs][act_dim]
 * (device memory) receives U[min_action, max_action) draws, counter based: element i of draw `counter` is a function of
 * (seed, counter, i) only.  Enqueued on the context's stream, so a tg_step(dev_actions, on_device = 1) that follows sees it. */
int tg_sample_actions(tg_ctx* ctx, uint64_t seed, uint64_t counter, float* dev_actions);
/* A random-action rollout step (the north_star's synthetic rollout: `env.step(env.action_space.sample())`, examples/demo_rl_env_base.py:34, for the
 * whole batch): tg_sample_actions' draw followed by tg_step on it as one sequence of launches - the draw is made by the step kernel itself where the
 * lane-mapped k_step / k_step_body_wave runs, by the sequence's first launch elsewhere; the draw counter lives in device memory and moves on by one
 * per call.  (Until round 6 the sequence was replayed as one captured hipGraph; TG_STEP_GRAPH=1 still does: a graph launch costs ~6.6 us before its
 * first kernel starts on this stack, a launch on the stream ~2 us.)  restart != 0 (or a new seed): the next step uses draw first_draw + 1.  The
 * actions are the context's own buffer (tg_get_actions: device float32 [num_envs][act_dim]); draw k equals tg_sample_actions(seed, k). */
int tg_step_random(tg_ctx* ctx, uint64_t seed, uint64_t first_draw, int32_t restart);
int tg_get_actions(tg_ctx* ctx, void** dev_actions);
#ifdef __cplusplus
}
#endif
#endif
