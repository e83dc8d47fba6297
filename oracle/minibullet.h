/*
 * oracle/minibullet.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU (double precision, scalar, deliberately naive) restatement of the handful of PyBullet calls that the
 * reference's per-step hot path makes.  The reference is pure Python; all arithmetic on this path lives in the
 * un-vendored third-party `pybullet` wheel (requirements.txt:6, `pybullet>=3.1.0`, no pinned version), which is
 * not installable here.  Each function below therefore restates the *published* Bullet algorithm for the call
 * site it replaces and cites that call site; every Bullet-internal behaviour that is assumed rather than
 * verified is listed in PARITY_ASSUMPTIONS.md.
 *
 * Pinning status:
 *   - mb_render_depth / camera chain: PINNED against the reference's committed fixtures
 *     (assets/robot_assets/<sensor>/reference_images/<type>/<N>x<N>/nodef_dep.npy) — tests/test_oracle_golden.py.
 *   - mb_fk: pinned against the reference's rest poses / work frames (edge_follow/rest_poses.py:6-20,
 *     edge_follow_env.py:95,106-107,305).
 *   - mb_step / mb_inverse_dynamics / mb_ik (post-step poses): PARITY UNPINNED — the reference holds no golden
 *     vectors for them and PyBullet cannot be run here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef MINIBULLET_H
#define MINIBULLET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MB_MAX_DOF 8
#define MB_MAX_BODIES 24

typedef struct {
    int32_t ndof;
    int32_t nbodies;
    int32_t parent[MB_MAX_DOF];          /* parent moving link, -1 = fixed base */
    double joint_pos[MB_MAX_DOF][3];     /* joint origin in parent link frame */
    double joint_rot[MB_MAX_DOF][9];     /* joint frame orientation in parent link frame (row major) */
    double joint_axis[MB_MAX_DOF][3];    /* revolute axis in joint frame */
    int32_t body_link[MB_MAX_BODIES];
    double body_com[MB_MAX_BODIES][3];
    double body_rot[MB_MAX_BODIES][9];
    double body_mass[MB_MAX_BODIES];
    double body_inertia[MB_MAX_BODIES][3];
    double gravity[3];                   /* base_tactile_env.py:126 setGravity(0,0,-9.81) */
    double linear_damping;               /* base_robot_arm.py:24 (0.04) */
    double angular_damping;              /* base_robot_arm.py:24 (0.04) */
    double joint_damping;                /* base_robot_arm.py:25 (0.01) */
} mb_model;

enum { MB_MOTOR_OFF = 0, MB_MOTOR_VELOCITY = 1, MB_MOTOR_POSITION = 2 };

typedef struct {
    double q[MB_MAX_DOF];
    double qd[MB_MAX_DOF];
    double applied_torque[MB_MAX_DOF];   /* TORQUE_CONTROL feed-forward, cleared every tick like Bullet's forces */
    int32_t motor_mode[MB_MAX_DOF];
    double motor_q_des[MB_MAX_DOF];
    double motor_qd_des[MB_MAX_DOF];
    double motor_kp[MB_MAX_DOF];
    double motor_kd[MB_MAX_DOF];
    double motor_max_force[MB_MAX_DOF];
} mb_state;

/* getLinkState(..)[0:2] forward kinematics: world rotation (row-major 3x3) and origin of every moving link. */
void mb_fk(const mb_model* m, const double* q, double* R /*[ndof][9]*/, double* p /*[ndof][3]*/);

/* getLinkState for a frame rigidly attached to moving link `link` (PyBullet reports the inertial frame):
 * pos[3], rot[9], and (computeLinkVelocity=1) world linear velocity of the frame origin and angular velocity. */
void mb_frame_state(const mb_model* m, const double* q, const double* qd, int link, const double* fpos,
                    const double* frot, double* pos, double* rot, double* linvel, double* angvel);

/* calculateInverseDynamics(q, qd, qdd) — base_robot_arm.py:176-178. */
void mb_inverse_dynamics(const mb_model* m, const double* q, const double* qd, const double* qdd, double* tau);

/* joint-space inertia matrix M(q) [ndof x ndof] (used by mb_step; exposed for tests). */
void mb_mass_matrix(const mb_model* m, const double* q, double* M);

/* calculateJacobian(link, localPosition=[0,0,0]) — base_robot_arm.py:300-307.  J is [6][ndof] row-major,
 * rows 0-2 translational, 3-5 rotational, world frame. */
void mb_jacobian(const mb_model* m, const double* q, int link, const double* fpos, double* J);

/* stepSimulation() — robot.py:141 with base_tactile_env.py:127-130 parameters. */
void mb_step(const mb_model* m, mb_state* s, double dt, int solver_iterations);

/* calculateInverseKinematics(link, pos, orn, maxNumIterations, residualThreshold) — base_robot_arm.py:201-209.
 * target_rot row-major 3x3.  q is updated in place from its starting value.  Returns iterations used. */
int mb_ik(const mb_model* m, int link, const double* fpos, const double* frot, const double* target_pos,
          const double* target_rot, double* q, int max_iters, double residual_threshold);

/* ----------------------------------------------------------------------------------------------------------
 * getCameraImage() depth channel — tactile_sensor.py:239-246 — for a triangle soup given in object coordinates
 * and a camera<-object rigid transform (GL eye space: x right, y up, -z forward).  Single precision, following the
 * raster specification in DESIGN.md section "Raster specification" operation by operation.
 * depth[h*w] must be pre-filled (1.0f = far plane or an existing depth image); the routine z-tests into it. */
void mb_render_depth(const float* verts /*[nv][3]*/, int nv, const int32_t* tris /*[nt][3]*/, int nt,
                     const float* cam_from_obj /*[12]: R row-major 3x3 then t[3]*/, float fov_deg, float near_,
                     float far_, int w, int h, float* depth);

/* TactileSensor.t_s_camera() post-process — tactile_sensor.py:271-292 — on a current depth image. */
/* Scene camera rgb (get_visual_obs, base_tactile_env.py:212-245) - PARITY UNPINNED, see minibullet.c.  xf [n_frames][12]: eye <- frame
 * (R row-major, then t) as float; tri_xf picks a triangle's frame; zbuf [W*H] scratch; out [H][W][3]. */
void mb_render_scene(const float* verts, const int32_t* tris, const uint8_t* tri_xf, const uint8_t* tri_rgb, int nt, const float* xf,
                     const float* light_eye, float fov_deg, float near_, float far_, int W, int H, const uint8_t* background,
                     uint64_t* zbuf, uint8_t* out);
void mb_blend_spheres(const float* spheres, int n, const float* light_eye, float fov_deg, float near_, float far_, int W, int H, const uint64_t* zbuf,
                      uint8_t* out);
void mb_t_s_camera(const float* cur_dep, const float* nodef_dep, const float* nodef_gray, const uint8_t* border_mask,
                   int npix, int turn_off_border, uint8_t* out);

/* ----------------------------------------------------------------------------------------------------------
 * OpenSimplex noise (2-D), the third-party `opensimplex` package the reference imports for surface generation
 * (requirements.txt:4, unpinned; call sites base_surface_env.py:319-337 `noise2(x*.05, y*.05) * .025`, :448
 * `OpenSimplex(seed=np_random.randint(1e8))`).  Restated from the published algorithm (K. Spencer, 2014:
 * stretch/squish constants, 8 gradients, 64-bit LCG permutation shuffle) — PARITY UNPINNED: the package is not
 * installed here and the reference holds no golden surfaces.  Compiled without FMA contraction so the HIP
 * implementation (tactile_gym_amd/csrc/tg_noise.hip) is bit-identical. */
void mb_opensimplex_perm(int64_t seed, int16_t* perm /*[256]*/);
double mb_opensimplex_noise2(const int16_t* perm, double x, double y);
/* gen_heigtfield_simplex_2d (base_surface_env.py:319-337): out[x*cols + y] = noise2(x*interp, y*interp) * range */
void mb_heightfield_simplex2d(int64_t seed, int rows, int cols, double interp, double range, double* out);

/* ----------------------------------------------------------------------------------------------------------
 * object_balance: a free rigid body (the pole: base plate + pole links welded together) tied to the arm's TCP link by a
 * point-to-point constraint (object_balance_env.py:261-283 createConstraint(JOINT_POINT2POINT)).  stepSimulation then
 * solves the arm's joint motors and the three P2P rows in one projected Gauss-Seidel loop [PARITY_ASSUMPTIONS A18-A21].
 * PARITY UNPINNED (no golden data). */
typedef struct {
    double mass;
    double com[3];          /* composite centre of mass in the base (inertial) frame that get/resetBasePositionAndOrientation use */
    double inertia[9];      /* composite inertia about com, base-frame axes, row major */
    double pos[3];          /* base frame origin, world */
    double rot[9];          /* base frame orientation, world, row major */
    double linvel[3];       /* velocity of the composite centre of mass, world */
    double angvel[3];       /* world */
    double ext_force[3];    /* applyExternalForce(WORLD_FRAME): force and application point; used by the next tick, then cleared */
    double ext_pos[3];
    int32_t ext_pending;
} mb_body;

typedef struct {
    int32_t link;           /* arm link carrying pivot A */
    double pivot_a[3];      /* in that link's frame (URDF frame of the moving link) */
    double pivot_b[3];      /* in the body's base frame */
    double erp;             /* 0.2 */
    double max_impulse;     /* 500 */
} mb_p2p;

/* stepSimulation() for arm + body + P2P.  Row order: joint motors (joint order), then P2P x, y, z. */
void mb_step_body(const mb_model* m, mb_state* s, mb_body* b, const mb_p2p* c, double dt, int solver_iterations);

/* object_balance, object_mode "ball_on_plate" (object_balance_env.py:105-106, 187-199, 245-260): the body tied to the TCP is the round plate
 * (round_plate.urdf: cylinder r 0.1, length 0.0025, mass 0.01) and a free ball (sphere.urdf, globalScaling 7.5: r 0.01875, mass 0.05,
 * lateralFriction 10) rolls on it.  The ball touches only the plate here: one contact from the closest point of the plate's solid cylinder to
 * the ball's centre (the marble / tip-cylinder closed form of A30), normal + two friction rows with cone friction, rigid (erp, cfm 0); a ball
 * that has left the plate falls freely [PARITY_ASSUMPTIONS A39].  Neither reward, termination nor the tactile image look at the ball
 * (object_balance_env.py:444-526): it matters through its weight and rolling on the plate.  PARITY UNPINNED. */
typedef struct {
    double radius, mass, inertia;          /* solid sphere about its centre: 0.4 m r^2 (Bullet recomputes it from the collision shape, A30) */
    double pos[3], linvel[3], angvel[3];
    double ext_torque[3];                  /* applyExternalTorque(LINK_FRAME) on a ball that was just reset (identity orientation): next tick only */
    int32_t ext_pending;
    double mu;                             /* combined friction ball x plate */
    double plate_radius, plate_half_len;   /* the plate's collision cylinder: axis = z of the plate's base frame, centred on it */
    double breaking, erp;                  /* contactBreakingThreshold, contact ERP */
    int32_t in_contact;                    /* out: the last tick made a ball - plate contact */
    double depth, normal_impulse;          /* out */
} mb_ball;
/* Row order: joint motors, P2P x y z (these in reversed order on even sweeps), then the contact's normal, then its friction pair. */
void mb_step_body_ball(const mb_model* m, mb_state* s, mb_body* plate, const mb_p2p* c, mb_ball* ball, double dt, int solver_iterations);

/* object_balance, object_mode "spinning_plate" (object_balance_env.py:107-108, 198-239, 267-269, 355-358): the body tied to the TCP is the
 * spool (plate_buffer.urdf, mass 0.1) and the free object is the dish (spinning_plate.urdf, mass 0.6) standing on the spool's spindle.  Both
 * collide as the convex hull of their mesh (btConvexHullShape with the URDF margin 1e-3); the pair's contacts come from the general
 * narrowphase (oracle/narrowphase.c: GJK / EPA on the two hulls in the spool's frame behind an AABB overlap test, one new point per tick into
 * a persistent manifold of up to four, A35-A38) and are solved as in mb_step_push: normal rows, then cone-friction pairs, rigid (erp, cfm 0).
 * The spool keeps Bullet's default velocity damping 0.04 (reset_object clears the damping of obj_id = the dish only, :338-345); the dish gets
 * the one-tick torque and force of apply_random_torque_obj / apply_random_force_base (:357-358).  Neither body touches anything else: the
 * tip's collision is off (t_s_core "no_core", :54) [PARITY_ASSUMPTIONS A41].  PARITY UNPINNED. */
typedef struct mb_spin_s mb_spin;

/* ----------------------------------------------------------------------------------------------------------
 * object_push: a free box (the cube) resting on the table and pushed by the sensor tip's collision core
 * (object_push_env.py; tip core collision on: t_s_core = "fixed", :60).  Restatement of a Bullet-style pipeline with this
 * repo's own, explicitly simplified contact generation [PARITY_ASSUMPTIONS A23-A27] — PARITY UNPINNED:
 *   broadphase   AABB overlap of (cube, table) and (cube, tip core);
 *   narrowphase  cube-table: cube vertices within the breaking threshold of the table plane (<= 4 when resting flat);
 *                cube-tip: deepest tip-core hull vertex against the box's signed distance field (one point per tick,
 *                no persistent manifold, no warm start, no friction anchors);
 *   solver       joint motors, then contact normals (lambda >= 0, soft-contact cfm/erp for the tip), then friction pairs
 *                (implicit cone, |f| <= mu lambda_n), `iters` sweeps, in one projected Gauss-Seidel loop. */
/* Persistent contact manifold of the tip core - cube pair (narrowphase = 1; btPersistentManifold restated, PARITY_ASSUMPTIONS A36-A38):
 * up to 4 points, each with its local anchors on the tip link (la) and on the cube (lb), the world normal it was made with (from the cube
 * towards the tip), its refreshed world positions and distance. */
typedef struct {
    int32_t n;
    double la[4][3], lb[4][3], nrm[4][3], pa[4][3], pb[4][3], depth[4];
} mb_manifold;

typedef struct {
    double table_z;                    /* table top plane (base_tactile_env.py:131-139: table at z = -0.625, top at 0) */
    double half[3];                    /* cube half extents (cube.urdf: 0.08^3) */
    double mu_table, mu_tip;           /* combined friction: 0.065 x 1.0 and 0.065 x 10 (object_push_env.py:216-225, :61-66) */
    double margin_cube, margin_tip;    /* collision margins: 1e-4 (:224) and the URDF convex-hull default 1e-3 */
    double breaking;                   /* contactBreakingThreshold 1e-4 (base_tactile_env.py:128-130) */
    double erp;                        /* contact ERP 0.2 */
    double tip_stiffness, tip_damping; /* contactStiffness / contactDamping of the tip link (tactile_sensor.py:324-332) */
    double lin_damp, ang_damp;         /* cube velocity damping 0.04 (Bullet multibody default) */
    int32_t tip_link;                  /* arm link carrying the tip core */
    int32_t n_tip;
    const double* tip_verts;           /* [n_tip][3] hull vertices in that link's frame */
    int32_t cone_friction;             /* enableConeFriction = 1 */
    /* outputs of the last tick (inspection / parity): number of contacts and the tip contact */
    int32_t n_contacts;
    double tip_depth, tip_normal[3], tip_impulse;
    double residual_threshold;   /* in: PGS leaves the loop when the largest squared impulse change of a sweep is <= this (0: exact fixed point) */
    int32_t sweeps_used;         /* out */
    /* object_roll: the free body is a sphere (object_roll/sphere/sphere.urdf) under the flat TacTip, whose tip collision shape is a
     * URDF cylinder (ur5_with_flat_tactip.urdf:320-325).  shape 0: box vs hull vertices (object_push), 1: sphere vs solid cylinder. */
    int32_t shape;
    double radius;               /* sphere radius (default_obj_radius x scaling_factor) */
    double cyl_pos[3], cyl_rot[9];   /* cylinder frame (axis = local z) in the frame of tip_link */
    double cyl_half_len, cyl_radius;
    /* out: the tick's contact pairs in solver row order (n_contacts of them, the rest -1; 8 slots: 4 table + 4 manifold points), named by their feature: 0-7 = the cube
     * vertex (4 ix + 2 iy + iz) that touches the table (the marble's table contact: 0); 8 + k = hull vertex k of the tip core against
     * the cube (the marble against the tip's cylinder: 8).  north_star: "bit-exact for contact-pair indices". */
    int32_t contact_ids[8];
    /* narrowphase 0: the closed forms above (one tip point per tick).  1 (object_push only): support-mapping GJK distance / EPA penetration
     * between the tip core's hull and the box (oracle/narrowphase.c), behind an AABB overlap test of the pair, feeding a persistent manifold
     * of up to 4 points (ids 8 + manifold slot); a reset clears the manifold. */
    int32_t narrowphase;
    mb_manifold mani;
} mb_push_scene;

struct mb_spin_s {
    mb_body dish;                      /* the free object (body A of the manifold) */
    double ext_torque[3];              /* applyExternalTorque(LINK_FRAME) on the dish: used by the next tick (with dish.ext_force), then cleared */
    int32_t torque_pending;
    int32_t n_dish, n_spool;
    const double* dish_hull;           /* [n_dish][3] in the dish's base frame */
    const double* spool_hull;          /* [n_spool][3] in the spool's base frame */
    double margin, breaking, erp, mu;  /* URDF hull margin 1e-3 (both), contactBreakingThreshold 1e-4, contact ERP 0.2, friction 0.5 x 0.5 */
    double lin_damp, ang_damp;         /* the spool's: 0.04 */
    mb_manifold mani;                  /* la: on the dish, lb: on the spool, normals from the spool towards the dish */
    int32_t n_contacts;                /* out */
    double normal_impulse;             /* out: summed over the manifold's points */
};
/* Row order: joint motors, P2P x y z (these in reversed order on even sweeps), the contacts' normals, then their friction pairs. */
void mb_step_spin(const mb_model* m, mb_state* s, mb_body* spool, const mb_p2p* c, mb_spin* sp, double dt, int solver_iterations);

/* oracle/narrowphase.c.  hull [n][3] and the results in the box frame; *sdist < 0: overlap depth of the cores.  Returns 0 for touching cores. */
int mb_gjk_epa_hull_box(const double* hull, int n, const double* half, double* sdist, double* nrm, double* pa, double* pb);
int mb_gjk_epa_hull_hull(const double* hull, int n, const double* hull_b, int nb, double* sdist, double* nrm, double* pa, double* pb);
void mb_manifold_add(mb_manifold* m, double breaking, const double* oa, const double* Ra, const double* ob, const double* Rb,
                     const double* pa_w, const double* pb_w, const double* n_w, double depth);
void mb_manifold_refresh(mb_manifold* m, double breaking, const double* oa, const double* Ra, const double* ob, const double* Rb);

extern int mb_last_sweeps;   /* PGS sweeps executed by the last mb_step / mb_step_body / mb_step_body_ball / mb_step_push */
/* btContactSolverInfo::m_leastSquaresResidualThreshold for every mb_step* (PARITY_ASSUMPTIONS A7b, A7c): the solver loop leaves after the
 * sweep whose largest squared row velocity change (deltaImpulse / jacDiagABInv; a cone-friction pair counts once, with the sum of its two)
 * is <= t, never before the first sweep, at the latest after `solver_iterations`.  0 (default): only at an exact fixed point. */
void mb_set_solver_residual_threshold(double t);
double mb_get_solver_residual_threshold(void);
void mb_step_push(const mb_model* m, mb_state* s, mb_body* cube, mb_push_scene* sc, double dt, int solver_iterations);

#ifdef __cplusplus
}
#endif
#endif
