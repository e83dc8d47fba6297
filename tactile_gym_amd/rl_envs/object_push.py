"""object_push-v0 on the HIP path.

Reference: tactile_gym/rl_envs/nonprehensile_manipulation/object_push/object_push_env.py on top of base_object_env.py:
an arm with a right-angle tactile sensor pushes a cube across the table along a per-episode trajectory of goals.  This is
the one task whose tip has its collision core enabled (t_s_core "fixed", :48), so it is the task that needs rigid contacts:
cube-table and cube-tip, restated in tg_physics.hpp:sim_tick_push [PARITY_ASSUMPTIONS A23-A28].
"""
import ctypes as C
import math
import os

import numpy as np

from .. import _capi as capi
from ..robot_model import ASSETS, MeshDesc, SensorDesc, load_tgmodel, make_robot
from ..vec_env import SingleTactileEnv, TactileVecEnv

REST_POSES = {  # object_push/rest_poses.py, control-joint order; MG400 + TacTip carries the `mini_right_angle` sensor (object_push_env.py:70-75)
    "mg400": {
        "tactip": [-0.4675810386176251, 1.2330268637269028, -0.042146321181746195, -1.1915354526403177, 0.4668115359824357,
                   1.2330268635741901, -1.2330268635741901, 1.1908822248875286],
        "digit": [-0.4558165479388624, 1.2857227247064174, 0.26532296230426017, -1.5518769541832729, 0.45743009274925944,
                  1.28573249852019, -1.2857285129498681, 1.5510764390458196],
        "digitac": [-0.4745979999944637, 1.2836350191938928, 0.254159419927845, -1.5395417027560878, 0.47634420683617346,
                    1.2838656861791102, -1.283854805915325, 1.5380912693333302],
    },
    "ur5": {
        "tactip": [-0.29446578243858357, -2.1633703222876646, -1.7712875440608364, -0.7758826291678864, 1.569501010720629, -1.8628739133606422],
        "digit": [-0.2363248329397155, -2.1381281530498035, -1.8208841358171288, -0.751838113524854, 1.5711258995033301, -1.80239847761509],
        "digitac": [-0.24571108391609556, -2.142076416487341, -1.8135315230114846, -0.7552488203413393, 1.5711290394202047, -1.8118003855516092],
    },
}

env_modes_default = {  # object_push_env.py:11-20
    "movement_mode": "TyRz",
    "control_mode": "TCP_velocity_control",
    "rand_init_orn": False,
    "rand_obj_mass": False,
    "traj_type": "simplex",
    "observation_mode": "oracle",
    "reward_mode": "dense",
}

# contactStiffness, contactDamping, lateralFriction of the tip core (object_push_env.py:50-56)
TIP_DYNAMICS = {"tactip": (50.0, 100.0, 10.0), "digitac": (300.0, 100.0, 10.0), "digit": (50.0, 200.0, 10.0)}


def _goal_reached_at_reset(wf_pos, wf_rpy, obj_init_pos, init_offset, term_dist):
    """reset() ends with get_step_data() (base_object_env.py:183-185): termination() (object_push_env.py:520-537) advances the goal if the
    cube is closer than termination_pos_dist to it.  Goal 0 = work-frame (obj_width/2 + spacing, 0, 0) lies exactly that far from the cube's
    start position, so the outcome is a rounding question of the configuration's constants.  Evaluated here in double the way the reference
    chain does (getQuaternionFromEuler -> multiplyTransforms -> np.linalg.norm) [PARITY_ASSUMPTIONS A29]."""
    phi, the, psi = (0.5 * float(v) for v in wf_rpy)
    q = np.array([math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi),
                  math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi),
                  math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi),
                  math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi)])
    x, y, z, w = (float(v) for v in q / math.sqrt(float(q @ q)))
    sc = 2.0 / (x * x + y * y + z * z + w * w)
    xs, ys, zs = x * sc, y * sc, z * sc
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    R = np.array([[1.0 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1.0 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1.0 - (xx + yy)]])
    goal0 = np.asarray(wf_pos, dtype=np.float64) + R @ np.array([init_offset, 0.0, 0.0])
    return bool(float(np.linalg.norm(np.asarray(obj_init_pos, dtype=np.float64) - goal0)) < term_dist)


def build_config(num_envs, max_steps, image_size, env_modes, physics_dtype="f64", auto_reset=True, device=0, inertia_mode="collision_aabb"):
    modes = dict(env_modes)
    for k in ("movement_mode", "control_mode", "rand_init_orn", "rand_obj_mass", "traj_type", "observation_mode", "reward_mode", "arm_type",
              "tactile_sensor_name"):
        if k not in modes:
            raise KeyError(k)                                                                   # object_push_env.py:34-43
    arm, t_s_name, t_s_type = modes["arm_type"], modes["tactile_sensor_name"], "right_angle"    # :48
    if arm == "mg400" and t_s_name == "tactip":
        t_s_type = "mini_right_angle"                                                           # :70-75
    if modes["movement_mode"] not in capi.PMOVE:
        raise ValueError(f"unknown movement_mode {modes['movement_mode']}")
    if modes["traj_type"] not in capi.TRAJ:
        raise SystemExit(f"Incorrect traj_type specified: {modes['traj_type']}")               # :339
    if modes["control_mode"] not in capi.CONTROL:
        if modes["control_mode"] in ("joint_velocity_control",):
            raise NotImplementedError(f"control_mode {modes['control_mode']} is outside the built hot path (SURVEY 8f rank 2)")
        raise SystemExit(f"Incorrect control mode specified: {modes['control_mode']}")
    if arm not in REST_POSES:
        if arm in ("franka_panda", "kuka_iiwa"):
            raise NotImplementedError(f"arm_type {arm} is not built yet for object_push")
        raise SystemExit(f"Incorrect arm type specified {arm}")
    if t_s_name not in TIP_DYNAMICS:
        raise SystemExit(f"Incorrect sensor specified {t_s_name}")
    cfg = capi.TgConfig()
    cfg.abi_version, cfg.env_kind = capi.ABI_VERSION, capi.ENV_OBJECT_PUSH
    cfg.num_envs, cfg.max_steps = int(num_envs), int(max_steps)
    cfg.movement_mode, cfg.noise_mode, cfg.reward_mode = capi.PMOVE[modes["movement_mode"]], 0, capi.REWARD[modes["reward_mode"]]
    cfg.physics_dtype = capi.PHYSICS[physics_dtype]
    cfg.sim_dt = 1.0 / 240.0                                                                    # :30
    cfg.action_repeat = int(np.floor((1.0 / 10.0) / cfg.sim_dt))                                # :31-32 -> 24
    cfg.solver_iterations = 150
    cfg.auto_reset, cfg.device = int(auto_reset), int(device)
    cfg.min_action, cfg.max_action = -0.25, 0.25                                                # :116
    v, w = 0.01, 5.0 * (math.pi / 180)                                                          # :126-134
    cfg.control_mode, cfg.max_blocking_steps = capi.CONTROL[modes["control_mode"]], 10
    if modes["control_mode"] == "TCP_position_control":
        v, w = 0.001, 1 * (math.pi / 180)                                                       # :137-148 m / rad per step
    lo, hi = [-v, -v, 0.0, 0.0, 0.0, -w], [v, v, 0.0, 0.0, 0.0, w]
    a = 45 * math.pi / 180
    lims = [(-0.0, 0.3), (-0.1, 0.08 if arm == "mg400" else 0.1), (-0.0, 0.0), (-0.0, 0.0), (-0.0, 0.0), (-a, a)]   # :62-68 mg400, :81-87 ur5
    for d in range(6):
        cfg.act_lo[d], cfg.act_hi[d] = lo[d], hi[d]
        cfg.tcp_lims[d][0], cfg.tcp_lims[d][1] = lims[d]
    obj_w = obj_h = 0.08                                                                        # :45-46
    if arm == "mg400" and t_s_name != "tactip":                                                 # :70-79
        wd = (0.25, -0.1, obj_h / 2)
    elif arm == "mg400":
        wd = (0.30, -0.1, obj_h / 2)
    else:
        wd = (0.55, -0.20, obj_h / 2)
    wf_rpy = (-math.pi, 0.0, math.pi / 2)                                                       # :88
    init_pos = (wd[0], wd[1] + obj_w / 2, obj_h / 2)                                            # :160
    for k in range(3):
        cfg.workframe_pos[k], cfg.workframe_rpy[k] = wd[k], wf_rpy[k]
        cfg.obj_init_pos[k], cfg.obj_init_rpy[k] = init_pos[k], (-math.pi, 0.0, math.pi / 2)[k]  # :158
        cfg.obj_half[k] = obj_w / 2
    cfg.termination_dist = 0.025                                                                # :57
    suffix = "" if inertia_mode == "collision_aabb" else "_urdfinertia"
    z = np.load(os.path.join(ASSETS, "objects", f"cube{suffix}.npz"))
    cfg.obj_mass = float(z["mass"])
    for k in range(3):
        cfg.obj_com[k] = float(z["com"][k])
    for k in range(9):
        cfg.obj_inertia[k] = float(z["inertia"].reshape(9)[k])
    stiff, damp, tip_mu = TIP_DYNAMICS[t_s_name]
    cfg.table_z = 0.0
    cfg.mu_table, cfg.mu_tip = 0.065 * 1.0, 0.065 * tip_mu                                      # :216-225, Bullet multiplies the pair's frictions
    cfg.margin_cube, cfg.margin_tip, cfg.contact_breaking, cfg.contact_erp = 1e-4, 1e-3, 1e-4, 0.2   # PARITY_ASSUMPTIONS A24
    cfg.tip_stiffness, cfg.tip_damping = stiff, damp + 0.1                                      # A25: damping combines additively with the cube's 0.1
    cfg.obj_lin_damp, cfg.obj_ang_damp = 0.04, 0.04
    cfg.cone_friction = 1
    cfg.traj_type, cfg.traj_n_points = capi.TRAJ[modes["traj_type"]], 10                        # :229
    cfg.traj_spacing, cfg.traj_max_perturb = 0.025, 0.1                                         # :230-231
    cfg.traj_init_offset = obj_w / 2 + cfg.traj_spacing                                         # :262
    cfg.reset_goal_id = int(_goal_reached_at_reset(wd, wf_rpy, init_pos, cfg.traj_init_offset, cfg.termination_dist))
    cfg.rand_init_orn, cfg.rand_obj_mass = int(bool(modes["rand_init_orn"])), int(bool(modes["rand_obj_mass"]))
    cfg.mass_lo, cfg.mass_hi = 0.4, 0.8                                                         # :190-192
    cfg.init_orn_range, cfg.traj_ang_range = math.pi / 32, math.pi / 8                          # :170, :283
    tg = load_tgmodel(arm, t_s_type, t_s_name, inertia_mode)
    robot = make_robot(tg, REST_POSES[arm][t_s_name], t_s_name)
    r = np.load(os.path.join(ASSETS, "robots", f"{arm}_{t_s_type}_{t_s_name}{suffix}.npz"))
    tip_verts = np.ascontiguousarray(r["tip_hull_verts"], dtype=np.float64)
    cfg.tip_link, cfg.n_tip_verts = int(r["tip_hull_link"]), tip_verts.shape[0]
    cfg.tip_verts = tip_verts.ctypes.data_as(C.POINTER(C.c_double))
    sensor = SensorDesc(t_s_name, t_s_type, image_size, turn_off_border=False)
    mesh = MeshDesc(z["verts"], z["tris"])
    return cfg, robot, sensor, mesh, modes, tip_verts


class ObjectPushVecEnv(TactileVecEnv):
    def __init__(self, num_envs, max_steps=1000, image_size=(64, 64), env_modes=env_modes_default, physics_dtype="f64", auto_reset=True,
                 device=0, obs_mode="numpy", seed=None, pgs_full_sweeps=False, solver_residual_threshold=0.0, copy_obs=True, contact_mapping="auto", solver_iterations=None, narrowphase="closed_form",
                 max_force=None):
        cfg, robot, sensor, mesh, modes, tip_verts = build_config(num_envs, max_steps, image_size, env_modes, physics_dtype, auto_reset, device)
        if max_force is not None:
            robot.max_force = float(max_force)               # the arm's motor force limit (ur5.py:19 / mg400.py:27: 1000)
        if solver_iterations is not None:
            cfg.solver_iterations = int(solver_iterations)   # numSolverIterations (base_tactile_env.py:128-130: 150); measurements only
        cfg.narrowphase = capi.NARROWPHASE[narrowphase]   # "gjk_manifold": GJK / EPA + Bullet's persistent manifold for the tip - cube pair (tg_config.narrowphase)
        cfg.contact_mapping = capi.CONTACT_MAP[contact_mapping]   # "wave": one wavefront per env, "lane": one lane per env (tg_config.contact_mapping)
        cfg.pgs_full_sweeps = int(bool(pgs_full_sweeps))   # run all solver sweeps instead of leaving at convergence
        cfg.solver_residual_threshold = float(solver_residual_threshold)   # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b): 0 = exit at convergence only, 1e-7 = what PyBullet is believed to run
        self._tip_verts = tip_verts   # tg_create copies them; kept only until then
        self.env_modes = modes
        self.min_action, self.max_action = cfg.min_action, cfg.max_action
        act_dim = {"y": 1, "yRz": 2, "xyRz": 3, "TyRz": 2, "TxTyRz": 3}[modes["movement_mode"]]  # :631-644
        super().__init__(cfg, robot, sensor, mesh, observation_mode=modes["observation_mode"], obs_mode=obs_mode, seed=seed, copy_obs=copy_obs,
                         act_dim=act_dim, oracle_dim=30, feature_dim=12,
                         guard_spec={"arm_type": modes["arm_type"], "t_s_core": "fixed", "obj": "cube", "every_step": True},   # object_push_env.py:60
                         scene_spec={"arm_type": modes["arm_type"], "camera": ([0.1, 0.0, -0.35], 1.0, 90.0, -45.0, 75.0, 0.1, 100.0)})   # :170-179

    def oracle_obs_host(self):
        """object_push_env.py:571-609: TCP pos, rpy, lin/ang velocity, cube pos, rpy, lin/ang velocity and the current goal pos, rpy,
        all in the work frame; float32 [N, 30]."""
        st = self.get_state()
        tp, tr, _, tl, ta = self._tcp_workframe_state(st)
        op, orr, _, ol, oa = self._obj_workframe_state(st)
        gi = np.minimum(st["goal_id"], self._cfg.traj_n_points - 1)
        idx = np.arange(self.num_envs)
        gpos = np.stack([st["traj"][idx, 0, gi], st["traj"][idx, 1, gi], np.zeros(self.num_envs)], axis=1)
        grpy = np.stack([np.zeros(self.num_envs), np.zeros(self.num_envs), st["traj"][idx, 2, gi]], axis=1)
        return np.hstack([tp, tr, tl, ta, op, orr, ol, oa, gpos, grpy]).astype(np.float32)


class ObjectPushEnv(SingleTactileEnv):
    """Single-env gym.Env surface; constructor signature as object_push_env.py:24-31."""

    vec_cls = ObjectPushVecEnv
    default_env_modes = env_modes_default

    def __init__(self, max_steps=1000, image_size=[64, 64], env_modes=env_modes_default, show_gui=False, show_tactile=False, **kwargs):
        super().__init__(max_steps, image_size, env_modes, show_gui, show_tactile, **kwargs)
