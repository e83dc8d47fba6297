"""object_balance-v0 (object_mode "pole", "ball_on_plate" and "spinning_plate") on the HIP path.

Reference: tactile_gym/rl_envs/nonprehensile_manipulation/object_balance/object_balance_env.py on top of
base_object_env.py: a UR5 + TacTip pointing up carries a pole tied to its TCP by a point-to-point constraint; the agent
moves the TCP to keep the pole upright.  "ball_on_plate" (:187-199, 241-260): the round plate instead of the pole and a ball
rolling on it (one ball - plate contact, sim_tick_body_ball).  "spinning_plate" (:107-108, 198-239, 267-269, 355-358): the spool
(plate_buffer.urdf) on the constraint and the dish (the env's object) on its spindle - two convex hulls through the wave-mapped GJK / EPA and
a persistent manifold (csrc/tg_spin.hip).
"""
import math
import os

import numpy as np

from .. import _capi as capi
from ..robot_model import ASSETS, MeshDesc, SensorDesc, load_tgmodel, make_robot
from ..vec_env import SingleTactileEnv, TactileVecEnv

REST_POSES = {"ur5": {"standard": [0.19826, -2.01062, -1.96602, -0.73808, 4.71286, -3.34064]}}   # object_balance/rest_poses.py:4-20

env_modes_default = {  # object_balance_env.py:11-19
    "movement_mode": "xy",
    "control_mode": "TCP_velocity_control",
    "object_mode": "pole",
    "rand_gravity": False,
    "rand_embed_dist": False,
    "observation_mode": "oracle",
    "reward_mode": "dense",
}


def build_config(num_envs, max_steps, image_size, env_modes, physics_dtype="f64", auto_reset=True, device=0, inertia_mode="collision_aabb"):
    modes = dict(env_modes)
    for k in ("movement_mode", "control_mode", "object_mode", "rand_gravity", "rand_embed_dist", "observation_mode", "reward_mode",
              "arm_type", "tactile_sensor_name"):
        if k not in modes:
            raise KeyError(k)                                                                   # object_balance_env.py:37-44
    arm, t_s_name, t_s_type = modes["arm_type"], modes["tactile_sensor_name"], "standard"       # :46
    if modes["object_mode"] not in capi.BALANCE_OBJECT:
        raise ValueError(f"unknown object_mode {modes['object_mode']}")
    ball_mode, spin_mode = modes["object_mode"] == "ball_on_plate", modes["object_mode"] == "spinning_plate"
    if spin_mode and (physics_dtype != "f64" or inertia_mode != "collision_aabb"):
        raise NotImplementedError("object_mode spinning_plate is built for f64 physics and collision-shape inertias")
    if modes["movement_mode"] not in capi.BMOVE:
        raise ValueError(f"unknown movement_mode {modes['movement_mode']}")
    if modes["control_mode"] not in capi.CONTROL:
        if modes["control_mode"] in ("joint_velocity_control",):
            raise NotImplementedError(f"control_mode {modes['control_mode']} is outside the built hot path (SURVEY 8f rank 2)")
        raise SystemExit(f"Incorrect control mode specified: {modes['control_mode']}")
    if arm != "ur5":
        if arm in ("mg400", "franka_panda", "kuka_iiwa"):
            raise NotImplementedError(f"arm_type {arm} is not built yet for object_balance")
        raise SystemExit(f"Incorrect arm type specified {arm}")
    cfg = capi.TgConfig()
    cfg.abi_version, cfg.env_kind = capi.ABI_VERSION, capi.ENV_OBJECT_BALANCE
    cfg.num_envs, cfg.max_steps = int(num_envs), int(max_steps)
    cfg.movement_mode, cfg.noise_mode, cfg.reward_mode = capi.BMOVE[modes["movement_mode"]], 0, capi.REWARD[modes["reward_mode"]]
    cfg.physics_dtype = capi.PHYSICS[physics_dtype]
    cfg.sim_dt = 1.0 / 240.0                                                                    # :33
    cfg.action_repeat = int(np.floor((1.0 / 20.0) / cfg.sim_dt))                                # :34-35 -> 12
    cfg.solver_iterations = 150
    cfg.auto_reset, cfg.device = int(auto_reset), int(device)
    cfg.min_action, cfg.max_action = -0.25, 0.25                                                # :111
    v, w = 0.01, 5.0 * (math.pi / 180)                                                          # :123-131
    cfg.control_mode, cfg.max_blocking_steps = capi.CONTROL[modes["control_mode"]], 10
    if modes["control_mode"] == "TCP_position_control":
        v, w = 0.001, 1 * (math.pi / 180)                                                       # :129-140 m / rad per step
    lo, hi = [-v, -v, -v, -w, -w, 0.0], [v, v, v, w, w, 0.0]
    a = 45 * math.pi / 180
    lims = [(-0.1, 0.1)] * 3 + [(-a, a)] * 3                                                    # :64-76
    for d in range(6):
        cfg.act_lo[d], cfg.act_hi[d] = lo[d], hi[d]
        cfg.tcp_lims[d][0], cfg.tcp_lims[d][1] = lims[d]
    wf_pos, wf_rpy = (0.55, 0.0, 0.35), (0.0, 0.0, 0.0)                                         # :61-62
    for k in range(3):
        cfg.workframe_pos[k], cfg.workframe_rpy[k] = wf_pos[k], wf_rpy[k]
    cfg.embed_dist = {"tactip": 0.0035, "digitac": 0.0015, "digit": 0.0015}[t_s_name]           # :53-58
    cfg.embed_lo, cfg.embed_hi = {"tactip": (0.003, 0.006), "digitac": (0.001, 0.0025), "digit": (0.0015, 0.0025)}[t_s_name]   # :308-316
    cfg.rand_gravity, cfg.rand_embed = int(bool(modes["rand_gravity"])), int(bool(modes["rand_embed_dist"]))
    cfg.gravity_lo, cfg.gravity_hi, cfg.gravity_default = -1.0, -0.1, -0.1                      # :301-306
    suffix = "" if inertia_mode == "collision_aabb" else "_urdfinertia"
    name = "plate_buffer" if spin_mode else f"{'round_plate' if ball_mode else 'pole'}{suffix}"    # spinning_plate: the body on the constraint is the spool
    z = np.load(os.path.join(ASSETS, "objects", f"{name}.npz"))
    cfg.obj_mass = float(z["mass"])
    for k in range(3):
        cfg.obj_com[k] = float(z["com"][k])
        cfg.obj_root_inertial_pos[k] = float(z["root_inertial_pos"][k])
        cfg.obj_init_rpy[k] = [0.0, 0.0, -math.pi / 2][k]                                       # :190
        cfg.ext_force[k] = [0.0, 0.0, -1.0 if spin_mode else -0.1][k]                           # :347 / :358 force_mag 0.1 / 1.0, direction (0,0,-1) :374-375
    for k in range(9):
        cfg.obj_inertia[k] = float(z["inertia"].reshape(9)[k])
    cfg.obj_base_width, cfg.obj_base_height = (0.2 if ball_mode else 0.1), 0.0025               # :158-159, :188-189
    if spin_mode:                                                                               # :198-213, load_plate_buffer :223-239
        cfg.obj_base_width, cfg.obj_base_height, cfg.spin_buffer_height = 0.15, 0.0267, 0.026
        zd = np.load(os.path.join(ASSETS, "objects", "spinning_plate.npz"))
        cfg.spin_dish_mass = float(zd["mass"])
        for k in range(3):
            cfg.spin_dish_com[k] = float(zd["com"][k])
        for k in range(9):
            cfg.spin_dish_inertia[k] = float(zd["inertia"].reshape(9)[k])
        cfg.spin_hull_margin, cfg.spin_mu = 1e-3, 0.5 * 0.5                                     # URDF hull margin; default lateral frictions [A26, A41]
        dish_hull = np.ascontiguousarray(zd["hull"], dtype=np.float64)
        spool_hull = np.ascontiguousarray(z["hull"], dtype=np.float64)
        cfg.spin_n_dish, cfg.spin_n_spool = len(dish_hull), len(spool_hull)
        cfg.spin_dish_hull = dish_hull.ctypes.data_as(capi.C.POINTER(capi.C.c_double))
        cfg.spin_spool_hull = spool_hull.ctypes.data_as(capi.C.POINTER(capi.C.c_double))
        cfg._spin_keep = (dish_hull, spool_hull)                                                # (tg_create copies them)
        cfg.contact_breaking, cfg.contact_erp = 1e-4, 0.2                                       # base_tactile_env.py:128-130; [A24]
        cfg.obj_lin_damp, cfg.obj_ang_damp = 0.04, 0.04                                         # the spool keeps Bullet's defaults (:338-345 clear the dish's only)
        cfg.cone_friction = 1
    cfg.balance_object = capi.BALANCE_OBJECT[modes["object_mode"]]
    if ball_mode:                                                                               # load_ball :241-260
        zb = np.load(os.path.join(ASSETS, "objects", "balance_ball.npz"))
        cfg.ball_radius, cfg.ball_mass = float(zb["radius"]) * 7.5, float(zb["mass"])           # globalScaling 7.5 scales the shape, not the mass [A30]
        cfg.ball_mu = 10.0 * 0.5                                                                # lateralFriction=10 (:259) x the plate's default 0.5 [A26]
        cfg.plate_radius = float(zb["plate_radius"])
        cfg.contact_breaking, cfg.contact_erp = 1e-4, 0.2                                       # [A24]
        cfg.obj_lin_damp, cfg.obj_ang_damp = 0.04, 0.04                                         # Bullet's defaults: the ball's damping is never changed [A27]
        cfg.cone_friction = 1                                                                   # base_tactile_env.py:128-130
    cfg.term_deg, cfg.term_pos = 35.0, 0.1                                                      # :50-51
    cfg.p2p_erp, cfg.p2p_max_impulse = 0.2, 500.0                                               # PARITY_ASSUMPTIONS A18-A19
    tg = load_tgmodel(arm, t_s_type, t_s_name, inertia_mode)
    robot = make_robot(tg, REST_POSES[arm][t_s_type], t_s_name)
    sensor = SensorDesc(t_s_name, t_s_type, image_size, turn_off_border=False)
    mesh = MeshDesc(z["verts"], z["tris"])
    return cfg, robot, sensor, mesh, modes


class ObjectBalanceVecEnv(TactileVecEnv):
    def __init__(self, num_envs, max_steps=1000, image_size=(64, 64), env_modes=env_modes_default, physics_dtype="f64", auto_reset=True,
                 device=0, obs_mode="numpy", seed=None, pgs_full_sweeps=False, solver_residual_threshold=0.0, copy_obs=True, contact_mapping="auto", reset_bank="auto"):
        cfg, robot, sensor, mesh, modes = build_config(num_envs, max_steps, image_size, env_modes, physics_dtype, auto_reset, device)
        if modes["object_mode"] == "ball_on_plate" and contact_mapping == "wave":
            raise ValueError("object_mode ball_on_plate runs on the lane mapping (contact_mapping 'auto' or 'lane')")
        cfg.reset_bank = capi.RESET_BANK[reset_bank]              # "off": every reset recomputes the arm's blocking move (k_reset_body's template)
        cfg.contact_mapping = capi.CONTACT_MAP[contact_mapping]   # "wave": one wavefront per env (k_step_body_wave), "lane": one lane per env
        cfg.pgs_full_sweeps = int(bool(pgs_full_sweeps))   # run all solver sweeps instead of leaving at convergence
        cfg.solver_residual_threshold = float(solver_residual_threshold)   # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b): 0 = exit at convergence only, 1e-7 = what PyBullet is believed to run
        self.env_modes = modes
        self.min_action, self.max_action = cfg.min_action, cfg.max_action
        act_dim = {"xy": 2, "xyz": 3, "RxRy": 2, "xyRxRy": 4}[modes["movement_mode"]]           # :565-576
        super().__init__(cfg, robot, sensor, mesh, observation_mode=modes["observation_mode"], obs_mode=obs_mode, seed=seed, copy_obs=copy_obs,
                         act_dim=act_dim, oracle_dim=26,
                         guard_spec={"arm_type": modes["arm_type"], "t_s_core": "no_core",       # object_balance_env.py:54
                                     "obj": {"ball_on_plate": "round_plate", "pole": "pole"}.get(modes["object_mode"]),   # (spinning_plate: the arm and the table only)
                                     "ball_radius": cfg.ball_radius if modes["object_mode"] == "ball_on_plate" else None},
                         scene_spec={"arm_type": modes["arm_type"], "camera": ([-0.1, 0.0, 0.25], 1.0, 90.0, -10.0, 75.0, 0.1, 100.0)})   # :162-171

    def oracle_obs_host(self):
        """object_balance_env.py:528-563: TCP pos, orn (quaternion), lin/ang velocity and the pole's pos, orn, lin/ang velocity, all in
        the work frame; float32 [N, 26]."""
        st = self.get_state()
        if "dish_state" in st:                                    # spinning_plate: the env's object is the dish, body_* the spool
            d = st["dish_state"]
            st = dict(st, body_pos=d[:, 0:3], body_rot=d[:, 3:12].reshape(-1, 3, 3), body_linvel=d[:, 12:15], body_angvel=d[:, 15:18])
        tp, _, tq, tl, ta = self._tcp_workframe_state(st)
        op, _, oq, ol, oa = self._obj_workframe_state(st)
        return np.hstack([tp, tq, tl, ta, op, oq, ol, oa]).astype(np.float32)


class ObjectBalanceEnv(SingleTactileEnv):
    """Single-env gym.Env surface; constructor signature as object_balance_env.py:23-30."""

    vec_cls = ObjectBalanceVecEnv
    default_env_modes = env_modes_default

    def __init__(self, max_steps=1000, image_size=[64, 64], env_modes=env_modes_default, show_gui=False, show_tactile=False, **kwargs):
        super().__init__(max_steps, image_size, env_modes, show_gui, show_tactile, **kwargs)
