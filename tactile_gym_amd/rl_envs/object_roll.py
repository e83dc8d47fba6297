"""object_roll-v0 on the HIP path: a marble rolled on the table under the flat TacTip towards a goal given in the TCP frame.

Reference: tactile_gym/rl_envs/nonprehensile_manipulation/object_roll/object_roll_env.py (+ base_object_env.py).  The marble is a
free sphere (sphere.urdf), the flat tip collides as a URDF cylinder (ur5_with_flat_tactip.urdf:320-325); contact model and the
tessellation used for the marble's visual: PARITY_ASSUMPTIONS A30.  UR5 + TacTip, movement "xy", both control modes.
"""
import ctypes as C
import math
import os

import numpy as np

from .. import _capi as capi
from .. import pb_math as pbm
from ..robot_model import ASSETS, MeshDesc, SensorDesc, load_tgmodel, make_robot
from ..vec_env import SingleTactileEnv, TactileVecEnv

REST_POSES = {"ur5": {"flat": [0.16682, -2.23156, -1.66642, -0.81399, 1.57315, 1.74001]}}    # object_roll/rest_poses.py

env_modes_default = {  # object_roll_env.py:12-20
    "movement_mode": "xy",
    "control_mode": "TCP_velocity_control",
    "rand_init_obj_pos": False,
    "rand_obj_size": False,
    "rand_embed_dist": False,
    "observation_mode": "oracle",
    "reward_mode": "dense",
}


def build_config(num_envs, max_steps, image_size, env_modes, physics_dtype="f64", auto_reset=True, device=0):
    modes = dict(env_modes)
    for k in ("movement_mode", "control_mode", "rand_init_obj_pos", "rand_obj_size", "rand_embed_dist", "observation_mode", "reward_mode",
              "arm_type", "tactile_sensor_name"):
        if k not in modes:
            raise KeyError(k)                                                                   # object_roll_env.py:41-53
    arm, t_s_name, t_s_type = modes["arm_type"], modes["tactile_sensor_name"], "flat"           # :57
    if modes["movement_mode"] != "xy":
        raise ValueError(f"unknown movement_mode {modes['movement_mode']}")                     # get_act_dim :417-422
    if modes["control_mode"] not in capi.CONTROL:
        if modes["control_mode"] in ("joint_velocity_control",):
            raise NotImplementedError(f"control_mode {modes['control_mode']} is not built for object_roll")
        raise SystemExit(f"Incorrect control mode specified: {modes['control_mode']}")
    if arm not in REST_POSES:
        if arm in ("mg400", "franka_panda", "kuka_iiwa"):
            raise NotImplementedError(f"arm_type {arm} is not built for object_roll")
        raise SystemExit(f"Incorrect arm type specified {arm}")
    if t_s_name != "tactip":
        raise NotImplementedError("object_roll is built for the flat TacTip only (the only `flat` sensor upstream)")
    if modes["reward_mode"] not in capi.REWARD:
        raise SystemExit("Incorrect reward mode specified")
    cfg = capi.TgConfig()
    cfg.abi_version, cfg.env_kind = capi.ABI_VERSION, capi.ENV_OBJECT_ROLL
    cfg.num_envs, cfg.max_steps = int(num_envs), int(max_steps)
    cfg.movement_mode, cfg.noise_mode, cfg.reward_mode = 0, 0, capi.REWARD[modes["reward_mode"]]
    cfg.physics_dtype = capi.PHYSICS[physics_dtype]
    cfg.sim_dt = 1.0 / 240.0                                                                    # :34
    cfg.action_repeat = int(np.floor((1.0 / 10.0) / cfg.sim_dt))                                # :35-36 -> 24
    cfg.solver_iterations = 150
    cfg.auto_reset, cfg.device = int(auto_reset), int(device)
    cfg.min_action, cfg.max_action = -0.25, 0.25                                                # :110
    cfg.control_mode, cfg.max_blocking_steps = capi.CONTROL[modes["control_mode"]], 10
    v = 0.001 if modes["control_mode"] == "TCP_position_control" else 0.01                     # :113-135
    lo, hi = [-v, -v, 0.0, 0.0, 0.0, 0.0], [v, v, 0.0, 0.0, 0.0, 0.0]
    lims = [(-0.05, 0.05), (-0.05, 0.05), (-0.01, 0.01), (0.0, 0.0), (0.0, 0.0), (0.0, 0.0)]     # :74-80
    for d in range(6):
        cfg.act_lo[d], cfg.act_hi[d] = lo[d], hi[d]
        cfg.tcp_lims[d][0], cfg.tcp_lims[d][1] = lims[d]
    z = np.load(os.path.join(ASSETS, "objects", "sphere.npz"))
    radius, embed = float(z["radius"]), 0.0015                                                  # :161, :64
    wf_pos, wf_rpy = (0.65, 0.0, 2 * radius - embed), (-math.pi, 0.0, math.pi / 2)              # :70-71
    for k in range(3):
        cfg.workframe_pos[k], cfg.workframe_rpy[k] = wf_pos[k], wf_rpy[k]
        cfg.obj_init_pos[k] = (0.65, 0.0, radius)[k]                                            # :164
    cfg.termination_dist = 0.001                                                                # :60
    cfg.embed_dist, cfg.embed_lo, cfg.embed_hi = embed, 0.0015, 0.003                           # :189-190
    cfg.obj_mass, cfg.roll_radius = float(z["mass"]), radius
    cfg.roll_rand_init_pos = int(bool(modes["rand_init_obj_pos"]))
    cfg.roll_rand_size, cfg.roll_rand_embed = int(bool(modes["rand_obj_size"])), int(bool(modes["rand_embed_dist"]))
    cfg.roll_init_range = 0.009                                                                 # :210-214
    cfg.roll_goal_lo, cfg.roll_goal_hi = (0.0 if modes["rand_init_obj_pos"] else 0.005), 0.015  # :256-259
    cfg.table_z = 0.0
    cfg.mu_table, cfg.mu_tip = 10.0 * 1.0, 10.0 * 10.0                                          # :239-248 marble friction 10 x plane 1 / x tip 10 (:58)
    cfg.contact_breaking, cfg.contact_erp = 1e-4, 0.2
    cfg.tip_stiffness, cfg.tip_damping = 10.0, 100.0 + 0.1                                      # :58 t_s_dynamics [A25]
    cfg.obj_lin_damp, cfg.obj_ang_damp = 0.04, 0.04
    cfg.cone_friction = 1
    tg = load_tgmodel(arm, t_s_type, t_s_name)
    robot = make_robot(tg, REST_POSES[arm][t_s_type], t_s_name)
    r = np.load(os.path.join(ASSETS, "robots", f"{arm}_{t_s_type}_{t_s_name}.npz"))
    cfg.tip_link = int(r["tip_cyl_link"])
    for k in range(3):
        cfg.tip_cyl_pos[k] = float(r["tip_cyl_pos"][k])
    for k in range(9):
        cfg.tip_cyl_rot[k] = float(np.asarray(r["tip_cyl_rot"]).reshape(9)[k])
    cfg.tip_cyl_half_len, cfg.tip_cyl_radius = 0.5 * float(r["tip_cyl_length"]), float(r["tip_cyl_radius"])
    sensor = SensorDesc(t_s_name, t_s_type, image_size, turn_off_border=False)
    mesh = MeshDesc(z["verts"], z["tris"])
    return cfg, robot, sensor, mesh, modes


class ObjectRollVecEnv(TactileVecEnv):
    def __init__(self, num_envs, max_steps=1000, image_size=(64, 64), env_modes=env_modes_default, physics_dtype="f64", auto_reset=True,
                 device=0, obs_mode="numpy", seed=None, pgs_full_sweeps=False, solver_residual_threshold=0.0, copy_obs=True, contact_mapping="auto", solver_iterations=None):
        cfg, robot, sensor, mesh, modes = build_config(num_envs, max_steps, image_size, env_modes, physics_dtype, auto_reset, device)
        if solver_iterations is not None:
            cfg.solver_iterations = int(solver_iterations)   # numSolverIterations (base_tactile_env.py:128-130: 150); measurements only
        cfg.contact_mapping = capi.CONTACT_MAP[contact_mapping]   # "wave": one wavefront per env, "lane": one lane per env (tg_config.contact_mapping)
        cfg.pgs_full_sweeps = int(bool(pgs_full_sweeps))
        cfg.solver_residual_threshold = float(solver_residual_threshold)   # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b): 0 = exit at convergence only, 1e-7 = what PyBullet is believed to run
        self.env_modes = modes
        self.min_action, self.max_action = cfg.min_action, cfg.max_action
        super().__init__(cfg, robot, sensor, mesh, observation_mode=modes["observation_mode"], obs_mode=obs_mode, seed=seed, copy_obs=copy_obs,
                         act_dim=2, oracle_dim=34, feature_dim=3,
                         guard_spec={"arm_type": modes["arm_type"], "t_s_core": "fixed", "obj": "sphere", "every_step": True},   # object_roll_env.py:56
                         scene_spec={"arm_type": modes["arm_type"], "camera": ([0.75, 0.0, 0.00775], 0.01, 90.0, 0.0, 75.0, 0.01, 100.0)})   # :145-154                                # get_extended_feature_array :409-415

    def feature_numpy(self, terminal=False):
        buf = np.zeros((self.num_envs, 12), dtype=np.float32)                                   # device rows are 12 floats wide; 3 are used
        capi.check(self._L.tg_copy_obs_feature(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_float)), int(terminal)))
        return np.ascontiguousarray(buf[:, :3])

    def feature_torch(self, terminal=False):
        return super().feature_torch(terminal)[:, :3]

    def _episode_workframe(self, st):
        """update_workframe (:192-201): the work-frame origin follows the episode's marble radius and embed distance."""
        pos = np.tile(np.array([self._cfg.workframe_pos[k] for k in range(3)]), (self.num_envs, 1))
        pos[:, 2] = 2.0 * st["obj_mass"] - st["embed_dist"]
        return pos

    def oracle_obs_host(self):
        """get_oracle_obs (:367-407): TCP pos, orn (quaternion), lin / ang velocity, marble pos, orn, lin / ang velocity (work frame of
        the episode), goal pos and orn in the TCP frame, marble radius; float32 [N, 34]."""
        st = self.get_state()
        wpos = self._episode_workframe(st)
        rpy = np.array([self._cfg.workframe_rpy[k] for k in range(3)])
        wq = pbm.quat_from_euler(rpy)
        iq = np.array([-wq[0], -wq[1], -wq[2], wq[3]])
        Rinv = pbm.mat_from_quat(iq)

        def to_work(pos, rpy_world):
            p = (pos - wpos) @ Rinv.T
            q = pbm.quat_mul(np.tile(iq, (self.num_envs, 1)), pbm.quat_from_euler(rpy_world))
            return p, pbm.quat_from_euler(pbm.euler_from_quat(q))

        from .. import hip_ops
        J, tpos, trot = hip_ops.jacobian_tcp(self._robot, st["q"], dtype="f64")
        tp, tq = to_work(tpos, pbm.euler_from_quat(pbm.quat_from_mat(trot)))
        tl = np.einsum("nij,nj->ni", J[:, :3, :], st["qd"]) @ Rinv.T
        ta = np.einsum("nij,nj->ni", J[:, 3:, :], st["qd"]) @ Rinv.T
        orpy = pbm.euler_from_quat(pbm.quat_from_mat(st["body_rot"]))
        op, oq = to_work(st["body_pos"], orpy)
        ol, oa = st["body_linvel"] @ Rinv.T, st["body_angvel"] @ Rinv.T
        ident = np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (self.num_envs, 1))
        return np.hstack([tp, tq, tl, ta, op, oq, ol, oa, st["goal_pos"], ident, st["obj_mass"][:, None]]).astype(np.float32)


class ObjectRollEnv(SingleTactileEnv):
    """Single-env gym.Env surface; constructor signature as object_roll_env.py:24-31."""

    vec_cls = ObjectRollVecEnv
    default_env_modes = env_modes_default

    def __init__(self, max_steps=1000, image_size=[64, 64], env_modes=env_modes_default, show_gui=False, show_tactile=False, **kwargs):
        super().__init__(max_steps, image_size, env_modes, show_gui, show_tactile, **kwargs)
