"""edge_follow-v0 on the HIP path.

Reference: tactile_gym/rl_envs/exploration/edge_follow/edge_follow_env.py (task constants, action space, reset
randomisation, reward) on top of base_tactile_env.py (step / observation plumbing).  All per-step arithmetic runs in
libtactile_gym_hip.so; this module only assembles the constants the kernels need and presents the gym.Env /
VecEnv surface (old 4-tuple gym API, base_tactile_env.py:185).
"""
import math

import numpy as np

from .. import _capi as capi
from ..robot_model import MeshDesc, SensorDesc, load_tgmodel, make_robot
from ..vec_env import SingleTactileEnv, TactileVecEnv

# rest_poses.py:6-20 etc. — movable joints only, [arm][sensor][type]
REST_POSES = {
    "ur5": {
        "tactip": {"standard": [0.166827, -2.16515, -1.64365, -0.90317, 1.57315, 1.74001]},
        "digit": {"standard": [0.1666452116249431, -2.2334888481855204, -1.6642245054428424, -0.8142762445463524,
                               1.573151527964482, 1.7398309441833082]},
        "digitac": {"standard": [0.16664443404149898, -2.2242489977536737, -1.6618744232210114, -0.8258663681806591,
                                 1.5731514988184077, 1.7398302172182332]},
    },
    # control joints (j1, j2_1, j3_1, j4_1, j5, j2_2, j3_2, j4_2) = URDF joints 0-4, 9-11 of rest_poses.py:95-153
    "mg400": {
        "tactip": {"standard": [0.0, 1.1199979523765513, -0.027746434948259045, -1.094390587897371, 0.000795099112695166,
                                1.120002713232204, -1.1199729024887553, 1.0922685386653785]},
        "digit": {"standard": [0.0, 1.3190166816731614, -0.057932730559221525, -1.2611243932983605, 0.0006084288058448784,
                               1.3190195840338783, -1.3189925313906967, 1.2610906509351185]},
        "digitac": {"standard": [0.0, 1.3223687315585777, -0.06290495221125363, -1.2594762221064615, 0.0006084288058448784,
                                 1.3223720640647498, -1.3223720640647498, 1.2594757646221153]},
    },
}

env_modes_default = {  # edge_follow_env.py:12-19 (the reference default omits tactile_sensor_name and cannot be constructed)
    "movement_mode": "xy",
    "control_mode": "TCP_velocity_control",
    "noise_mode": "fixed_height",
    "observation_mode": "oracle",
    "reward_mode": "dense",
    "arm_type": "ur5",
    "tactile_sensor_name": "tactip",
}


def build_config(num_envs, max_steps, image_size, env_modes, physics_dtype="f64", auto_reset=True, device=0):
    """env ctor kwargs -> (tg_config, tg_robot, SensorDesc, MeshDesc).  Line refs: edge_follow_env.py."""
    modes = dict(env_modes)
    for k in ("movement_mode", "control_mode", "noise_mode", "observation_mode", "reward_mode", "arm_type", "tactile_sensor_name"):
        if k not in modes:
            raise KeyError(k)  # same failure as the reference when a mode is missing (:45-59)
    arm, t_s_name, t_s_type = modes["arm_type"], modes["tactile_sensor_name"], "standard"      # :52,59-64
    if modes["control_mode"] not in capi.CONTROL:
        if modes["control_mode"] in ("joint_velocity_control",):
            raise NotImplementedError(f"control_mode {modes['control_mode']} is outside the built hot path (SURVEY 8f rank 2)")
        raise SystemExit(f"Incorrect control mode specified: {modes['control_mode']}")             # robot.py:174
    if arm not in ("ur5", "mg400"):
        if arm in ("franka_panda", "kuka_iiwa"):
            raise NotImplementedError(f"arm_type {arm} is out of scope (SURVEY section 2: preliminary upstream)")
        raise SystemExit(f"Incorrect arm type specified {arm}")                                  # robot.py:65
    mg = arm == "mg400"
    if modes["movement_mode"] not in capi.MOVE:
        raise ValueError(f"unknown movement_mode {modes['movement_mode']}")
    cfg = capi.TgConfig()
    cfg.abi_version, cfg.env_kind = capi.ABI_VERSION, capi.ENV_EDGE_FOLLOW
    cfg.num_envs, cfg.max_steps = int(num_envs), int(max_steps)
    cfg.movement_mode = capi.MOVE[modes["movement_mode"]]
    cfg.noise_mode = capi.NOISE.get(modes["noise_mode"], 0)
    cfg.reward_mode = capi.REWARD[modes["reward_mode"]]
    cfg.physics_dtype = capi.PHYSICS[physics_dtype]
    sim_dt, control_rate = 1.0 / 240.0, 1.0 / 10.0                                               # :33-34
    cfg.sim_dt = sim_dt
    cfg.action_repeat = int(np.floor(control_rate / sim_dt))                                     # :35-37 -> 24
    cfg.solver_iterations = 150                                                                  # base_tactile_env.py:128-130
    cfg.auto_reset, cfg.device = int(auto_reset), int(device)
    cfg.min_action, cfg.max_action = -0.25, 0.25                                                 # :140
    cfg.control_mode, cfg.max_blocking_steps = capi.CONTROL[modes["control_mode"]], 10           # :38
    if modes["control_mode"] == "TCP_position_control":
        max_pos, max_ang = 0.001, 1 * (math.pi / 180)                                            # :143-153 m / rad per step
    else:
        max_pos, max_ang = 0.01, 5.0 * (math.pi / 180)                                           # :155-166 m/s, rad/s
    lo = [-max_pos] * 3 + [0.0, 0.0, -max_ang]
    hi = [max_pos] * 3 + [0.0, 0.0, max_ang]
    xy = (0.150, 0.11) if mg else (0.175, 0.175)                                                 # :75-90
    lims = [(-xy[0], xy[0]), (-xy[1], xy[1]), (-0.1, 0.1), (0.0, 0.0), (0.0, 0.0), (-math.pi, math.pi)]
    for d in range(6):
        cfg.act_lo[d], cfg.act_hi[d] = lo[d], hi[d]
        cfg.tcp_lims[d][0], cfg.tcp_lims[d][1] = lims[d]
    edge_pos, edge_height, edge_len = ((0.33, 0.0, 0.0) if mg else (0.65, 0.0, 0.0)), 0.035, (0.105 if mg else 0.175)   # :76,84,201-207
    wf_pos, wf_rpy = (edge_pos[0], edge_pos[1], edge_height), (-math.pi, 0.0, math.pi / 2)       # :106-107
    for k in range(3):
        cfg.workframe_pos[k], cfg.workframe_rpy[k], cfg.stim_pos[k] = wf_pos[k], wf_rpy[k], edge_pos[k]
    cfg.edge_height, cfg.edge_len, cfg.termination_dist = edge_height, edge_len, 0.01            # :67
    cfg.embed_dist = 0.0035                                                                      # :94-99
    cfg.embed_lo, cfg.embed_hi = {"tactip": (0.0015, 0.0065), "digit": (0.0011, 0.0028), "digitac": (0.0015, 0.0045)}[t_s_name]  # :291-298
    tg = load_tgmodel(arm, t_s_type, t_s_name)
    robot = make_robot(tg, REST_POSES[arm][t_s_name][t_s_type], t_s_name)
    sensor = SensorDesc(t_s_name, t_s_type, image_size, turn_off_border=False)                   # :121
    mesh = MeshDesc.load("short_edge" if mg else "long_edge")                                    # :220-223
    return cfg, robot, sensor, mesh, modes


class EdgeFollowVecEnv(TactileVecEnv):
    def __init__(self, num_envs, max_steps=250, image_size=(64, 64), env_modes=env_modes_default, physics_dtype="f64",
                 auto_reset=True, device=0, obs_mode="numpy", seed=None, pgs_full_sweeps=False, solver_residual_threshold=0.0, copy_obs=True, contact_mapping="auto", reset_bank="auto", fused_step="auto"):
        cfg, robot, sensor, mesh, modes = build_config(num_envs, max_steps, image_size, env_modes, physics_dtype, auto_reset, device)
        cfg.pgs_full_sweeps = int(bool(pgs_full_sweeps))   # run all solver sweeps instead of leaving at convergence
        cfg.solver_residual_threshold = float(solver_residual_threshold)   # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b): 0 = exit at convergence only, 1e-7 = what PyBullet is believed to run
        cfg.contact_mapping = capi.CONTACT_MAP[contact_mapping]   # \"wave\": every tick a full tick on the env's own wavefront (k_step_arm_wave; measured slower)
        cfg.reset_bank = capi.RESET_BANK[reset_bank]   # "off": every reset on the spot; "sync": the refill is waited for (tests); DESIGN 4.1h
        cfg.fused_step = capi.FUSED_STEP[fused_step]   # "on": the step as one launch (csrc/tg_fused.hip; measured slower); default: k_step -> k_reset -> render
        self.env_modes = modes
        self.min_action, self.max_action = cfg.min_action, cfg.max_action
        super().__init__(cfg, robot, sensor, mesh, observation_mode=modes["observation_mode"], obs_mode=obs_mode, seed=seed, copy_obs=copy_obs,
                         guard_spec={"arm_type": modes["arm_type"], "t_s_core": "no_core",       # edge_follow_env.py:64; the edge: :218-235
                                     "edge": "short_edge" if modes["arm_type"] == "mg400" else "long_edge"},
                         scene_spec={"arm_type": modes["arm_type"], "camera":                     # setup_rgb_obs_camera_params, edge_follow_env.py:176-195
                                     (([-0.20, 0.0, -0.25], 0.85) if modes["arm_type"] == "mg400" else ([0.35, 0.0, -0.25], 0.75)) + (90.0, -35.0, 75.0, 0.1, 100.0)})

    def oracle_obs_host(self):
        """edge_follow_env.py:454-476: [tcp_pos_work(3), tcp_lin_vel_work(3), goal_pos_work(3), edge_ang], float32 [N,10].
        Computed on the host from the device state read-back (the tactile path does not need it)."""
        from .. import hip_ops
        st = self.get_state()
        J, pos, _ = hip_ops.jacobian_tcp(self._robot, st["q"], dtype="f64")
        lin = np.einsum("nij,nj->ni", J[:, :3, :], st["qd"])
        wp = np.array([self._cfg.workframe_pos[k] for k in range(3)])
        r, p, y = (self._cfg.workframe_rpy[k] for k in range(3))
        cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
        Rw = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                       [-sp, cp * sr, cp * cr]])
        ang = st["edge_ang"]
        goal = np.stack([self._cfg.stim_pos[0] + self._cfg.edge_len * np.cos(ang), self._cfg.stim_pos[1] + self._cfg.edge_len * np.sin(ang),
                         np.full_like(ang, self._cfg.stim_pos[2] + self._cfg.edge_height)], axis=1)
        return np.hstack([(pos - wp) @ Rw, lin @ Rw, (goal - wp) @ Rw, ang[:, None]]).astype(np.float32)


class EdgeFollowEnv(SingleTactileEnv):
    """Single-env gym.Env surface; constructor signature as edge_follow_env.py:23-30."""

    vec_cls = EdgeFollowVecEnv
    default_env_modes = env_modes_default

    def __init__(self, max_steps=250, image_size=[64, 64], env_modes=env_modes_default, show_gui=False, show_tactile=False, **kwargs):
        super().__init__(max_steps, image_size, env_modes, show_gui, show_tactile, **kwargs)
