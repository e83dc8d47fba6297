"""surface_follow-v0 (auto-drive variant) and -v1 (goal variant) on the HIP path.

Reference: tactile_gym/rl_envs/exploration/surface_follow/base_surface_env.py (surface generation, goal, rewards),
surface_follow_auto/surface_follow_auto_env.py and surface_follow_goal/surface_follow_goal_env.py (action encoding, dense
reward, extended_feature); -v2 = surface_follow_vert/surface_follow_vert_env.py: the upright surface (`noise_mode`
"vertical_simplex", `movement_mode` "xRz") followed by a `forward` sensor on the MG400 or the UR5.
"""
import math

import numpy as np

from .. import _capi as capi
from .. import pb_math as pbm
from ..robot_model import SensorDesc, load_tgmodel, make_robot
from ..vec_env import SingleTactileEnv, TactileVecEnv

# surface_follow/rest_poses.py (movable joints): every ur5 "standard" entry holds the same pose
REST_POSES = {"ur5": {k: {"standard": [0.16682, -2.18943, -1.65357, -0.86897, 1.57315, 1.74001]} for k in ("tactip", "digit", "digitac")},
              "mg400": {k: {} for k in ("tactip", "digit", "digitac")}}
REST_POSES["ur5"]["tactip"]["forward"] = [0.20199342416011004, -1.8581332389746197, -1.8168154715398577, -1.0385402849835499,
                                          1.569399439236753, -1.3656188934713112]
REST_POSES["ur5"]["digit"]["forward"] = REST_POSES["ur5"]["digitac"]["forward"] = [
    0.19148011767408704, -1.92776038604851, -1.7217555613365743, -1.0625670745823885, 1.568310282843754, -1.3737671809549512]
REST_POSES["mg400"]["tactip"]["forward"] = [0, 0.27678229586424996, 0.6281543378436832, -0.9033290327498503, 0, 0.2767807985667566,
                                            -0.276782284688448, 0.9049031579057567]
REST_POSES["mg400"]["digit"]["forward"] = [0, 0.5905679775553622, 0.3143233272531256, -0.904800272812408, 0, 0.5905665736774282,
                                           -0.5905665736774282, 0.904800272812408]
REST_POSES["mg400"]["digitac"]["forward"] = [0, 0.5212839078833752, 0.4422081884778576, -0.9632925252126955, 0, 0.5212821789748887,
                                             -0.5212821789748887, 0.9632925252126955]

env_modes_default = {  # surface_follow_auto_env.py:6-12
    "movement_mode": "xyzRxRy",
    "control_mode": "TCP_velocity_control",
    "noise_mode": "simplex",
    "observation_mode": "oracle",
    "reward_mode": "dense",
}


def build_config(num_envs, max_steps, image_size, env_modes, physics_dtype="f64", auto_reset=True, device=0, goal_variant=False):
    modes = dict(env_modes)
    for k in ("movement_mode", "control_mode", "noise_mode", "observation_mode", "reward_mode", "arm_type", "tactile_sensor_name"):
        if k not in modes:
            raise KeyError(k)                                                                   # base_surface_env.py:33-63
    arm, t_s_name = modes["arm_type"], modes["tactile_sensor_name"]
    vertical = modes["noise_mode"] == "vertical_simplex"
    if modes["noise_mode"] not in capi.SNOISE:
        raise SystemExit("Incorrect noise mode specified")                                      # :466
    if modes["movement_mode"] not in capi.SMOVE or vertical != (modes["movement_mode"] == "xRz") or (vertical and goal_variant):
        raise SystemExit("Incorrect movement mode specified")                                   # :462-470
    if modes["control_mode"] not in capi.CONTROL:
        if modes["control_mode"] in ("joint_velocity_control",):
            raise NotImplementedError(f"control_mode {modes['control_mode']} is outside the built hot path (SURVEY 8f rank 2)")
        raise SystemExit(f"Incorrect control mode specified: {modes['control_mode']}")
    if arm not in REST_POSES:
        if arm in ("franka_panda", "kuka_iiwa"):
            raise NotImplementedError(f"arm_type {arm} is not built yet for surface_follow")
        raise SystemExit(f"Incorrect arm type specified {arm}")
    if modes["reward_mode"] not in capi.REWARD:
        raise SystemExit("Incorrect reward mode specified")
    t_s_type = "forward" if vertical else "standard"                                            # :60-63
    if t_s_type not in REST_POSES[arm][t_s_name]:
        raise KeyError(t_s_type)                       # e.g. mg400 on the horizontal surface: no rest pose upstream (rest_poses.py)
    cfg = capi.TgConfig()
    cfg.abi_version, cfg.env_kind = capi.ABI_VERSION, capi.ENV_SURFACE_FOLLOW_AUTO
    cfg.num_envs, cfg.max_steps = int(num_envs), int(max_steps)
    cfg.movement_mode, cfg.noise_mode, cfg.reward_mode = capi.SMOVE[modes["movement_mode"]], capi.SNOISE[modes["noise_mode"]], capi.REWARD[modes["reward_mode"]]
    cfg.physics_dtype = capi.PHYSICS[physics_dtype]
    cfg.sim_dt = 1.0 / 240.0                                                                    # :26
    cfg.action_repeat = int(np.floor((1.0 / 10.0) / cfg.sim_dt))                                # :27-28
    cfg.solver_iterations = 150
    cfg.auto_reset, cfg.device = int(auto_reset), int(device)
    cfg.min_action, cfg.max_action = -0.25, 0.25                                                # :160
    cfg.control_mode, cfg.max_blocking_steps = capi.CONTROL[modes["control_mode"]], 10          # :28
    if modes["control_mode"] == "TCP_position_control":
        v, w = 0.001, 1 * (math.pi / 180)                                                       # :167-177 m / rad per step
    else:
        v, w = 0.01, 5.0 * (math.pi / 180)                                                      # :186-194
    lo, hi = [-v, -v, -v, -w, -w, 0.0], [v, v, v, w, w, 0.0]
    height_range, extent = 0.025, 0.15                                                          # :239,245
    wd = (0.33, 0.0, 0.0) if arm == "mg400" else (0.65, 0.0, 0.0)                               # :51-55 well_designed_pos
    if vertical:
        if modes["control_mode"] == "TCP_velocity_control":                                     # :181-191
            lo, hi = [-v, -v, 0.0, 0.0, 0.0, -w], [v, v, 0.0, 0.0, 0.0, w]
        lims = [(-height_range, height_range), (-extent, extent), (0.0, 0.0), (0.0, 0.0), (0.0, 0.0), (-math.pi / 4, math.pi / 4)]   # :93-104
        surface_pos = (wd[0], wd[1], 0.15 + height_range)                                       # :264 (the bins use the same x, y: :249-261)
        wf_rpy = (-math.pi, 0.0, 0.0)                                                           # :87-91
    else:
        lims = [(-extent, extent), (-extent, extent), (-height_range, height_range), (-math.pi / 4, math.pi / 4),
                (-math.pi / 4, math.pi / 4), (0.0, 0.0)]                                        # :112-123
        surface_pos = (wd[0], wd[1], height_range)                                              # :57,266
        wf_rpy = (-math.pi, 0.0, math.pi / 2)                                                   # :108-109
    cfg.surf_vertical = int(vertical)
    for d in range(6):
        cfg.act_lo[d], cfg.act_hi[d] = lo[d], hi[d]
        cfg.tcp_lims[d][0], cfg.tcp_lims[d][1] = lims[d]
    for k in range(3):
        cfg.workframe_pos[k], cfg.workframe_rpy[k], cfg.stim_pos[k] = surface_pos[k], wf_rpy[k], surface_pos[k]
    cfg.termination_dist = 0.01                                                                 # :79
    cfg.embed_dist = {"tactip": 0.0025, "digitac": 0.0015, "digit": 0.0015}[t_s_name]           # :66-75
    cfg.surf_rows, cfg.surf_cols, cfg.surf_center_z = 64, 64, 1                                 # :240-243
    cfg.surf_grid_scale, cfg.surf_height_range, cfg.surf_interp, cfg.surf_xy_extent = 0.006, height_range, 0.05, extent
    cfg.auto_action_scale = {"tactip": 1.0, "digitac": 0.9, "digit": 0.7}[t_s_name]             # surface_follow_auto_env.py:33-41
    cfg.surf_goal_variant = int(bool(goal_variant))
    tg = load_tgmodel(arm, t_s_type, t_s_name)
    robot = make_robot(tg, REST_POSES[arm][t_s_name][t_s_type], t_s_name)
    sensor = SensorDesc(t_s_name, t_s_type, image_size, turn_off_border=False)
    return cfg, robot, sensor, modes


class SurfaceFollowAutoVecEnv(TactileVecEnv):
    def __init__(self, num_envs, max_steps=200, image_size=(64, 64), env_modes=env_modes_default, physics_dtype="f64", auto_reset=True,
                 device=0, obs_mode="numpy", seed=None, pgs_full_sweeps=False, solver_residual_threshold=0.0, copy_obs=True, contact_mapping="auto", reset_bank="auto"):
        cfg, robot, sensor, modes = build_config(num_envs, max_steps, image_size, env_modes, physics_dtype, auto_reset, device)
        cfg.pgs_full_sweeps = int(bool(pgs_full_sweeps))   # run all solver sweeps instead of leaving at convergence
        cfg.solver_residual_threshold = float(solver_residual_threshold)   # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b): 0 = exit at convergence only, 1e-7 = what PyBullet is believed to run
        cfg.contact_mapping = capi.CONTACT_MAP[contact_mapping]   # \"wave\": every tick a full tick on the env's own wavefront (k_step_arm_wave; measured slower)
        cfg.reset_bank = capi.RESET_BANK[reset_bank]   # "off": every reset on the spot; "sync": the refill is waited for (tests); DESIGN 4.1h
        self.env_modes = modes
        self.min_action, self.max_action = cfg.min_action, cfg.max_action
        act_dim = {"yz": 1, "xyz": 1, "yzRx": 2, "xyzRxRy": 3}[modes["movement_mode"]]          # surface_follow_auto_env.py:96-107
        super().__init__(cfg, robot, sensor, None, observation_mode=modes["observation_mode"], obs_mode=obs_mode, seed=seed, copy_obs=copy_obs,
                         act_dim=act_dim, oracle_dim=20,
                         guard_spec={"arm_type": modes["arm_type"], "t_s_core": "fixed"},   # base_surface_env.py:65; the heightfield's collisions are off (:432)
                         scene_spec={"arm_type": modes["arm_type"], "body_rgb": (0, 0, 255), "camera":    # base_surface_env.py:208-232, :431
                                     (([0.16, 0.0, 0.14], 0.45, -2.0, -30.0) if modes["arm_type"] == "mg400" else ([0.65, 0.0, 0.05], 0.4, 90.0, -30.0)) + (75.0, 0.1, 100.0)})

    def oracle_obs_host(self):
        """base_surface_env.py:789-819: TCP pos, orn (quaternion), lin/ang velocity, goal pos (all work frame), the surface height
        under the tip and the surface normal there (work frame); float32 [N, 20]."""
        cfg = self._cfg
        st = self.get_state()
        tp, _, tq, tl, ta = self._tcp_workframe_state(st)
        wf = self._workframe()
        R, Cc, sc = cfg.surf_rows, cfg.surf_cols, cfg.surf_grid_scale
        sp = np.array([cfg.stim_pos[k] for k in range(3)])
        x_bins = np.linspace(sp[0] - (R / 2) * sc, sp[0] + (R / 2) * sc, R)                     # :268-282
        y_bins = np.linspace(sp[1] - (Cc / 2) * sc, sp[1] + (Cc / 2) * sc, Cc)
        ti = np.minimum(np.digitize(st["tcp_pos"][:, 1], y_bins), Cc - 1)                       # xy_to_surface_idx :284-300
        tj = np.minimum(np.digitize(st["tcp_pos"][:, 0], x_bins), R - 1)
        H = st["heights"]
        idx = np.arange(self.num_envs)
        gy, gx = np.gradient(H, sc, axis=(1, 2))                                                # :502-503
        nrm = np.stack([-gx[idx, ti, tj], -gy[idx, ti, tj], np.ones(self.num_envs)], axis=1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        gp, _ = wf.pose(st["goal_pos"], np.zeros((self.num_envs, 3)))
        return np.hstack([tp, tq, tl, ta, gp, (H[idx, ti, tj] + sp[2])[:, None], wf.vec(nrm)]).astype(np.float32)


class SurfaceFollowGoalVecEnv(SurfaceFollowAutoVecEnv):
    """surface_follow-v1: the agent drives every dimension; `tactile_and_feature` adds [tcp_pos, goal_pos] in the work frame."""

    def __init__(self, num_envs, max_steps=200, image_size=(64, 64), env_modes=env_modes_default, physics_dtype="f64", auto_reset=True,
                 device=0, obs_mode="numpy", seed=None, pgs_full_sweeps=False, solver_residual_threshold=0.0, copy_obs=True, contact_mapping="auto", reset_bank="auto"):
        cfg, robot, sensor, modes = build_config(num_envs, max_steps, image_size, env_modes, physics_dtype, auto_reset, device,
                                                 goal_variant=True)
        cfg.pgs_full_sweeps = int(bool(pgs_full_sweeps))
        cfg.solver_residual_threshold = float(solver_residual_threshold)   # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b): 0 = exit at convergence only, 1e-7 = what PyBullet is believed to run
        cfg.contact_mapping = capi.CONTACT_MAP[contact_mapping]   # \"wave\": every tick a full tick on the env's own wavefront (k_step_arm_wave; measured slower)
        cfg.reset_bank = capi.RESET_BANK[reset_bank]   # "off": every reset on the spot; "sync": the refill is waited for (tests); DESIGN 4.1h
        self.env_modes = modes
        self.min_action, self.max_action = cfg.min_action, cfg.max_action
        act_dim = {"yz": 2, "xyz": 3, "yzRx": 3, "xyzRxRy": 5}[modes["movement_mode"]]          # surface_follow_goal_env.py:112-123
        TactileVecEnv.__init__(self, cfg, robot, sensor, None, observation_mode=modes["observation_mode"], obs_mode=obs_mode, seed=seed, copy_obs=copy_obs,
                               act_dim=act_dim, oracle_dim=20, feature_dim=6,
                               guard_spec={"arm_type": modes["arm_type"], "t_s_core": "fixed"},   # base_surface_env.py:65; the heightfield's collisions are off (:432)
                               scene_spec={"arm_type": modes["arm_type"], "body_rgb": (0, 0, 255), "camera":    # base_surface_env.py:208-232, :431
                                     (([0.16, 0.0, 0.14], 0.45, -2.0, -30.0) if modes["arm_type"] == "mg400" else ([0.65, 0.0, 0.05], 0.4, 90.0, -30.0)) + (75.0, 0.1, 100.0)})

    def feature_numpy(self, terminal=False):
        """get_extended_feature_array (surface_follow_goal_env.py:92-110): TCP and goal position in the work frame, float32 [N, 6], written
        by the step / reset kernels (device rows are 12 floats wide); terminal: the copy taken at the end of the step, before the auto-reset."""
        import ctypes as C
        buf = np.zeros((self.num_envs, 12), dtype=np.float32)
        capi.check(self._L.tg_copy_obs_feature(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_float)), int(terminal)))
        return np.ascontiguousarray(buf[:, :6])

    def feature_torch(self, terminal=False):
        return TactileVecEnv.feature_torch(self, terminal)[:, :6]

    def feature_host(self):
        """The same vector from a state read-back (cross-check of the device route)."""
        st = self.get_state()
        tp, _, _, _, _ = self._tcp_workframe_state(st)
        gp, _ = self._workframe().pose(st["goal_pos"], np.zeros((self.num_envs, 3)))
        return np.hstack([tp, gp]).astype(np.float32)


env_modes_default_vert = {  # surface_follow_vert_env.py:6-12
    "movement_mode": "xRz",
    "control_mode": "TCP_velocity_control",
    "noise_mode": "vertical_simplex",
    "observation_mode": "oracle",
    "reward_mode": "dense",
}


class SurfaceFollowVertVecEnv(SurfaceFollowGoalVecEnv):
    """surface_follow-v2: upright surface, y auto-driven, the agent controls x (approach) and Rz; `tactile_and_feature` adds
    [tcp_pos, goal_pos] in the work frame (surface_follow_vert_env.py:83-100)."""

    def __init__(self, num_envs, max_steps=200, image_size=(64, 64), env_modes=env_modes_default_vert, physics_dtype="f64", auto_reset=True,
                 device=0, obs_mode="numpy", seed=None, pgs_full_sweeps=False, solver_residual_threshold=0.0, copy_obs=True, contact_mapping="auto", reset_bank="auto"):
        cfg, robot, sensor, modes = build_config(num_envs, max_steps, image_size, env_modes, physics_dtype, auto_reset, device)
        cfg.pgs_full_sweeps = int(bool(pgs_full_sweeps))
        cfg.solver_residual_threshold = float(solver_residual_threshold)   # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b): 0 = exit at convergence only, 1e-7 = what PyBullet is believed to run
        cfg.contact_mapping = capi.CONTACT_MAP[contact_mapping]   # \"wave\": every tick a full tick on the env's own wavefront (k_step_arm_wave; measured slower)
        cfg.reset_bank = capi.RESET_BANK[reset_bank]   # "off": every reset on the spot; "sync": the refill is waited for (tests); DESIGN 4.1h
        self.env_modes = modes
        self.min_action, self.max_action = cfg.min_action, cfg.max_action
        TactileVecEnv.__init__(self, cfg, robot, sensor, None, observation_mode=modes["observation_mode"], obs_mode=obs_mode, seed=seed, copy_obs=copy_obs,
                               act_dim=2, oracle_dim=20, feature_dim=6,                          # get_act_dim :102-113
                               guard_spec={"arm_type": modes["arm_type"], "t_s_core": "fixed"},   # base_surface_env.py:65; the heightfield's collisions are off (:432)
                               scene_spec={"arm_type": modes["arm_type"], "body_rgb": (0, 0, 255), "camera":    # base_surface_env.py:208-232, :431
                                     (([0.16, 0.0, 0.14], 0.45, -2.0, -30.0) if modes["arm_type"] == "mg400" else ([0.65, 0.0, 0.05], 0.4, 90.0, -30.0)) + (75.0, 0.1, 100.0)})

    def oracle_obs_host(self):
        """base_surface_env.py:789-819 on the flipped surface_array / normals (:486-516)."""
        cfg = self._cfg
        st = self.get_state()
        tp, _, tq, tl, ta = self._tcp_workframe_state(st)
        wf = self._workframe()
        R, Cc, sc = cfg.surf_rows, cfg.surf_cols, cfg.surf_grid_scale
        sp = np.array([cfg.stim_pos[k] for k in range(3)])
        x_bins = np.linspace(sp[0] - (R / 2) * sc, sp[0] + (R / 2) * sc, R)
        y_bins = np.linspace(sp[1] - (Cc / 2) * sc, sp[1] + (Cc / 2) * sc, Cc)
        ti = np.minimum(np.digitize(st["tcp_pos"][:, 1], y_bins), Cc - 1)
        tj = np.minimum(np.digitize(st["tcp_pos"][:, 0], x_bins), R - 1)
        H = st["heights"]
        idx = np.arange(self.num_envs)
        gy, gx = np.gradient(H, sc, axis=(1, 2))
        nrm = np.stack([-gx[idx, ti, tj], -gy[idx, ti, tj], np.ones(self.num_envs)], axis=1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        flip = pbm.mat_from_quat(pbm.quat_from_euler(np.array([0.0, -math.pi / 2, 0.0])))
        nrm = nrm @ flip.T
        gp, _ = wf.pose(st["goal_pos"], np.zeros((self.num_envs, 3)))
        surf_z = sp[2] + (x_bins[tj] - sp[0])                                                   # z of the flipped surface point
        return np.hstack([tp, tq, tl, ta, gp, surf_z[:, None], wf.vec(nrm)]).astype(np.float32)


class SurfaceFollowVertEnv(SingleTactileEnv):
    """Single-env gym.Env surface; constructor signature as surface_follow_vert_env.py:15-27."""

    vec_cls = SurfaceFollowVertVecEnv
    default_env_modes = env_modes_default_vert

    def __init__(self, max_steps=200, image_size=[64, 64], env_modes=env_modes_default_vert, show_gui=False, show_tactile=False, **kwargs):
        super().__init__(max_steps, image_size, env_modes, show_gui, show_tactile, **kwargs)



class SurfaceFollowGoalEnv(SingleTactileEnv):
    """Single-env gym.Env surface; constructor signature as surface_follow_goal_env.py:15-25."""

    vec_cls = SurfaceFollowGoalVecEnv
    default_env_modes = env_modes_default

    def __init__(self, max_steps=200, image_size=[64, 64], env_modes=env_modes_default, show_gui=False, show_tactile=False, **kwargs):
        super().__init__(max_steps, image_size, env_modes, show_gui, show_tactile, **kwargs)



class SurfaceFollowAutoEnv(SingleTactileEnv):
    """Single-env gym.Env surface; constructor signature as surface_follow_auto_env.py:16-25."""

    vec_cls = SurfaceFollowAutoVecEnv
    default_env_modes = env_modes_default

    def __init__(self, max_steps=200, image_size=[64, 64], env_modes=env_modes_default, show_gui=False, show_tactile=False, **kwargs):
        super().__init__(max_steps, image_size, env_modes, show_gui, show_tactile, **kwargs)
