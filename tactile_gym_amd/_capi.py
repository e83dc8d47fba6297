"""ctypes binding of libtactile_gym_hip.so (C ABI declared in include/tactile_gym_hip.h).

The HIP library is the product path; there is no CPU fallback.  Importing this module is cheap, loading the library
(`lib()`) fails loudly with build instructions if the shared object is missing.
"""
import ctypes as C
import os

MAX_DOF = 8
MAX_BODIES_PER_LINK = 4
ABI_VERSION = 14
MAX_TRAJ_POINTS = 16

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TG_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libtactile_gym_hip.so")   # TG_HIP_LIBRARY: another build of the same sources (A/B measurements)

ENV_EDGE_FOLLOW, ENV_SURFACE_FOLLOW_AUTO, ENV_OBJECT_BALANCE, ENV_OBJECT_PUSH, ENV_OBJECT_ROLL = 0, 1, 2, 3, 4
PMOVE = {"y": 0, "yRz": 1, "xyRz": 2, "TyRz": 3, "TxTyRz": 4}
TRAJ = {"simplex": 0, "straight": 1}
BMOVE = {"xy": 0, "xyz": 1, "RxRy": 2, "xyRxRy": 3}
CONTROL = {"TCP_velocity_control": 0, "TCP_position_control": 1}
SNOISE = {"simplex": 0, "none": 1, "random": 2, "vertical_simplex": 3}
SMOVE = {"yz": 0, "xyz": 1, "yzRx": 2, "xyzRxRy": 3, "xRz": 4}
MOVE = {"xy": 0, "xyz": 1, "xyRz": 2, "xyzRz": 3}
NOISE = {"fixed_height": 0, "rand_height": 1}
REWARD = {"dense": 0, "sparse": 1}
PHYSICS = {"f64": 0, "f32": 1}
CONTACT_MAP = {"auto": 0, "lane": 1, "wave": 2}
RESET_BANK = {"auto": 0, "off": 1, "sync": 2, "on": 3}
FUSED_STEP = {"auto": 0, "off": 1, "on": 2}
NARROWPHASE = {"closed_form": 0, "gjk_manifold": 1, "gjk_single": 2}
BALANCE_OBJECT = {"pole": 0, "ball_on_plate": 1, "spinning_plate": 2}
MOTOR_OFF, MOTOR_VELOCITY, MOTOR_POSITION = 0, 1, 2

_d3 = C.c_double * 3
_d9 = C.c_double * 9


class TgRobot(C.Structure):
    _fields_ = [
        ("ndof", C.c_int32), ("topology", C.c_int32),
        ("joint_pos", _d3 * MAX_DOF), ("joint_rot", _d9 * MAX_DOF), ("joint_axis", _d3 * MAX_DOF),
        ("body_mass", (C.c_double * MAX_BODIES_PER_LINK) * MAX_DOF),
        ("body_com", (_d3 * MAX_BODIES_PER_LINK) * MAX_DOF),
        ("body_rot", (_d9 * MAX_BODIES_PER_LINK) * MAX_DOF),
        ("body_inertia", (_d3 * MAX_BODIES_PER_LINK) * MAX_DOF),
        ("tcp_link", C.c_int32), ("tcp_pos", _d3), ("tcp_rot", _d9),
        ("sensor_link", C.c_int32), ("sensor_pos", _d3), ("sensor_rot", _d9),
        ("gravity", _d3), ("linear_damping", C.c_double), ("angular_damping", C.c_double), ("joint_damping", C.c_double),
        ("max_force", C.c_double), ("pos_gain", C.c_double), ("vel_gain", C.c_double),
        ("rest_q", C.c_double * MAX_DOF),
    ]


class TgSensor(C.Structure):
    _fields_ = [
        ("image_h", C.c_int32), ("image_w", C.c_int32), ("cam_pos", _d3), ("cam_rpy", _d3),
        ("fov_deg", C.c_double), ("near_plane", C.c_double), ("far_plane", C.c_double), ("turn_off_border", C.c_int32),
        ("nodef_dep", C.POINTER(C.c_float)), ("nodef_gray", C.POINTER(C.c_float)), ("border_mask", C.POINTER(C.c_uint8)),
    ]


class TgMesh(C.Structure):
    _fields_ = [("n_verts", C.c_int32), ("n_tris", C.c_int32), ("verts", C.POINTER(C.c_float)), ("tris", C.POINTER(C.c_int32))]


class TgScene(C.Structure):
    _fields_ = [("image_h", C.c_int32), ("image_w", C.c_int32), ("n_verts", C.c_int32), ("n_tris", C.c_int32),
                ("verts", C.POINTER(C.c_float)), ("tris", C.POINTER(C.c_int32)), ("tri_frame", C.POINTER(C.c_uint8)),
                ("tri_rgb", C.POINTER(C.c_uint8)), ("cam_target", C.c_double * 3), ("cam_dist", C.c_double), ("cam_yaw_deg", C.c_double),
                ("cam_pitch_deg", C.c_double), ("fov_deg", C.c_double), ("near_plane", C.c_double), ("far_plane", C.c_double),
                ("light_dir", C.c_double * 3), ("background", C.c_uint8 * 3), ("body_rgb", C.c_uint8 * 3), ("body_heightfield", C.c_int32),
                ("every_step", C.c_int32)]


class TgConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("env_kind", C.c_int32), ("num_envs", C.c_int32), ("max_steps", C.c_int32),
        ("movement_mode", C.c_int32), ("noise_mode", C.c_int32), ("reward_mode", C.c_int32), ("physics_dtype", C.c_int32),
        ("action_repeat", C.c_int32), ("solver_iterations", C.c_int32), ("auto_reset", C.c_int32), ("device", C.c_int32),
        ("sim_dt", C.c_double), ("min_action", C.c_double), ("max_action", C.c_double),
        ("act_lo", C.c_double * 6), ("act_hi", C.c_double * 6), ("tcp_lims", (C.c_double * 2) * 6),
        ("workframe_pos", _d3), ("workframe_rpy", _d3), ("stim_pos", _d3),
        ("edge_height", C.c_double), ("edge_len", C.c_double), ("termination_dist", C.c_double),
        ("embed_dist", C.c_double), ("embed_lo", C.c_double), ("embed_hi", C.c_double),
        ("surf_rows", C.c_int32), ("surf_cols", C.c_int32), ("surf_center_z", C.c_int32), ("pgs_full_sweeps", C.c_int32),
        ("surf_grid_scale", C.c_double), ("surf_height_range", C.c_double), ("surf_interp", C.c_double),
        ("surf_xy_extent", C.c_double), ("auto_action_scale", C.c_double),
        ("rand_gravity", C.c_int32), ("rand_embed", C.c_int32),
        ("gravity_lo", C.c_double), ("gravity_hi", C.c_double), ("gravity_default", C.c_double),
        ("obj_mass", C.c_double), ("obj_com", _d3), ("obj_inertia", _d9), ("obj_root_inertial_pos", _d3),
        ("obj_base_width", C.c_double), ("obj_base_height", C.c_double), ("obj_init_rpy", _d3), ("ext_force", _d3),
        ("term_deg", C.c_double), ("term_pos", C.c_double), ("p2p_erp", C.c_double), ("p2p_max_impulse", C.c_double),
        ("traj_type", C.c_int32), ("traj_n_points", C.c_int32), ("rand_init_orn", C.c_int32), ("rand_obj_mass", C.c_int32),
        ("tip_link", C.c_int32), ("n_tip_verts", C.c_int32), ("cone_friction", C.c_int32), ("surf_goal_variant", C.c_int32),
        ("tip_verts", C.POINTER(C.c_double)),
        ("obj_half", _d3), ("obj_init_pos", _d3), ("table_z", C.c_double), ("mu_table", C.c_double), ("mu_tip", C.c_double),
        ("margin_cube", C.c_double), ("margin_tip", C.c_double), ("contact_breaking", C.c_double), ("contact_erp", C.c_double),
        ("tip_stiffness", C.c_double), ("tip_damping", C.c_double), ("obj_lin_damp", C.c_double), ("obj_ang_damp", C.c_double),
        ("traj_spacing", C.c_double), ("traj_max_perturb", C.c_double), ("traj_init_offset", C.c_double),
        ("mass_lo", C.c_double), ("mass_hi", C.c_double), ("init_orn_range", C.c_double), ("traj_ang_range", C.c_double),
        ("control_mode", C.c_int32), ("max_blocking_steps", C.c_int32), ("reset_goal_id", C.c_int32), ("surf_vertical", C.c_int32),
        ("roll_rand_init_pos", C.c_int32), ("roll_rand_size", C.c_int32), ("roll_rand_embed", C.c_int32),
        ("roll_radius", C.c_double), ("roll_init_range", C.c_double), ("roll_goal_lo", C.c_double), ("roll_goal_hi", C.c_double),
        ("tip_cyl_pos", _d3), ("tip_cyl_rot", _d9), ("tip_cyl_half_len", C.c_double), ("tip_cyl_radius", C.c_double),
        ("contact_mapping", C.c_int32), ("reset_bank", C.c_int32), ("narrowphase", C.c_int32),
        ("balance_object", C.c_int32), ("ball_radius", C.c_double), ("ball_mass", C.c_double), ("ball_mu", C.c_double), ("plate_radius", C.c_double),
        ("fused_step", C.c_int32),
        ("solver_residual_threshold", C.c_double),
        ("spin_dish_mass", C.c_double), ("spin_dish_com", C.c_double * 3), ("spin_dish_inertia", C.c_double * 9),
        ("spin_buffer_height", C.c_double), ("spin_hull_margin", C.c_double), ("spin_mu", C.c_double),
        ("spin_n_dish", C.c_int32), ("spin_n_spool", C.c_int32),
        ("spin_dish_hull", C.POINTER(C.c_double)), ("spin_spool_hull", C.POINTER(C.c_double)),
    ]


BP_NONE, BP_LINK, BP_WORLD, BP_EDGE, BP_BODY, BP_SPHERE, BP_BALL = range(7)     # tg_bp_box.src
BP_SLOTS = 22


class TgBpBox(C.Structure):
    _fields_ = [("center", _d3), ("rot", _d9), ("half", _d3), ("src", C.c_int32), ("link", C.c_int32), ("body", C.c_int32), ("is_static", C.c_int32),
                ("hull_off", C.c_int32), ("hull_n", C.c_int32), ("expected", C.c_uint32), ("conj", C.c_int32)]


class TgBroadphase(C.Structure):
    _fields_ = [("box", TgBpBox * BP_SLOTS), ("margin", C.c_double), ("hull_margin", C.c_double), ("sphere_half", C.c_double), ("ball_radius", C.c_double),
                ("n_hull_verts", C.c_int32), ("every_step", C.c_int32), ("hull_verts", C.POINTER(C.c_double))]


class TgStateView(C.Structure):
    _fields_ = [
        ("q", C.POINTER(C.c_double)), ("qd", C.POINTER(C.c_double)), ("qd_target", C.POINTER(C.c_double)),
        ("tcp_pos", C.POINTER(C.c_double)), ("tcp_rpy", C.POINTER(C.c_double)), ("edge_ang", C.POINTER(C.c_double)),
        ("embed_dist", C.POINTER(C.c_double)), ("stim_xform", C.POINTER(C.c_float)), ("step_count", C.POINTER(C.c_int32)),
        ("reset_ticks", C.POINTER(C.c_int32)), ("rng_state", C.POINTER(C.c_uint64)),
        ("goal_pos", C.POINTER(C.c_double)), ("direction", C.POINTER(C.c_double)), ("heights", C.POINTER(C.c_double)),
        ("surf_zoff", C.POINTER(C.c_float)),
        ("body_pos", C.POINTER(C.c_double)), ("body_rot", C.POINTER(C.c_double)), ("body_linvel", C.POINTER(C.c_double)),
        ("body_angvel", C.POINTER(C.c_double)), ("gravity_z", C.POINTER(C.c_double)),
        ("traj", C.POINTER(C.c_double)), ("goal_id", C.POINTER(C.c_int32)), ("obj_mass", C.POINTER(C.c_double)),
        ("contact_count", C.POINTER(C.c_int32)), ("contact_ids", C.POINTER(C.c_int32)),
        ("ball_pos", C.POINTER(C.c_double)), ("ball_linvel", C.POINTER(C.c_double)), ("ball_angvel", C.POINTER(C.c_double)),
        ("ball_impulse", C.POINTER(C.c_double)),
        ("dish_state", C.POINTER(C.c_double)),
        ("broadphase_pairs", C.POINTER(C.c_int32)), ("broadphase_hits", C.POINTER(C.c_int32)), ("broadphase_mask", C.POINTER(C.c_int32)),
        ("solver_sweeps", C.POINTER(C.c_int32)),
    ]


# name -> (restype, argtypes); every symbol include/tactile_gym_hip.h declares
_dp, _fp, _u8p, _vpp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_void_p)
_ctx = C.c_void_p
SYMBOLS = {
    "tg_last_error": (C.c_char_p, []),
    "tg_abi_version": (C.c_int, []),
    "tg_create": (C.c_int, [C.POINTER(TgConfig), C.POINTER(TgRobot), C.POINTER(TgSensor), C.POINTER(TgMesh), C.POINTER(_ctx)]),
    "tg_destroy": (C.c_int, [_ctx]),
    "tg_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "tg_seed": (C.c_int, [_ctx, C.POINTER(C.c_uint64), C.c_int32]),
    "tg_reset": (C.c_int, [_ctx, _u8p]),
    "tg_step": (C.c_int, [_ctx, C.c_void_p, C.c_int32]),
    "tg_sync": (C.c_int, [_ctx]),
    "tg_get_obs_tactile": (C.c_int, [_ctx, _vpp]),
    "tg_get_terminal_obs": (C.c_int, [_ctx, _vpp]),
    "tg_get_reward_done_dev": (C.c_int, [_ctx, _vpp, _vpp]),
    "tg_get_packed_outputs": (C.c_int, [_ctx, _vpp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tg_get_packed_feature": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tg_sample_actions": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_void_p]),
    "tg_step_random": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_int32]),
    "tg_get_actions": (C.c_int, [_ctx, _vpp]),
    "tg_get_interior_count": (C.c_int, [_ctx, C.POINTER(C.c_int32)]),
    "tg_get_bank_stats": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tg_get_step_mode": (C.c_int, [_ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tg_set_obs_targets": (C.c_int, [_ctx, C.c_int32, C.POINTER(C.c_void_p)]),
    "tg_select_obs_target": (C.c_int, [_ctx, C.c_int32]),
    "tg_pack_interior": (C.c_int, [_ctx, C.c_void_p]),
    "tg_unpack_interior": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_void_p]),
    "tg_get_episode_stats": (C.c_int, [_ctx, _vpp, _vpp]),
    "tg_copy_episode_stats": (C.c_int, [_ctx, _fp, C.POINTER(C.c_int32)]),
    "tg_get_tile_template": (C.c_int, [_ctx, _vpp]),
    "tg_tiles_capacity": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "tg_pack_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "tg_unpack_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "tg_unpack_tiles_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "tg_ipc_alloc": (C.c_int, [C.c_int64, _vpp, _u8p]),
    "tg_ipc_alloc_was_uncached": (C.c_int, []),
    "tg_ipc_free": (C.c_int, [C.c_void_p]),
    "tg_ipc_open": (C.c_int, [_u8p, _vpp]),
    "tg_ipc_close": (C.c_int, [C.c_void_p]),
    "tg_copy_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "tg_copy_bytes2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]),
    "tg_copy_bytes2_flag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32]),
    "tg_flag_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32]),
    "tg_flag_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32]),
    "tg_get_obs_oracle": (C.c_int, [_ctx, _vpp, C.POINTER(C.c_int32)]),
    "tg_copy_obs_oracle": (C.c_int, [_ctx, _fp]),
    "tg_enable_oracle_obs": (C.c_int, [_ctx]),
    "tg_get_obs_oracle_terminal": (C.c_int, [_ctx, _vpp]),
    "tg_copy_obs_oracle_terminal": (C.c_int, [_ctx, _fp]),
    "tg_set_scene": (C.c_int, [_ctx, C.POINTER(TgScene)]),
    "tg_render_scene": (C.c_int, [_ctx]),
    "tg_done_rows_bytes": (C.c_int, [_ctx, C.c_int32, C.POINTER(C.c_int64)]),
    "tg_pack_done_rows": (C.c_int, [_ctx, C.c_void_p, C.c_int32]),
    "tg_set_broadphase": (C.c_int, [_ctx, C.POINTER(TgBroadphase)]),
    "tg_check_broadphase": (C.c_int, [_ctx]),
    "tg_get_broadphase_totals": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tg_get_obs_visual": (C.c_int, [_ctx, _vpp, C.c_int32]),
    "tg_copy_obs_visual": (C.c_int, [_ctx, _u8p, C.c_int32]),
    "tg_get_obs_feature": (C.c_int, [_ctx, _vpp, C.POINTER(C.c_int32), C.c_int32]),
    "tg_get_reward_done": (C.c_int, [_ctx, _fp, _u8p]),
    "tg_copy_obs_tactile": (C.c_int, [_ctx, _u8p, C.c_int32]),
    "tg_copy_obs_rows": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _u8p]),
    "tg_copy_obs_feature": (C.c_int, [_ctx, _fp, C.c_int32]),
    "tg_get_state": (C.c_int, [_ctx, C.POINTER(TgStateView)]),
    "tg_set_joint_state": (C.c_int, [_ctx, _dp, _dp]),
    "tg_profile_enable": (C.c_int, [_ctx, C.c_int32]),
    "tg_profile_get": (C.c_int, [_ctx, C.c_int32, _dp, C.POINTER(C.c_int64)]),
    "tg_inverse_dynamics": (C.c_int, [C.POINTER(TgRobot), C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]),
    "tg_mass_matrix": (C.c_int, [C.POINTER(TgRobot), C.c_int32, C.c_int32, _dp, _dp]),
    "tg_jacobian_tcp": (C.c_int, [C.POINTER(TgRobot), C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]),
    "tg_sim_ticks": (C.c_int, [C.POINTER(TgRobot), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, _dp, _dp,
                               C.c_double, _dp, _dp]),
    "tg_inverse_kinematics": (C.c_int, [C.POINTER(TgRobot), C.c_int32, C.c_int32, _dp, _dp, _dp, C.c_int32, C.c_double, _dp,
                                        C.POINTER(C.c_int32)]),
    "tg_render_tactile": (C.c_int, [C.POINTER(TgSensor), C.POINTER(TgMesh), C.c_int32, _fp, _u8p]),
    "tg_render_tactile_heightfield": (C.c_int, [C.POINTER(TgSensor), C.c_int32, C.c_int32, C.c_double, C.c_int32, _dp, _fp, _fp, _u8p]),
    "tg_gen_heightfield": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_double, C.c_double, _dp, _fp]),
}

_lib = None


class TactileGymHipError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64; a process must not end up with two HIP runtimes (torch then
    reports "No HIP GPUs are available").  Load torch's copy first (without importing torch) so this library binds to
    the same runtime whichever is imported first; with no torch installed the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.isfile(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


# libtactile_gym_hip_test.so (include/tactile_gym_hip_test.h): device self-tests, test infrastructure - tests/ are the only callers
TEST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libtactile_gym_hip_test.so")
TEST_SYMBOLS = {
    "tg_selftest_narrowphase": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tg_selftest_narrowphase_hulls": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tg_selftest_division": (C.c_int, [C.c_int64, C.c_uint64, C.POINTER(C.c_int64)]),
    "tg_selftest_penetration_division": (C.c_int, [C.POINTER(C.c_int64)]),
    "tg_selftest_edge_exclusion": (C.c_int, [C.c_int64, C.c_uint64, C.POINTER(C.c_int64)]),
}
_test_lib = None


def test_lib():
    """Load libtactile_gym_hip_test.so (built next to the product library by csrc/build.sh)."""
    global _test_lib
    if _test_lib is None:
        if not os.path.isfile(TEST_LIB_PATH):
            raise TactileGymHipError(f"{TEST_LIB_PATH} is missing: run tactile_gym_amd/csrc/build.sh")
        _share_torch_hip_runtime()
        L = C.CDLL(TEST_LIB_PATH)
        for name, (res, args) in TEST_SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _test_lib = L
    return _test_lib


def lib():
    """Load libtactile_gym_hip.so; raise with build instructions if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise TactileGymHipError(
                f"{LIB_PATH} is missing.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                f"`tactile_gym_amd/csrc/build.sh` (hipcc, --offload-arch=gfx950).  There is no CPU fallback for the env step.")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if L.tg_abi_version() != ABI_VERSION:
            raise TactileGymHipError("libtactile_gym_hip.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def step_graphs_enabled():
    """Whether the library replays a step as a captured graph (TG_STEP_GRAPH=1) or enqueues its launches on the stream (the default since round
    6: a graph launch costs ~6.6 us before its first kernel on this stack, a stream launch ~2 us; csrc/tg_api.hip: step_as_graph).  Mirrors the
    library's own reading of the variable: callers that must keep other threads' event polls away from a stream capture (parallel.py) ask here."""
    v = os.environ.get("TG_STEP_GRAPH")
    try:
        return v is not None and int(v) != 0
    except ValueError:
        return False


def check(rc):
    if rc != 0:
        raise TactileGymHipError(lib().tg_last_error().decode("utf-8", "replace") or f"libtactile_gym_hip error {rc}")
