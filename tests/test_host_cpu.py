"""CPU tests of the host layer: registry, constructor-kwarg handling, URDF flattening, and that the C-ABI library
loads and exports every symbol include/tactile_gym_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_has_reference_ids():
    import tactile_gym_amd as tg
    ids = set(tg.registered_ids())
    # tactile_gym/rl_envs/__init__.py:3-41
    assert {"edge_follow-v0", "edge_follow_aotu-v0", "surface_follow-v0", "surface_follow-v1", "surface_follow-v2", "object_roll-v0",
            "object_push-v0", "object_balance-v0"} <= ids
    with pytest.raises(KeyError):
        tg.make("no_such_env-v0")
    with pytest.raises(ImportError):          # the upstream id points at a class that does not exist either
        tg.make("edge_follow_aotu-v0")


def test_header_symbols_match_binding_and_library():
    from tactile_gym_amd import _capi
    header = open(os.path.join(ROOT, "include", "tactile_gym_hip.h")).read()
    declared = set(re.findall(r"\b(tg_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    L = _capi.lib()                           # raises if the .so is missing: build() must have run
    for name in declared:
        assert hasattr(L, name)
    assert L.tg_abi_version() == _capi.ABI_VERSION
    assert not any("selftest" in name for name in declared)            # test hooks are not part of the product ABI ...
    test_header = open(os.path.join(ROOT, "include", "tactile_gym_hip_test.h")).read()
    test_declared = set(re.findall(r"\b(tg_[a-z_0-9]+)\s*\(", test_header))
    assert test_declared == set(_capi.TEST_SYMBOLS), test_declared ^ set(_capi.TEST_SYMBOLS)
    T = _capi.test_lib()                       # ... they live in libtactile_gym_hip_test.so, which the product library does not export
    for name in test_declared:
        assert hasattr(T, name) and not hasattr(L, name)
    assert ctypes.sizeof(_capi.TgRobot) == 4 * 2 + 8 * (8 * 3 + 8 * 9 + 8 * 3 + 8 * 4 * (1 + 3 + 9 + 3)) + (8 + 8 * 12) * 2 + 8 * (3 + 6 + 8)


def test_no_gpu_fails_loudly(edge_modes):
    """Without a GPU the product path must refuse to run rather than fall back to anything."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import tactile_gym_amd as tg
    from tactile_gym_amd._capi import TactileGymHipError
    with pytest.raises(TactileGymHipError, match="no HIP device"):
        tg.make_vec("edge_follow-v0", num_envs=2, max_steps=10, image_size=[128, 128], env_modes=edge_modes)


def test_constructor_kwarg_errors(edge_modes):
    from tactile_gym_amd.rl_envs.edge_follow import build_config
    bad = dict(edge_modes)
    bad.pop("tactile_sensor_name")
    with pytest.raises(KeyError):             # the reference raises KeyError for the same omission (edge_follow_env.py:59)
        build_config(4, 200, [128, 128], bad)
    with pytest.raises(SystemExit):           # robot.py:174 sys.exit on an unknown control mode
        build_config(4, 200, [128, 128], dict(edge_modes, control_mode="bogus"))
    with pytest.raises(SystemExit):           # robot.py:65
        build_config(4, 200, [128, 128], dict(edge_modes, arm_type="bogus"))
    with pytest.raises(FileNotFoundError):    # no reference images for that size, as np.load would fail upstream
        build_config(4, 200, [100, 100], edge_modes)


def test_config_matches_reference_constants(edge_modes):
    from tactile_gym_amd.rl_envs.edge_follow import build_config
    cfg, robot, sensor, mesh, _ = build_config(1024, 200, [128, 128], edge_modes)
    assert cfg.action_repeat == 24 and cfg.solver_iterations == 150 and abs(cfg.sim_dt - 1 / 240) < 1e-18
    assert (cfg.min_action, cfg.max_action) == (-0.25, 0.25)
    assert list(cfg.act_hi)[:3] == [0.01] * 3 and abs(cfg.act_hi[5] - np.deg2rad(5)) < 1e-15 and cfg.act_hi[3] == 0
    assert robot.ndof == 6 and robot.topology == 0 and robot.tcp_link == 5 and robot.sensor_link == 5
    assert mesh.tris.shape == (12, 3) and sensor.nodef_dep.shape == (128, 128)
    assert (cfg.embed_lo, cfg.embed_hi) == (0.0015, 0.0065)


def test_urdf_compile_roundtrip(tmp_path):
    """compile_urdf on a hand-written 2-link URDF: tree, merged fixed link, inertial-frame convention."""
    from tactile_gym_amd.urdf_compile import TGModel, compile_urdf
    urdf = tmp_path / "arm.urdf"
    urdf.write_text("""<robot name="t">
      <link name="world"/>
      <link name="a"><inertial><mass value="2"/><origin xyz="0 0 0.1" rpy="0 0 0"/><inertia ixx="1" iyy="2" izz="3" ixy="0" ixz="0" iyz="0"/></inertial></link>
      <link name="b"><inertial><mass value="1"/><origin xyz="0.2 0 0" rpy="0 0 1.57"/><inertia ixx="0.1" iyy="0.2" izz="0.3" ixy="0" ixz="0" iyz="0"/></inertial></link>
      <link name="tool"><inertial><mass value="0.5"/><origin xyz="0 0 0"/><inertia ixx="0.01" iyy="0.01" izz="0.01" ixy="0" ixz="0" iyz="0"/></inertial></link>
      <joint name="j0" type="revolute"><parent link="world"/><child link="a"/><origin xyz="0 0 1" rpy="0 0 0"/><axis xyz="0 0 1"/></joint>
      <joint name="j1" type="revolute"><parent link="a"/><child link="b"/><origin xyz="0 0 0.5" rpy="0 1.57 0"/><axis xyz="0 1 0"/></joint>
      <joint name="jt" type="fixed"><parent link="b"/><child link="tool"/><origin xyz="0.4 0 0" rpy="0 0 0"/></joint>
    </robot>""")
    m = compile_urdf(str(urdf), frames_of_interest=("tool", "b"), inertia_mode="urdf")
    assert m.ndof == 2 and m.parent.tolist() == [-1, 0] and m.joint_names == ["j0", "j1"]
    assert m.body_names == ["a", "b", "tool"] and m.body_link.tolist() == [0, 1, 1]
    assert np.allclose(m.frames["tool"][1], [0.4, 0, 0]) and m.frames["tool"][0] == 1
    assert np.allclose(m.frames["b"][1], [0.2, 0, 0]) and abs(m.frames["b"][2][0, 1] + np.sin(1.57)) < 1e-12
    m2 = TGModel.from_npz(m.to_npz_dict())
    assert m2.ndof == 2 and np.allclose(m2.body_com, m.body_com) and set(m2.frames) == {"tool", "b"}
    assert m.dof_of_urdf_joint.tolist() == [0, 1, -1]


def test_malformed_urdf_numbers_are_read_strtod_style():
    from tactile_gym_amd.urdf_compile import _floats
    assert _floats("0.00443 -0.01409 4.96E-09+0.035") == [0.00443, -0.01409, 4.96e-09]


def test_rng_stream_is_splitmix64():
    """The oracle's task-randomisation stream (shared bit-for-bit with the HIP kernels)."""
    from oracle.ref_env import Rng
    r = Rng(1)
    a = [r.random() for _ in range(3)]
    assert all(0.0 <= x < 1.0 for x in a) and len(set(a)) == 3
    r2 = Rng(1)
    assert [r2.random() for _ in range(3)] == a
    assert Rng(2).random() != a[0]
    assert Rng.mix(0) == 0 and Rng.mix(1) == 0x5692161D100B05E5   # SplitMix64 finaliser known answer


def test_pb_math_matches_oracle_restatement():
    """tactile_gym_amd/pb_math.py (product side, batched) against oracle/pb_math.py (checker, scalar): the PyBullet frame helpers the
    oracle-observation vectors chain, including the gimbal branches of getEulerFromQuaternion."""
    from oracle import pb_math as O
    from tactile_gym_amd import pb_math as P
    rng = np.random.default_rng(0)
    rpy = rng.uniform(-3.1, 3.1, size=(64, 3))
    rpy[::8, 1] = np.pi / 2
    rpy[4::8, 1] = -np.pi / 2
    q = P.quat_from_euler(rpy)
    R = P.mat_from_quat(q)
    e = P.euler_from_quat(q)
    qm = P.quat_from_mat(R)
    p = rng.normal(size=(64, 3))
    ip, iq = P.invert_transform(p, q)
    for i in range(64):
        assert np.allclose(q[i], O.quat_from_euler(rpy[i]), atol=1e-15)
        assert np.allclose(R[i], O.mat_from_quat(q[i]), atol=1e-15)
        assert np.allclose(e[i], O.euler_from_quat(q[i]), atol=1e-12)
        assert np.allclose(qm[i], O.quat_from_mat(R[i]), atol=1e-15)
        op, oq = O.invert_transform(p[i], q[i])
        assert np.allclose(ip[i], op, atol=1e-15) and np.allclose(iq[i], oq, atol=1e-15)
    wf = P.WorkFrame([0.25, -0.1, 0.04], [-np.pi, 0.0, np.pi / 2])
    wp, wr = wf.pose(p, rpy)
    back_p, back_q = P.multiply_transforms(wf.pos, wf.orn, wp, P.quat_from_euler(wr))
    assert np.allclose(back_p, p, atol=1e-12) and np.allclose(np.abs(np.sum(back_q * q, axis=1)), 1.0, atol=1e-12)


def test_object_push_config_and_registry():
    """object_push-v0 host logic without a GPU: env-id resolution, the config struct filled from the reference's env_modes
    (object_push_env.py line by line), error behaviour on bad modes."""
    import tactile_gym_amd as tg
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd.rl_envs import object_push as op
    assert "object_push-v0" in tg.registered_ids()
    modes = dict(op.env_modes_default, arm_type="mg400", tactile_sensor_name="digitac", observation_mode="tactile_and_feature")
    cfg, robot, sensor, mesh, m, tip = op.build_config(8, 1000, (128, 128), modes)
    assert cfg.env_kind == capi.ENV_OBJECT_PUSH and cfg.action_repeat == 24 and cfg.solver_iterations == 150
    assert cfg.movement_mode == capi.PMOVE["TyRz"] and cfg.traj_n_points == 10 and cfg.cone_friction == 1
    assert abs(cfg.mu_tip - 0.65) < 1e-12 and abs(cfg.mu_table - 0.065) < 1e-12 and cfg.tip_stiffness == 300.0
    assert [cfg.workframe_pos[k] for k in range(3)] == [0.25, -0.1, 0.04] and abs(cfg.obj_init_pos[1] + 0.06) < 1e-15
    assert robot.ndof == 8 and robot.topology == 1 and cfg.n_tip_verts == tip.shape[0] > 100 and 0 <= cfg.tip_link < 5
    assert abs(cfg.obj_mass - 0.491) < 1e-9
    with pytest.raises(KeyError):
        op.build_config(8, 1000, (128, 128), {"movement_mode": "TyRz"})
    with pytest.raises(ValueError):
        op.build_config(8, 1000, (128, 128), dict(modes, movement_mode="sideways"))
    cfg_u, robot_u, sensor_u, _, _, tip_u = op.build_config(8, 1000, (128, 128), dict(modes, arm_type="ur5"))       # object_push_env.py:81-90
    assert robot_u.ndof == 6 and robot_u.topology == 0 and [cfg_u.workframe_pos[k] for k in range(3)] == [0.55, -0.2, 0.04]
    assert cfg_u.tcp_lims[1][1] == 0.1 and cfg_u.tip_link == 5
    cfg_t, _, sensor_t, _, _, _ = op.build_config(8, 1000, (128, 128), dict(modes, tactile_sensor_name="tactip"))   # :70-75
    assert cfg_t.workframe_pos[0] == 0.30 and sensor_t.struct.cam_pos[2] == 0.001                                     # mini_right_angle
    with pytest.raises(NotImplementedError):
        op.build_config(8, 1000, (128, 128), dict(modes, arm_type="franka_panda"))
    with pytest.raises(SystemExit):
        op.build_config(8, 1000, (128, 128), dict(modes, traj_type="zigzag"))


def test_control_mode_config():
    """control_mode handling without a GPU: TCP_position_control switches the action ranges to per-step pose changes (1 mm, 1 deg;
    edge_follow_env.py:143-153, base_surface_env.py:167-177) and sets blocking_move's step cap; unknown modes exit like robot.py:174."""
    import math
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd.rl_envs import edge_follow as ef, surface_follow as sf
    modes = dict(ef.env_modes_default, control_mode="TCP_position_control", movement_mode="xyzRz")
    cfg = ef.build_config(4, 200, (128, 128), modes)[0]
    assert cfg.control_mode == capi.CONTROL["TCP_position_control"] and cfg.max_blocking_steps == 10
    assert [cfg.act_hi[d] for d in range(6)] == [0.001, 0.001, 0.001, 0.0, 0.0, math.pi / 180]
    cfg = ef.build_config(4, 200, (128, 128), dict(modes, control_mode="TCP_velocity_control"))[0]
    assert cfg.control_mode == 0 and cfg.act_hi[0] == 0.01 and abs(cfg.act_hi[5] - 5 * math.pi / 180) < 1e-15
    smodes = dict(sf.env_modes_default, control_mode="TCP_position_control", arm_type="ur5", tactile_sensor_name="digit")
    cfg = sf.build_config(4, 200, (128, 128), smodes)[0]
    assert cfg.control_mode == 1 and [cfg.act_lo[d] for d in range(6)] == [-0.001] * 3 + [-math.pi / 180] * 2 + [0.0]
    with pytest.raises(NotImplementedError):
        ef.build_config(4, 200, (128, 128), dict(modes, control_mode="joint_velocity_control"))
    with pytest.raises(SystemExit):
        ef.build_config(4, 200, (128, 128), dict(modes, control_mode="teleport"))


def test_surface_follow_vertical_config_and_registry():
    """surface_follow-v2 host logic without a GPU: the vertical_simplex / xRz pairing, forward sensor, work frame and limits
    (base_surface_env.py:51-104, 181-191, 249-266), failure modes."""
    import math
    import tactile_gym_amd as tg
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd.rl_envs import surface_follow as sf
    assert "surface_follow-v2" in tg.registered_ids()
    m = dict(sf.env_modes_default_vert, arm_type="mg400", tactile_sensor_name="digitac", observation_mode="tactile")
    cfg, robot, sensor, _ = sf.build_config(8, 200, (128, 128), m)
    assert cfg.surf_vertical == 1 and cfg.movement_mode == capi.SMOVE["xRz"] and cfg.noise_mode == capi.SNOISE["vertical_simplex"]
    assert [cfg.stim_pos[k] for k in range(3)] == [0.33, 0.0, 0.175] and [cfg.workframe_rpy[k] for k in range(3)] == [-math.pi, 0.0, 0.0]
    assert [cfg.act_hi[d] for d in range(6)] == [0.01, 0.01, 0.0, 0.0, 0.0, 5.0 * (math.pi / 180)]
    assert cfg.tcp_lims[0][1] == 0.025 and cfg.tcp_lims[1][1] == 0.15 and cfg.tcp_lims[2][1] == 0.0 and abs(cfg.tcp_lims[5][1] - math.pi / 4) < 1e-15
    assert robot.ndof == 8 and sensor.struct.cam_pos[2] == 0.005                       # MG400, digitac `forward` camera offset
    with pytest.raises(SystemExit):
        sf.build_config(8, 200, (128, 128), dict(m, movement_mode="xyz"))               # vertical surface needs xRz
    with pytest.raises(SystemExit):
        sf.build_config(8, 200, (128, 128), dict(m, noise_mode="simplex"))              # xRz needs the vertical surface
    with pytest.raises(KeyError):
        sf.build_config(8, 200, (128, 128), dict(m, noise_mode="simplex", movement_mode="xyz"))   # no `standard` rest pose for the MG400 upstream


def test_object_roll_config_and_registry():
    """object_roll-v0 host logic without a GPU (object_roll_env.py line by line): flat TacTip, tip cylinder from the URDF, friction
    products, randomisation flags and ranges, failure modes."""
    import tactile_gym_amd as tg
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd.rl_envs import object_roll as orl
    assert "object_roll-v0" in tg.registered_ids()
    modes = dict(orl.env_modes_default, arm_type="ur5", tactile_sensor_name="tactip", observation_mode="tactile_and_feature",
                 rand_init_obj_pos=True, rand_obj_size=True, rand_embed_dist=True)
    cfg, robot, sensor, mesh, _ = orl.build_config(8, 250, (128, 128), modes)
    assert cfg.env_kind == capi.ENV_OBJECT_ROLL and cfg.action_repeat == 24 and cfg.termination_dist == 0.001
    assert abs(cfg.workframe_pos[2] - (2 * 0.0025 - 0.0015)) < 1e-15 and cfg.roll_radius == 0.0025 and abs(cfg.obj_mass - 0.05) < 1e-12
    assert cfg.mu_tip == 100.0 and cfg.mu_table == 10.0 and cfg.tip_stiffness == 10.0
    assert (cfg.roll_rand_init_pos, cfg.roll_rand_size, cfg.roll_rand_embed) == (1, 1, 1) and cfg.roll_goal_lo == 0.0 and cfg.roll_goal_hi == 0.015
    assert cfg.tip_link == 5 and cfg.tip_cyl_radius == 0.02 and abs(cfg.tip_cyl_half_len - 0.00325) < 1e-15
    assert mesh.struct.n_tris == 960 and robot.ndof == 6
    assert orl.build_config(8, 250, (128, 128), dict(modes, rand_init_obj_pos=False))[0].roll_goal_lo == 0.005
    with pytest.raises(ValueError):
        orl.build_config(8, 250, (128, 128), dict(modes, movement_mode="xyz"))
    with pytest.raises(NotImplementedError):
        orl.build_config(8, 250, (128, 128), dict(modes, arm_type="mg400"))
    assert orl.build_config(8, 250, (128, 128), dict(modes, control_mode="TCP_position_control"))[0].act_hi[0] == 0.001
    with pytest.raises(NotImplementedError):
        orl.build_config(8, 250, (128, 128), dict(modes, control_mode="joint_velocity_control"))


def test_env_classes_subclass_gym_and_sb3_when_importable(monkeypatch):
    """The reference's callers hand the envs to SB3 (Monitor, VecFrameStack, VecTransposeImage: sb3_helpers/rl_utils.py:17-35, 49-68),
    which checks isinstance(env, VecEnv) / gym.Env.  Neither package exists in the build image, so stand-ins are injected: with them
    importable, TactileVecEnv must BE a VecEnv and the single-env classes gym.Envs; without them the same classes are plain duck types."""
    import importlib
    import sys
    import types

    class FakeVecEnv:
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space

    class FakeGymEnv:
        pass

    sb3 = types.ModuleType("stable_baselines3")
    common = types.ModuleType("stable_baselines3.common")
    vec = types.ModuleType("stable_baselines3.common.vec_env")
    base = types.ModuleType("stable_baselines3.common.vec_env.base_vec_env")
    base.VecEnv = FakeVecEnv
    gym = types.ModuleType("gym")
    gym.Env = FakeGymEnv
    for name, mod in {"stable_baselines3": sb3, "stable_baselines3.common": common, "stable_baselines3.common.vec_env": vec,
                      "stable_baselines3.common.vec_env.base_vec_env": base, "gym": gym}.items():
        monkeypatch.setitem(sys.modules, name, mod)
    import tactile_gym_amd.vec_env as ve
    try:
        ve = importlib.reload(ve)
        assert issubclass(ve.TactileVecEnv, FakeVecEnv) and issubclass(ve.SingleTactileEnv, FakeGymEnv)
        for name in ("reset", "step_async", "step_wait", "close", "get_attr", "set_attr", "env_method", "env_is_wrapped", "seed", "render", "get_images"):
            assert callable(getattr(ve.TactileVecEnv, name)), name
        for name in ("reset", "step", "render", "seed", "close"):
            assert callable(getattr(ve.SingleTactileEnv, name)), name
    finally:
        for name in ("stable_baselines3", "stable_baselines3.common", "stable_baselines3.common.vec_env",
                     "stable_baselines3.common.vec_env.base_vec_env", "gym"):
            monkeypatch.delitem(sys.modules, name, raising=False)
        ve = importlib.reload(ve)
    assert ve.TactileVecEnv.__mro__[1] is object and ve.SingleTactileEnv.__mro__[1] is object


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The ctypes mirrors in tactile_gym_amd/_capi.py against include/tactile_gym_hip.h as a C compiler lays it out: sizeof of every struct
    that crosses the boundary and the offsets of a few late fields (a mismatch would silently corrupt every call; no GPU needed)."""
    import ctypes as C
    import os
    import subprocess
    from tactile_gym_amd import _capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    src.write_text('''
#include <stddef.h>
#include <stdio.h>
#include "tactile_gym_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(tg_robot), sizeof(tg_sensor), sizeof(tg_mesh), sizeof(tg_config), sizeof(tg_scene), sizeof(tg_state_view));
    printf("%zu %zu %zu %zu %zu\\n", offsetof(tg_config, contact_mapping), offsetof(tg_config, solver_iterations), offsetof(tg_scene, every_step),
           offsetof(tg_scene, body_heightfield), offsetof(tg_state_view, contact_ids));
    return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    sizes, offs = [int(x) for x in out[:6]], [int(x) for x in out[6:]]
    assert sizes == [C.sizeof(t) for t in (_capi.TgRobot, _capi.TgSensor, _capi.TgMesh, _capi.TgConfig, _capi.TgScene, _capi.TgStateView)]
    assert offs == [_capi.TgConfig.contact_mapping.offset, _capi.TgConfig.solver_iterations.offset, _capi.TgScene.every_step.offset,
                    _capi.TgScene.body_heightfield.offset, _capi.TgStateView.contact_ids.offset]


def test_constant_time_digitize_equals_the_edge_by_edge_count(tmp_path):
    """digitize_linspace (csrc/tg_kernels.hpp) locates the bin with one division and four exact comparisons instead of one per edge; the same
    body, compiled for the host, against the full count at and around every edge (tests/aux/digitize_check.cpp)."""
    import subprocess
    exe = tmp_path / "digitize_check"
    src = os.path.join(os.path.dirname(__file__), "aux", "digitize_check.cpp")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", src, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("0 mismatches"), out.stdout
    dev = open(os.path.join(os.path.dirname(__file__), "..", "tactile_gym_amd", "csrc", "tg_kernels.hpp")).read()
    assert "int base = (int)t - 1;" in dev and "for (int j = 0; j < 4; ++j)" in dev      # the device body is this one


class _FakeMonitor:
    """What stable_baselines3.common.monitor.Monitor is to HipVecEnv: a gym.Wrapper around the env (attribute `env`)."""

    def __init__(self, env):
        self.env = env

    def close(self):
        self.env.close()


def sb3_like_make_vec_env(env_id, n_envs=1, seed=None, start_index=0, env_kwargs=None, vec_env_cls=None, vec_env_kwargs=None):
    """The body of stable_baselines3.common.env_util.make_vec_env as the reference calls it (sb3_helpers/rl_utils.py:17-30, 49-57): a list
    of constructors, each gym.make(env_id, **env_kwargs) + seed(seed + rank) + Monitor, handed to vec_env_cls.  SB3 itself is not in the
    image; tg.make stands in for gym.make (the same registry)."""
    import tactile_gym_amd as tg
    env_kwargs, vec_env_kwargs = env_kwargs or {}, vec_env_kwargs or {}

    def make_env(rank):
        def _init():
            env = tg.make(env_id, **env_kwargs)
            if seed is not None:
                env.seed(seed + rank)
            return _FakeMonitor(env)
        return _init

    return vec_env_cls([make_env(i + start_index) for i in range(n_envs)], **vec_env_kwargs)


def test_hipvecenv_is_a_vec_env_cls_for_make_vec_env(monkeypatch, edge_modes):
    """HipVecEnv(env_fns): ONE probe env is constructed (to learn class + constructor arguments + base seed) and closed, then one N-env
    context with the same arguments, env i seeded seed + i.  No GPU here: the vectorised class is replaced by a recorder."""
    import tactile_gym_amd as tg
    from tactile_gym_amd import spaces
    from tactile_gym_amd.rl_envs import edge_follow as ef
    made = []

    class Recorder:
        def __init__(self, num_envs, max_steps=250, image_size=(64, 64), env_modes=None, physics_dtype="f64", auto_reset=True, device=0,
                     obs_mode="numpy", seed=None, **kw):
            self.args = dict(num_envs=num_envs, max_steps=max_steps, image_size=list(image_size), env_modes=dict(env_modes), physics_dtype=physics_dtype,
                             auto_reset=auto_reset, device=device, obs_mode=obs_mode, seed=seed, **kw)
            self.action_space = spaces.Box(low=-0.25, high=0.25, shape=(2,), dtype=np.float32)
            self.observation_space = spaces.Dict({})
            self.min_action, self.max_action, self.closed, self.seeds = -0.25, 0.25, False, []
            made.append(self)

        def seed(self, s=None):
            self.seeds.append(s)
            return [s]

        def close(self):
            self.closed = True

        def set_obs_transfer(self, how):
            self.transfer = how
            return self

    monkeypatch.setattr(ef.EdgeFollowEnv, "vec_cls", Recorder)
    venv = sb3_like_make_vec_env("edge_follow-v0", n_envs=6, seed=40, env_kwargs=dict(max_steps=77, image_size=[128, 128], env_modes=edge_modes),
                                 vec_env_cls=tg.HipVecEnv, vec_env_kwargs=dict(obs_mode="torch"))
    assert len(made) == 2                                     # the probe and the batch, not six single envs
    probe, batch = made
    assert venv is batch and probe.closed and not batch.closed
    assert probe.args["num_envs"] == 1 and probe.args["auto_reset"] is False and probe.seeds == [40]
    assert batch.args["num_envs"] == 6 and batch.args["seed"] == 40 and batch.args["auto_reset"] is True      # VecEnv semantics: auto-reset
    assert batch.args["max_steps"] == 77 and batch.args["image_size"] == [128, 128] and batch.args["env_modes"] == edge_modes
    assert batch.args["obs_mode"] == "torch"                  # vec_env_kwargs reach the vectorised constructor
    assert "obs_transfer" not in batch.args and not hasattr(batch, "transfer")
    made.clear()
    venv = sb3_like_make_vec_env("edge_follow-v0", n_envs=3, seed=1, env_kwargs=dict(env_modes=edge_modes), vec_env_cls=tg.HipVecEnv,
                                 vec_env_kwargs=dict(obs_transfer="tiles"))
    assert venv.transfer == "tiles" and "obs_transfer" not in venv.args      # the tile download is switched on after construction
    with pytest.raises(TypeError):
        tg.HipVecEnv([lambda: object()])
    with pytest.raises(ValueError):
        tg.HipVecEnv([])


def test_monitor_csv_has_sb3s_format_and_hipvecenv_takes_the_probe_monitors_directory(monkeypatch, edge_modes, tmp_path):
    """make_vec_env(..., monitor_dir=d) wraps every env constructor in Monitor(filename=d/<rank>); HipVecEnv builds ONE context, so it writes
    ONE Monitor file into that directory (sb3_helpers/rl_utils.py:22, 59; stable_baselines3's load_results - what the reference's
    rl_plot_utils.py / custom_callbacks.py call - reads every *monitor.csv): `#{"t_start": ..., "env_id": ...}`, `r,l,t`, one row per episode.
    The probe's header-only file is removed."""
    import csv
    import json
    import tactile_gym_amd as tg
    from tactile_gym_amd import spaces
    from tactile_gym_amd.rl_envs import edge_follow as ef
    from tactile_gym_amd.vec_env import MonitorCsv

    m = MonitorCsv(str(tmp_path / "a"), 123.5, "edge_follow-v0")
    m.write({"r": -12.25, "l": 200, "t": 3.5}); m.write({"r": 1.0, "l": 7, "t": 4.25}); m.close()
    lines = open(m.path).read().splitlines()
    assert m.path.endswith(".monitor.csv") and lines[0][0] == "#" and json.loads(lines[0][1:]) == {"t_start": 123.5, "env_id": "edge_follow-v0"}
    rows = list(csv.DictReader(lines[1:]))                     # exactly how stable_baselines3.common.monitor.load_results parses the body
    assert list(rows[0].keys()) == ["r", "l", "t"] and [float(r["r"]) for r in rows] == [-12.25, 1.0] and [int(r["l"]) for r in rows] == [200, 7]

    class Recorder:
        def __init__(self, num_envs, max_steps=250, image_size=(64, 64), env_modes=None, physics_dtype="f64", **kw):
            self.args = dict(num_envs=num_envs, **kw)
            self.action_space = spaces.Box(low=-0.25, high=0.25, shape=(2,), dtype=np.float32)
            self.observation_space = spaces.Dict({})
            self.min_action, self.max_action, self.monitor = -0.25, 0.25, None

        def seed(self, s=None):
            return [s]

        def close(self):
            pass

        def set_monitor(self, d, env_id=None):
            self.monitor = (d, env_id)

    class FakeMonitor:                                         # the two attributes of SB3's Monitor that name its file
        def __init__(self, env, filename):
            self.env = env
            fh = open(filename + ".monitor.csv", "w"); fh.write('#{"t_start": 0}\nr,l,t\n'); fh.flush()
            self.results_writer = type("RW", (), {"file_handler": fh})()

        def close(self):
            self.results_writer.file_handler.close()
            self.env.close()

    monkeypatch.setattr(ef.EdgeFollowEnv, "vec_cls", Recorder)
    d = tmp_path / "runs"
    d.mkdir()
    fns = [(lambda r=r: FakeMonitor(tg.make("edge_follow-v0", env_modes=edge_modes), str(d / str(r)))) for r in range(3)]
    venv = tg.HipVecEnv(fns)
    assert venv.monitor is not None and venv.monitor[0] == str(d)
    assert not (d / "0.monitor.csv").exists()                  # the probe's header-only file is gone


def test_lazy_info_shares_one_empty_dict_and_gives_finished_envs_their_own():
    from tactile_gym_amd import vec_env as ve
    assert ve._EMPTY_INFO == {} and isinstance(ve._EMPTY_INFO, dict)
    src = open(ve.__file__).read()
    assert "infos[i] = {\"episode\":" in src                  # a finished env never writes into the shared dict


def test_step_graphs_switch_mirrors_the_library(monkeypatch):
    """TG_STEP_GRAPH (csrc/tg_api.hip: step_as_graph): unset / 0 -> the step's launches go on the stream and nothing is ever captured (round 6
    default), non-zero -> one replayed hipGraph per step.  parallel.py asks this mirror before it quiesces the RCCL watchdog for a capture."""
    from tactile_gym_amd import _capi
    monkeypatch.delenv("TG_STEP_GRAPH", raising=False)
    assert _capi.step_graphs_enabled() is False
    for v, want in (("0", False), ("1", True), ("2", True), ("junk", False)):
        monkeypatch.setenv("TG_STEP_GRAPH", v)
        assert _capi.step_graphs_enabled() is want
