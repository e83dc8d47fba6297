"""GPU parity: every device function on the hot path against the CPU oracle, through the C ABI.

Tolerances (stated per north_star: "bit-exact for contact-pair indices, stated float tolerance for poses and tactile
depth"):
  * f64 physics vs the f64 oracle: joint angles / velocities 1e-9 (abs), torques and inertia 1e-9 relative — the two
    sides use different formulations (body-by-body sums vs merged links + composite recursion), so agreement is to
    rounding, not bitwise;
  * f32 physics vs the f64 oracle: 2e-4 relative on dynamics terms, 2e-5 rad on joint angles after one env step;
  * tactile image given identical float32 camera<-stimulus transforms: BIT-EXACT (uint8 equality on every pixel).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_states(rest, n, seed, spread=0.4, vel=0.5):
    rng = np.random.default_rng(seed)
    q = np.asarray(rest)[None] + spread * rng.standard_normal((n, len(rest)))
    qd = vel * rng.standard_normal((n, len(rest)))
    return q, qd, rng


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-10), ("f32", 2e-5)])
def test_fk_and_jacobian(ur5_tactip, dtype, tol):
    from tactile_gym_amd import hip_ops
    tg, mk_arm, robot, rest = ur5_tactip
    q, _, _ = _rand_states(rest, 128, 0)
    J, pos, rot = hip_ops.jacobian_tcp(robot, q, dtype)
    arm = mk_arm()
    for i in range(q.shape[0]):
        p, _, _, _, R = arm.link_state("tcp_link", q=q[i], qd=np.zeros(6))
        assert np.abs(pos[i] - p).max() < tol
        assert np.abs(rot[i] - R).max() < tol
        assert np.abs(J[i] - arm.jacobian("tcp_link", q[i])).max() < tol


@pytest.mark.parametrize("dtype,rtol", [("f64", 1e-9), ("f32", 2e-4)])
def test_inverse_dynamics_and_mass_matrix(ur5_tactip, dtype, rtol):
    from tactile_gym_amd import hip_ops
    tg, mk_arm, robot, rest = ur5_tactip
    q, qd, rng = _rand_states(rest, 128, 1)
    qdd = rng.standard_normal(q.shape)
    tau = hip_ops.inverse_dynamics(robot, q, qd, qdd, dtype)
    M = hip_ops.mass_matrix(robot, q, dtype)
    arm = mk_arm()
    for i in range(q.shape[0]):
        ref = arm.inverse_dynamics(q[i], qd[i], qdd[i])
        assert np.abs(tau[i] - ref).max() < rtol * (1.0 + np.abs(ref).max())
        Mref = arm.mass_matrix(q[i])
        assert np.abs(M[i] - Mref).max() < rtol * 10 * np.abs(Mref).max()


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-9), ("f32", 2e-5)])
@pytest.mark.parametrize("mode", ["velocity", "position"])
def test_sim_ticks(ur5_tactip, dtype, tol, mode):
    """24 x (gravity compensation + stepSimulation) from random states with saturating and non-saturating motors."""
    from oracle import minibullet as mb
    from tactile_gym_amd import _capi as capi, hip_ops
    tg, mk_arm, robot, rest = ur5_tactip
    n = 64
    q, qd, rng = _rand_states(rest, n, 2, spread=0.3, vel=0.2)
    qd_des = 0.1 * rng.standard_normal(q.shape)
    q_des = q + 0.002 * rng.standard_normal(q.shape)
    for max_force in (1000.0, 5.0):     # 5 N m saturates the shoulder motors: exercises the impulse clamp
        if mode == "velocity":
            q1, qd1 = hip_ops.sim_ticks(robot, q, qd, 24, capi.MOTOR_VELOCITY, qd_des=qd_des, max_force=max_force, dtype=dtype)
        else:
            q1, qd1 = hip_ops.sim_ticks(robot, q, qd, 24, capi.MOTOR_POSITION, q_des=q_des, qd_des=np.zeros_like(q), max_force=max_force, dtype=dtype)
        for i in range(n):
            arm = mk_arm()
            arm.reset_joint_states(q[i])
            for k in range(6):
                arm.state.qd[k] = qd[i, k]
            if mode == "velocity":
                arm.set_motors_velocity(qd_des[i], 1.0, max_force)
            else:
                arm.set_motors_position(q_des[i], np.zeros(6), 1.0, 1.0, max_force)
            for _ in range(24):
                arm.apply_torques(arm.inverse_dynamics(arm.q, arm.qd, np.zeros(6)))
                arm.step_simulation()
            scale = 1.0 if max_force > 100 else 50.0   # saturated motors leave large accelerations: errors scale with |qd|
            assert np.abs(q1[i] - arm.q).max() < tol * scale, (mode, max_force, i)
            assert np.abs(qd1[i] - arm.qd).max() < 100 * tol * scale, (mode, max_force, i)


@pytest.mark.parametrize("size", [64, 128, 256])
def test_render_bit_exact(size):
    """Raster + t_s_camera on random in-contact transforms: uint8 image identical to the oracle on every pixel."""
    from oracle import minibullet as mb
    from oracle.ref_env import OracleEdgeFollowEnv
    from tactile_gym_amd import hip_ops
    from tactile_gym_amd.robot_model import MeshDesc, SensorDesc
    sensor, mesh = SensorDesc("tactip", "standard", (size, size)), MeshDesc.load("long_edge")
    env = OracleEdgeFollowEnv(seed=3, image_size=(size, size))
    rng = np.random.default_rng(4)
    xfs = []
    for _ in range(24):
        env.reset()
        q = env.arm.q + 0.01 * rng.standard_normal(6)     # tilt / shift the sensor so all box faces get exercised
        env.arm.reset_joint_states(q)
        xfs.append(env.stimulus_transform())
    xfs = np.stack(xfs)
    imgs = hip_ops.render_tactile(sensor, mesh, xfs)
    touched = 0
    for i in range(xfs.shape[0]):
        cur = sensor.nodef_dep.copy()
        mb.render_depth(mesh.verts, mesh.tris, xfs[i], sensor.cam["fov"], sensor.cam["near"], sensor.cam["far"], size, size, cur)
        ref = mb.t_s_camera(cur, sensor.nodef_dep, sensor.nodef_gray, sensor.border_mask)
        assert np.array_equal(imgs[i], ref), f"image {i}: {(imgs[i] != ref).sum()} pixels differ"
        touched += int((ref[sensor.border_mask == 0] > 0).sum())
    assert touched > 100 * xfs.shape[0] / 4   # the cases really are in contact


def test_render_near_clip_and_ragged():
    """Triangles crossing the near plane / behind the camera / empty mesh (edge cases of getCameraImage)."""
    from oracle import minibullet as mb
    from tactile_gym_amd import hip_ops
    from tactile_gym_amd.robot_model import MeshDesc, SensorDesc
    sensor = SensorDesc("tactip", "standard", (128, 128))
    rng = np.random.default_rng(5)
    verts = (rng.uniform(-0.08, 0.08, size=(300, 3))).astype(np.float32)
    tris = rng.integers(0, 300, size=(1500, 3)).astype(np.int32)       # > one LDS batch (512 input triangles)
    mesh = MeshDesc(verts, tris)
    xfs = []
    for _ in range(6):
        A = np.linalg.qr(rng.standard_normal((3, 3)))[0]
        t = np.array([0.0, 0.0, -0.03]) + 0.01 * rng.standard_normal(3)  # camera inside the cloud: many near-plane crossings
        xfs.append(np.concatenate([A.reshape(9), t]))
    xfs = np.asarray(xfs, dtype=np.float32)
    imgs = hip_ops.render_tactile(sensor, mesh, xfs)
    for i in range(len(xfs)):
        cur = sensor.nodef_dep.copy()
        mb.render_depth(verts, tris, xfs[i], 60.0, 0.01, 1.0, 128, 128, cur)
        ref = mb.t_s_camera(cur, sensor.nodef_dep, sensor.nodef_gray, sensor.border_mask)
        assert np.array_equal(imgs[i], ref), f"image {i}: {(imgs[i] != ref).sum()} pixels differ"
    # empty mesh: zero-contact known answer  u8(nodef_gray) on the border, 0 elsewhere (SURVEY 8c known-answer 2)
    empty = MeshDesc(np.zeros((1, 3), np.float32), np.zeros((0, 3), np.int32))
    img = hip_ops.render_tactile(sensor, empty, xfs[:1])[0]
    expect = np.where(sensor.border_mask == 1, sensor.nodef_gray.astype(np.uint8), 0).astype(np.uint8)
    assert np.array_equal(img, expect)


@pytest.mark.parametrize("dtype,qtol,max_px", [("f64", 1e-9, 0), ("f32", 5e-5, 400)])
def test_env_reset_and_steps_match_oracle(edge_modes, dtype, qtol, max_px):
    """edge_follow-v0: reset + 6 random-action steps, 8 envs, against 8 independent oracle envs with the same seeds."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleEdgeFollowEnv
    n = 8
    venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=edge_modes, seed=11,
                       auto_reset=False, physics_dtype=dtype)
    oracles = [OracleEdgeFollowEnv(seed=11 + i, max_steps=200, image_size=(128, 128), env_modes=edge_modes) for i in range(n)]
    obs = venv.reset()
    ref_obs = [o.reset() for o in oracles]
    st = venv.get_state()
    for i, o in enumerate(oracles):
        assert st["edge_ang"][i] == o.edge_ang and st["embed_dist"][i] == o.embed_dist     # identical integer RNG stream
        if dtype == "f64":
            assert st["reset_ticks"][i] == o.reset_ticks
        assert np.abs(st["q"][i] - o.arm.q).max() < qtol * 10
        assert int((obs["tactile"][i] != ref_obs[i]["tactile"]).sum()) <= max_px
    rng = np.random.default_rng(12)
    for step in range(6):
        a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        obs, rew, done, info = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["qd_target"][i] - o.last_req_joint_vels).max() < max(qtol, 1e-9) * 100
            assert np.abs(st["q"][i] - o.arm.q).max() < qtol * 10, (step, i)
            assert np.abs(st["tcp_pos"][i] - o.cur_tcp_pos).max() < qtol * 10
            assert abs(rew[i] - rr) < (1e-5 if dtype == "f64" else 2e-4) and bool(done[i]) == rd
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= max_px, (step, i)
    venv.close()


def test_autoreset_terminal_observation(edge_modes):
    """VecEnv semantics: at max_steps the env reports done, keeps the terminal observation, and is already reset."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleEdgeFollowEnv
    n = 4
    venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=3, image_size=[128, 128], env_modes=edge_modes, seed=21, auto_reset=True)
    oracles = [OracleEdgeFollowEnv(seed=21 + i, max_steps=3, image_size=(128, 128), env_modes=edge_modes) for i in range(n)]
    venv.reset()
    for o in oracles:
        o.reset()
    rng = np.random.default_rng(22)
    for step in range(3):
        a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        obs, rew, done, info = venv.step(a)
        refs = [o.step(a[i]) for i, o in enumerate(oracles)]
        assert bool(done.all()) == (step == 2)
    for i, o in enumerate(oracles):
        assert np.array_equal(info[i]["terminal_observation"]["tactile"], refs[i][0]["tactile"])
        new = o.reset()
        assert np.array_equal(obs["tactile"][i], new["tactile"])
    st = venv.get_state()
    assert (st["step_count"] == 0).all()
    venv.close()


def test_full_size_properties(edge_modes):
    """BASELINE config 2 size (1024 envs, 128x128): size-independent properties.
    (a) determinism: two contexts with the same seeds produce identical images and states after 3 steps;
    (b) border invariance: every border pixel equals u8(nodef_gray) in every env, whatever the contact;
    (c) different seeds give different edges (no accidental stream sharing);
    (d) envs 0..7 of the big batch equal an 8-env batch with the same seeds (batch-size independence)."""
    import tactile_gym_amd as tg
    n = 1024
    rng = np.random.default_rng(31)
    acts = rng.uniform(-0.25, 0.25, size=(3, n, 2)).astype(np.float32)
    outs = []
    for rep in range(2):
        venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=edge_modes, seed=5)
        venv.reset()
        for k in range(3):
            obs, rew, done, _ = venv.step(acts[k])
        outs.append((obs["tactile"].copy(), rew.copy(), venv.get_state()))
        if rep == 0:
            border = venv._sensor.border_mask == 1
            gray = venv._sensor.nodef_gray.astype(np.uint8)
        venv.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2]["q"], outs[1][2]["q"])
    imgs = outs[0][0][..., 0]
    assert (imgs[:, border] == gray[border][None]).all()
    assert len(np.unique(outs[0][2]["edge_ang"])) == n
    assert (imgs[:, ~border].reshape(n, -1).max(axis=1) > 0).mean() > 0.9     # nearly every env is in contact
    small = tg.make_vec("edge_follow-v0", num_envs=8, max_steps=200, image_size=[128, 128], env_modes=edge_modes, seed=5)
    small.reset()
    for k in range(3):
        o8, r8, _, _ = small.step(acts[k, :8])
    assert np.array_equal(o8["tactile"], outs[0][0][:8]) and np.array_equal(r8, outs[0][1][:8])
    small.close()


def test_torch_zero_copy_and_device_actions(edge_modes):
    import torch
    import tactile_gym_amd as tg
    n = 64
    venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=edge_modes, seed=7, obs_mode="torch")
    obs = venv.reset()
    assert obs["tactile"].is_cuda and obs["tactile"].dtype == torch.uint8 and tuple(obs["tactile"].shape) == (n, 128, 128, 1)
    a = (torch.rand(n, 2, device="cuda") - 0.5) * 0.5
    obs2, rew, done, _ = venv.step(a)
    assert obs2["tactile"].data_ptr() == venv.tactile_device_ptr()
    host = venv.tactile_numpy()
    assert np.array_equal(obs2["tactile"].cpu().numpy(), host)
    venv.close()


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-11), ("f32", 5e-5)])
def test_inverse_kinematics(ur5_tactip, dtype, tol):
    """calculateInverseKinematics restatement (base_robot_arm.py:201-209): same root as the oracle, reached in the same
    number of iterations (f64), residual below the reference's residualThreshold = 1e-8."""
    from oracle import pb_math as pm
    from tactile_gym_amd import hip_ops
    tg, mk_arm, robot, rest = ur5_tactip
    arm = mk_arm()
    rng = np.random.default_rng(6)
    n = 32
    q_true = np.asarray(rest)[None] + 0.05 * rng.standard_normal((n, 6))
    tps, trs, refs, its = [], [], [], []
    for i in range(n):
        p, quat, _, _, R = arm.link_state("tcp_link", q=q_true[i], qd=np.zeros(6))
        arm.reset_joint_states(rest)
        q = arm.q.copy()
        from oracle import minibullet as mb
        link, fpos, frot = tg.frames["tcp_link"]
        it = arm.L.mb_ik(mb.C.byref(arm.model), link, mb._dp(np.ascontiguousarray(fpos)), mb._dp(np.ascontiguousarray(frot)),
                         mb._dp(np.ascontiguousarray(p)), mb._dp(np.ascontiguousarray(R.reshape(9))), mb._dp(q), 100, 1e-8)
        tps.append(p); trs.append(R); refs.append(q); its.append(it)
    q, iters = hip_ops.inverse_kinematics(robot, np.tile(rest, (n, 1)), np.array(tps), np.array(trs), dtype=dtype)
    assert np.abs(q - np.array(refs)).max() < tol
    if dtype == "f64":
        assert np.array_equal(iters, np.array(its)) and iters.max() < 12
        for i in range(n):
            p, _, _, _, R = arm.link_state("tcp_link", q=q[i], qd=np.zeros(6))
            assert np.abs(p - tps[i]).max() < 1e-8 and np.abs(R - trs[i]).max() < 1e-8


SURF_MODES = dict(movement_mode="xyzRxRy", control_mode="TCP_velocity_control", noise_mode="simplex", observation_mode="tactile",
                  reward_mode="dense", arm_type="ur5", tactile_sensor_name="digit")


def test_heightfield_generation_bit_exact():
    """gen_heigtfield_simplex_2d (base_surface_env.py:319-337): device OpenSimplex == oracle OpenSimplex, every bit."""
    from oracle.ref_env import opensimplex_heightfield
    from tactile_gym_amd import hip_ops
    seeds = np.array([0, 1, 3, 12345, 97984136, 99999999, 2 ** 40 + 17, -5], dtype=np.int64)
    h, z = hip_ops.gen_heightfield(seeds)
    for k, s in enumerate(seeds):
        ref = opensimplex_heightfield(int(s))
        assert np.array_equal(h[k], ref), f"seed {s}: {(h[k] != ref).sum()} samples differ"
        rf = ref.astype(np.float32)
        assert z[k] == np.float32(0.5) * (rf.min() + rf.max())
        assert np.abs(ref).max() <= 0.025 and np.abs(ref).max() > 0.005


def test_render_heightfield_bit_exact():
    """Tactile image of a per-env heightfield (7 938 triangles each, several LDS batches) — uint8 equal to the oracle."""
    from oracle import minibullet as mb
    from oracle.ref_env import OracleSurfaceFollowAutoEnv
    from tactile_gym_amd import hip_ops
    from tactile_gym_amd.robot_model import SensorDesc
    sensor = SensorDesc("digit", "standard", (128, 128))
    rng = np.random.default_rng(8)
    hs, zs, xfs, refs = [], [], [], []
    for k in range(10):
        env = OracleSurfaceFollowAutoEnv(seed=100 + k, env_modes=SURF_MODES, center_z=(k % 2 == 0))
        env.reset()
        q = env.arm.q + np.array([0, 0.004, -0.004, 0.02, 0.02, 0.0]) * rng.standard_normal(6)   # press in / tilt
        env.arm.reset_joint_states(q)
        hs.append(env.heightfield_data); zs.append(env.surf_zoff); xfs.append(env.stimulus_transform())
        refs.append(env.tactile_image())
    imgs = hip_ops.render_tactile_heightfield(sensor, np.stack(hs), np.array(zs, np.float32), np.stack(xfs))
    touched = 0
    for k in range(len(refs)):
        assert np.array_equal(imgs[k], refs[k]), f"image {k}: {(imgs[k] != refs[k]).sum()} pixels differ"
        touched += int((refs[k] > 0).sum())
    assert touched > 2000


def test_surface_follow_env_matches_oracle():
    """surface_follow-v0 (UR5 + DIGIT, BASELINE config 3 modes): reset + 5 steps, 6 envs vs 6 oracle envs."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleSurfaceFollowAutoEnv
    n = 6
    venv = tg.make_vec("surface_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=SURF_MODES, seed=51, auto_reset=False)
    oracles = [OracleSurfaceFollowAutoEnv(seed=51 + i, max_steps=200, image_size=(128, 128), env_modes=SURF_MODES) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    st = venv.get_state()
    for i, o in enumerate(oracles):
        assert np.array_equal(st["heights"][i], o.heightfield_data) and st["surf_zoff"][i] == o.surf_zoff
        assert np.abs(st["goal_pos"][i] - o.goal_pos_world).max() < 1e-12
        assert st["reset_ticks"][i] == o.reset_ticks
        assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8
        assert np.array_equal(obs["tactile"][i], ref[i]["tactile"])
    rng = np.random.default_rng(52)
    for step in range(5):
        a = rng.uniform(-0.25, 0.25, size=(n, 3)).astype(np.float32)
        a[:, 0] = np.abs(a[:, 0])          # bias z into the surface so the images are not empty
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (step, i)
            assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd
            assert np.array_equal(obs["tactile"][i], ro["tactile"]), (step, i, int((obs["tactile"][i] != ro["tactile"]).sum()))
    venv.close()


BAL_MODES = dict(movement_mode="xy", control_mode="TCP_velocity_control", object_mode="pole", rand_gravity=True, rand_embed_dist=True,
                 observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")


@pytest.mark.parametrize("size,mapping", [(128, "wave"), (256, "wave"), (128, "lane")])
def test_object_balance_env_matches_oracle(size, mapping):
    """object_balance-v0 (UR5 + TacTip, pole on a point-to-point constraint; BASELINE config 5 modes, 256x256 there): two
    consecutive episodes (the second reset drags the fallen pole along, base_object_env.py:146-173), 6 envs vs 6 oracle envs.
    Joint angles 1e-8 rad, pole pose 1e-7 (the coupled solve runs in a different but equivalent form), images bit-exact
    except where a 1e-8 pose difference straddles a float32 rounding (<= 3 pixels allowed per image)."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectBalanceEnv
    n = 6
    venv = tg.make_vec("object_balance-v0", num_envs=n, max_steps=8, image_size=[size, size], env_modes=BAL_MODES, seed=71, auto_reset=False,
                       contact_mapping=mapping)   # wave: k_step_body_wave (tick-parallel kinematics); lane: k_step_body (the mapping of >= 4096 envs)
    oracles = [OracleObjectBalanceEnv(seed=71 + i, max_steps=8, image_size=(size, size), env_modes=BAL_MODES) for i in range(n)]
    rng = np.random.default_rng(72)
    for episode in range(2):
        obs = venv.reset()
        ref = [o.reset() for o in oracles]
        st = venv.get_state()
        for i, o in enumerate(oracles):
            assert st["gravity_z"][i] == o.gravity and st["embed_dist"][i] == o.embed_dist
            assert st["reset_ticks"][i] == o.reset_ticks
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8
            assert np.abs(st["body_pos"][i] - o.body_pose()[0]).max() < 1e-12
            assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 3
        for step in range(8):
            a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
            obs, rew, done, _ = venv.step(a)
            st = venv.get_state()
            for i, o in enumerate(oracles):
                ro, rr, rd, _ = o.step(a[i])
                pos, R = o.body_pose()
                assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (episode, step, i)
                assert np.abs(st["body_pos"][i] - pos).max() < 1e-7 and np.abs(st["body_rot"][i] - R).max() < 1e-7, (episode, step, i)
                assert rew[i] == rr and bool(done[i]) == rd
                assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 3, (episode, step, i)
        assert done.all()
    venv.close()


def test_object_balance_autoreset_forked_reset_matches_oracle():
    """object_balance with auto_reset: the reset of the finished envs runs on a second stream beside the render of the step's
    observations (fork/join inside the step graph).  Two auto-resets in a row: terminal observations, post-reset
    observations and states against oracle envs that are stepped and reset by hand."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectBalanceEnv
    n, size = 4, 128
    venv = tg.make_vec("object_balance-v0", num_envs=n, max_steps=3, image_size=[size, size], env_modes=BAL_MODES, seed=91, auto_reset=True)
    oracles = [OracleObjectBalanceEnv(seed=91 + i, max_steps=3, image_size=(size, size), env_modes=BAL_MODES) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    rng = np.random.default_rng(92)
    n_resets = 0
    for step in range(7):
        a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        obs, rew, done, info = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert rew[i] == rr and bool(done[i]) == rd, (step, i)
            if rd:
                n_resets += 1
                assert int((info[i]["terminal_observation"]["tactile"] != ro["tactile"]).sum()) <= 3, (step, i)
                ro = o.reset()
                assert st["gravity_z"][i] == o.gravity and st["embed_dist"][i] == o.embed_dist and st["step_count"][i] == 0
            pos, R = o.body_pose()
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (step, i)
            assert np.abs(st["body_pos"][i] - pos).max() < 1e-7 and np.abs(st["body_rot"][i] - R).max() < 1e-7, (step, i)
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 3, (step, i)
    assert n_resets >= 2 * n
    venv.close()


def test_mg400_tree_functions(mg400_tactip):
    """MG400 (8 control joints, two-branch tree): FK / Jacobian / inverse dynamics / inertia /
    24 sim ticks / IK against the oracle, f64."""
    from oracle import minibullet as mb
    from tactile_gym_amd import _capi as capi, hip_ops
    tg, mk_arm, robot, rest = mg400_tactip
    assert robot.ndof == 8 and robot.topology == 1
    n = 48
    q, qd, rng = _rand_states(rest, n, 9, spread=0.2, vel=0.3)
    qdd = rng.standard_normal(q.shape)
    J, pos, rot = hip_ops.jacobian_tcp(robot, q)
    tau = hip_ops.inverse_dynamics(robot, q, qd, qdd)
    M = hip_ops.mass_matrix(robot, q)
    qd_des = 0.1 * rng.standard_normal(q.shape)
    q1, qd1 = hip_ops.sim_ticks(robot, q, qd, 24, capi.MOTOR_VELOCITY, qd_des=qd_des, max_force=1000.0)
    arm = mk_arm()
    for i in range(n):
        p, _, _, _, R = arm.link_state("tcp_link", q=q[i], qd=np.zeros(8))
        assert np.abs(pos[i] - p).max() < 1e-10 and np.abs(rot[i] - R).max() < 1e-10
        assert np.abs(J[i] - arm.jacobian("tcp_link", q[i])).max() < 1e-10
        ref = arm.inverse_dynamics(q[i], qd[i], qdd[i])
        assert np.abs(tau[i] - ref).max() < 1e-9 * (1 + np.abs(ref).max())
        Mref = arm.mass_matrix(q[i])
        assert np.abs(M[i] - Mref).max() < 1e-8 * np.abs(Mref).max()
        a2 = mk_arm()
        a2.reset_joint_states(q[i])
        for k in range(8):
            a2.state.qd[k] = qd[i, k]
        a2.set_motors_velocity(qd_des[i], 1.0, 1000.0)
        for _ in range(24):
            a2.apply_torques(a2.inverse_dynamics(a2.q, a2.qd, np.zeros(8)))
            a2.step_simulation()
        assert np.abs(q1[i] - a2.q).max() < 1e-8 and np.abs(qd1[i] - a2.qd).max() < 1e-6


def test_edge_follow_mg400_env_matches_oracle(edge_modes):
    """edge_follow-v0 with arm_type mg400: pseudo-inverse controller + hand-slaved linkage joints (mg400.py:77-129), reset
    with the target_joints override (:191-232); 6 envs vs 6 oracle envs."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleEdgeFollowEnv
    modes = dict(edge_modes, arm_type="mg400")
    n = 6
    venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=modes, seed=91, auto_reset=False)
    oracles = [OracleEdgeFollowEnv(seed=91 + i, max_steps=200, image_size=(128, 128), env_modes=modes) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    st = venv.get_state()
    for i, o in enumerate(oracles):
        assert st["reset_ticks"][i] == o.reset_ticks
        assert np.abs(st["q"][i] - o.arm.q).max() < 1e-7
        assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 2
    rng = np.random.default_rng(92)
    for step in range(5):
        a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["qd_target"][i] - o.last_req_joint_vels).max() < 1e-7, (step, i)
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-7, (step, i)
            assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 2, (step, i)
    venv.close()


PUSH_MODES = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=True, rand_obj_mass=True, traj_type="simplex",
                  observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")


@pytest.mark.gpu
@pytest.mark.parametrize("arm,sensor,movement,traj,mapping", [
    ("mg400", "digitac", "TyRz", "simplex", "wave"), ("mg400", "digitac", "TyRz", "simplex", "lane"), ("mg400", "tactip", "xyRz", "straight", "wave"),
    ("mg400", "digit", "TxTyRz", "simplex", "lane"), ("mg400", "digit", "TxTyRz", "simplex", "wave"), ("ur5", "digitac", "TyRz", "simplex", "wave"),
    ("ur5", "tactip", "TxTyRz", "straight", "lane"), ("ur5", "tactip", "TxTyRz", "straight", "wave")])
def test_object_push_env_matches_oracle(arm, sensor, movement, traj, mapping):
    """object_push-v0 (BASELINE config 4: MG400 + DigiTac right-angle sensor, cube on the table, tip collision core ON): rigid
    contacts cube-table and cube-tip with friction.  Two consecutive episodes (the second Robot.reset runs with the cube where the
    first episode left it), 4 envs vs 4 oracle envs.  Contact dynamics amplify rounding differences (the HIP tick uses FMA
    contraction and the residual-free PGS bookkeeping), so: joints 1e-9 rad, cube pose 1e-8, reward 1e-6, extended_feature to
    float32, goal index / done exact, tactile images within 3 pixels; the simplex goal trajectory is bit-exact.  Both mappings of the
    contact solve (tg_config.contact_mapping: one wavefront per env / one lane per env) against the same oracle."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectPushEnv
    modes = dict(PUSH_MODES, arm_type=arm, tactile_sensor_name=sensor, movement_mode=movement, traj_type=traj)   # MG400 + TacTip: the
    n, steps, size = 4, 7, 128                                                   # mini_right_angle sensor; UR5: object_push_env.py:81-90
    act_dim = {"TyRz": 2, "xyRz": 3, "TxTyRz": 3}[movement]
    venv = tg.make_vec("object_push-v0", num_envs=n, max_steps=steps, image_size=[size, size], env_modes=modes, seed=31, auto_reset=False,
                       contact_mapping=mapping)
    assert venv.observation_space["extended_feature"].shape == (12,) and venv.action_space.shape == (act_dim,)
    oracles = [OracleObjectPushEnv(seed=31 + i, max_steps=steps, image_size=(size, size), env_modes=modes) for i in range(n)]
    rng = np.random.default_rng(5)
    touched, tip_ids, knife_edges = 0, set(), 0
    for episode in range(2):
        obs = venv.reset()
        ref = [o.reset() for o in oracles]
        st = venv.get_state()
        for i, o in enumerate(oracles):
            assert st["reset_ticks"][i] == o.reset_ticks and st["obj_mass"][i] == o.cube.mass and st["goal_id"][i] == o.targ_traj_list_id
            assert st["contact_count"][i] == o.scene.n_contacts and list(st["contact_ids"][i]) == list(o.scene.contact_ids)   # last blocking-move tick
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-9
            assert np.abs(st["body_pos"][i] - o.cube_pose()[0]).max() < 1e-12 and np.abs(st["body_rot"][i] - o.cube_pose()[1]).max() < 1e-12
            if traj == "simplex":
                assert np.array_equal(st["traj"][i][:2, :10].T, o.traj_pos_work[:, :2]) and np.array_equal(st["traj"][i][2, :10], o.traj_rpy_work[:, 2])
            else:
                assert np.abs(st["traj"][i][:2, :10].T - o.traj_pos_work[:, :2]).max() < 1e-15
                assert np.abs(st["traj"][i][2, :10] - o.traj_rpy_work[:, 2]).max() < 1e-12
            assert np.abs(obs["extended_feature"][i] - ref[i]["extended_feature"]).max() < 1e-6
            assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 3
        for step in range(steps):
            a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
            if movement != "TyRz":
                a[:, 0] = np.abs(a[:, 0])          # keep pushing forward
            obs, rew, done, _ = venv.step(a)
            st = venv.get_state()
            for i, o in enumerate(oracles):
                goal_before = o.goal_pos_world.copy()
                ro, rr, rd, _ = o.step(a[i])
                pos, R = o.cube_pose()
                assert np.abs(st["q"][i] - o.arm.q).max() < 1e-9, (episode, step, i)
                assert np.abs(st["body_pos"][i] - pos).max() < 1e-8 and np.abs(st["body_rot"][i] - R).max() < 1e-8, (episode, step, i)
                if st["goal_id"][i] == o.targ_traj_list_id + 1 and abs(np.linalg.norm(pos - goal_before) - o.termination_pos_dist) < 1e-12:
                    # PARITY_ASSUMPTIONS A29: goal 0 sits EXACTLY termination_pos_dist from the cube's start, so until the tip moves the
                    # cube `pos_dist < termination_pos_dist` (object_push_env.py:520-537) is decided by the 1e-17 m resting noise of the
                    # contact solve - in the reference as here.  Not a parity datum: follow the HIP side and carry on.
                    assert o._update_goal()
                    ro = o._observation()
                    knife_edges += 1
                assert abs(rew[i] - rr) < 1e-6 and bool(done[i]) == rd and st["goal_id"][i] == o.targ_traj_list_id
                assert np.abs(obs["extended_feature"][i] - ro["extended_feature"]).max() < 1e-6
                assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 3, (episode, step, i)
                # contact-pair indices of the step's last sim tick: BIT-EXACT (count and ids in solver row order)
                assert st["contact_count"][i] == o.scene.n_contacts, (episode, step, i)
                assert list(st["contact_ids"][i]) == list(o.scene.contact_ids), (episode, step, i, st["contact_ids"][i], list(o.scene.contact_ids))
                touched += int(o.scene.tip_depth < 1.0)
                tip_ids.add(int(o.scene.contact_ids[o.scene.n_contacts - 1]))
        assert done.all()
    assert touched > n * steps          # the tip core really pushed the cube in most steps
    assert max(tip_ids) >= 8            # and the compared ids include tip-core hull vertices, not only table corners
    assert knife_edges <= 2 * n         # at most the first step of each env's episodes
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mapping", ["wave", "lane"])
def test_object_push_with_saturating_motors_matches_oracle(mapping):
    """The wave mapping solves a tick with the joint motors unclamped while watching their impulses, and solves it again with the literal
    clamped step when the limit max_force * dt was reached.  With a 2 N m limit (the reference's arms: 1000) the motors saturate whenever the
    commanded velocity changes: the states must still follow the oracle (whose motors clamp in every sweep), and differ from the 1000 N m run."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectPushEnv
    modes = dict(PUSH_MODES, arm_type="ur5", tactile_sensor_name="digitac", movement_mode="TyRz", traj_type="straight")
    n, steps = 4, 5
    weak = tg.make_vec("object_push-v0", num_envs=n, max_steps=50, image_size=[64, 64], env_modes=modes, seed=9, auto_reset=False,
                       contact_mapping=mapping, max_force=2.0)
    strong = tg.make_vec("object_push-v0", num_envs=n, max_steps=50, image_size=[64, 64], env_modes=modes, seed=9, auto_reset=False, contact_mapping=mapping)
    oracles = [OracleObjectPushEnv(seed=9 + i, max_steps=50, image_size=(64, 64), env_modes=modes) for i in range(n)]
    weak.reset(); strong.reset()
    for o in oracles:
        o.reset()
        o.max_force = 2.0                     # from the first step on (the reset's blocking move has run with the default on both sides)
    rng = np.random.default_rng(2)
    for step in range(steps):
        a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        weak.step(a); strong.step(a)
        sw, ss = weak.get_state(), strong.get_state()
        for i, o in enumerate(oracles):
            o.step(a[i])
            assert np.abs(sw["q"][i] - o.arm.q).max() < 1e-9 and np.abs(sw["qd"][i] - o.arm.qd).max() < 1e-7, (step, i)
            assert np.abs(sw["body_pos"][i] - o.cube_pose()[0]).max() < 1e-8, (step, i)
    assert np.abs(sw["q"] - ss["q"]).max() > 1e-5    # the limit did bite
    weak.close(); strong.close()


@pytest.mark.gpu
def test_object_push_f32_and_autoreset():
    """f32 physics variant of object_push: finite, the cube moves forward and stays on the table; auto-reset hands back the terminal
    observation (tactile + extended_feature) and restarts the episode at goal 0."""
    import tactile_gym_amd as tg
    modes = dict(PUSH_MODES, rand_init_orn=False, rand_obj_mass=False)
    n, steps = 64, 5
    # 128x128: the reference's committed 64x64 DigiTac right-angle nodef fixture is stale (different depth range, blank gray image)
    venv = tg.make_vec("object_push-v0", num_envs=n, max_steps=steps, image_size=[128, 128], env_modes=modes, seed=3, physics_dtype="f32")
    venv.reset()
    y0 = venv.get_state()["body_pos"][:, 1].copy()
    for step in range(steps):
        obs, rew, done, infos = venv.step(np.zeros((n, 2), np.float32))
    st = venv.get_state()
    assert done.all() and all("terminal_observation" in i for i in infos)
    term = infos[0]["terminal_observation"]
    assert term["tactile"].shape == (128, 128, 1) and term["extended_feature"].shape == (12,) and (term["tactile"] > 0).sum() > 100
    assert np.isfinite(rew).all() and np.isfinite(st["q"]).all()
    assert (st["goal_id"] == 0).all() and (st["step_count"] == 0).all()
    assert np.abs(st["body_pos"][:, 1] - y0).max() < 1e-12          # teleported back by the auto-reset
    assert term["extended_feature"][0] > 0.002                      # the TCP advanced along the work-frame push direction (x)
    venv.close()


@pytest.mark.gpu
def test_default_solver_equals_literal_solver(edge_modes):
    """The default tick (PGS convergence exit + analytic fixed point where licensed) against pgs_full_sweeps=1
    (dynamics + exactly 150 sweeps in every tick) on the same seeds and actions: 40 steps of 64 envs incl. auto-resets.  Joints agree
    to 1e-11 rad, rewards to float32 rounding, dones exactly, images bit-exact but for float32-rounding straddles (<= 2 pixels)."""
    import tactile_gym_amd as tg
    n, steps = 64, 40
    a = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=15, image_size=[128, 128], env_modes=edge_modes, seed=5)
    b = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=15, image_size=[128, 128], env_modes=edge_modes, seed=5, pgs_full_sweeps=True)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa["tactile"], ob["tactile"])
    rng = np.random.default_rng(8)
    for step in range(steps):
        act = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        sa, sb = a.get_state(), b.get_state()
        assert np.abs(sa["q"] - sb["q"]).max() < 1e-11 and np.abs(sa["qd"] - sb["qd"]).max() < 1e-11, step
        assert np.array_equal(da, db) and np.abs(ra - rb).max() < 1e-6
        assert np.array_equal(sa["reset_ticks"], sb["reset_ticks"])
        assert int((oa["tactile"] != ob["tactile"]).sum(axis=(1, 2, 3)).max()) <= 2, step
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,size,max_steps", [("edge_follow-v0", 128, 200), ("surface_follow-v0", 128, 200), ("object_balance-v0", 256, 250)])
def test_default_solver_equals_literal_solver_at_config_scale(env_id, size, max_steps, edge_modes):
    """The guard of the headline rate at the scale it is quoted on: 1024 envs x 260 steps (crossing the max_steps auto-resets of every
    env, plus the early ones of envs that reach their goal / drop the pole) of BASELINE configs 2, 3 and 5, default solver (convergence
    exit + run-time-licensed analytic fixed point) against pgs_full_sweeps=1 on the same seeds and actions.  Asserted:
    joints and joint velocities within 1e-11 rad (rad/s) at every checkpoint, dones and reset tick counts exactly equal on every
    step, rewards to float32 rounding, and at most 2 differing pixels per image (float32 straddles of the camera transform)."""
    import tactile_gym_amd as tg
    modes = {"edge_follow-v0": edge_modes, "surface_follow-v0": SURF_MODES, "object_balance-v0": BAL_MODES}[env_id]
    act_dim = {"edge_follow-v0": 2, "surface_follow-v0": 3, "object_balance-v0": 2}[env_id]
    n, steps = 1024, 260
    kw = dict(num_envs=n, max_steps=max_steps, image_size=[size, size], env_modes=modes, seed=9)
    a = tg.make_vec(env_id, **kw)
    b = tg.make_vec(env_id, pgs_full_sweeps=True, **kw)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa["tactile"], ob["tactile"])
    rng = np.random.default_rng(21)
    worst_q, worst_px, n_done = 0.0, 0, 0
    for step in range(steps):
        act = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert np.array_equal(da, db), step
        assert np.abs(ra - rb).max() < 1e-6, step
        n_done += int(da.sum())
        px = int((oa["tactile"] != ob["tactile"]).sum(axis=(1, 2, 3)).max())
        worst_px = max(worst_px, px)
        assert px <= 2, step
        if step % 20 == 19 or step == steps - 1:
            sa, sb = a.get_state(), b.get_state()
            dq = max(float(np.abs(sa["q"] - sb["q"]).max()), float(np.abs(sa["qd"] - sb["qd"]).max()))
            worst_q = max(worst_q, dq)
            assert dq < 1e-11, (step, dq)
            assert np.array_equal(sa["reset_ticks"], sb["reset_ticks"]) and np.array_equal(sa["step_count"], sb["step_count"])
            if "body_pos" in sa:
                assert np.abs(sa["body_pos"] - sb["body_pos"]).max() < 1e-10 and np.abs(sa["body_rot"] - sb["body_rot"]).max() < 1e-9
    assert n_done >= n                     # every env went through at least one auto-reset
    print(f"{env_id}: worst |dq| {worst_q:.2e}, worst differing pixels {worst_px}, resets {n_done}")
    a.close(); b.close()


@pytest.mark.gpu
def test_oracle_observation_vectors():
    """observation_mode "oracle" (SURVEY 8f rank 1): the feature vectors of surface_follow (20), object_balance (26) and object_push
    (30) against the CPU oracle's restatement of the reference's get_oracle_obs, after a reset and three steps; float32, 1e-5."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectBalanceEnv, OracleObjectPushEnv, OracleSurfaceFollowAutoEnv
    cases = [("surface_follow-v0", OracleSurfaceFollowAutoEnv, dict(SURF_MODES, observation_mode="oracle"), 3, 20),
             ("object_balance-v0", OracleObjectBalanceEnv, dict(BAL_MODES, observation_mode="oracle"), 2, 26),
             ("object_push-v0", OracleObjectPushEnv, dict(PUSH_MODES, observation_mode="oracle"), 2, 30)]
    for env_id, ocls, modes, act_dim, dim in cases:
        n = 3
        venv = tg.make_vec(env_id, num_envs=n, max_steps=50, image_size=[128, 128], env_modes=modes, seed=11, auto_reset=False)
        assert venv.observation_space["oracle"].shape == (dim,)
        oracles = [ocls(seed=11 + i, max_steps=50, image_size=(128, 128), env_modes=modes) for i in range(n)]
        obs = venv.reset()
        for o in oracles:
            o.reset()
        rng = np.random.default_rng(4)
        for step in range(3):
            a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
            obs, _, _, _ = venv.step(a)
            for i, o in enumerate(oracles):
                o.step(a[i])
                ref = o.oracle_obs()
                assert obs["oracle"].shape == (n, dim) and ref.shape == (dim,)
                assert np.abs(obs["oracle"][i] - ref).max() < 1e-5, (env_id, step, i, obs["oracle"][i], ref)
            # the vector is computed on the device (tg_get_obs_oracle); the host route from a state read-back agrees to float32 rounding
            assert np.abs(venv.oracle_obs_host() - obs["oracle"]).max() < 2e-6, (env_id, step)
        venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,modes_name,dim", [("edge_follow-v0", "EDGE", 10), ("surface_follow-v2", "VERT", 20), ("object_roll-v0", "ROLL", 34),
                                                   ("object_push-v0", "PUSH", 30), ("object_balance-v0", "BAL", 26)])
def test_device_oracle_observation_equals_host_route(env_id, modes_name, dim, edge_modes):
    """tg_get_obs_oracle (one kernel, no state read-back) against oracle_obs_host (tg_get_state + numpy) on 256 envs over 12 steps with
    auto-resets, both array modes; 2e-6 (float32 rounding of values computed in double on either side)."""
    import torch
    import tactile_gym_amd as tg
    import bench
    modes = dict({"EDGE": edge_modes, "VERT": bench.VERT_MODES, "ROLL": bench.ROLL_MODES, "PUSH": bench.PUSH_MODES, "BAL": bench.BAL_MODES}[modes_name],
                 observation_mode="oracle")
    for obs_mode in ("numpy", "torch"):
        venv = tg.make_vec(env_id, num_envs=256, max_steps=5, image_size=[64, 64], env_modes=modes, seed=3, obs_mode=obs_mode)
        obs = venv.reset()
        rng = np.random.default_rng(0)
        for step in range(12):
            a = rng.uniform(-0.25, 0.25, size=(256, venv.act_dim)).astype(np.float32)
            obs, _, _, _ = venv.step(a)
            got = obs["oracle"].cpu().numpy() if obs_mode == "torch" else obs["oracle"]
            assert got.shape == (256, dim) and got.dtype == np.float32
            assert np.abs(got - venv.oracle_obs_host()).max() < 2e-6, (obs_mode, step)
        venv.close()


@pytest.mark.gpu
def test_full_size_properties_surface_balance_push():
    """BASELINE configs 3-5 at their per-GPU size (1024 envs; balance at 256x256): run-to-run determinism and domain invariants that
    do not need the oracle: finite state, the pole stays tied to the TCP, the cube stays on the table and is pushed forward, goal
    indices in range, the tactile image shows contact in most envs, every heightfield sample within the generator's range."""
    import tactile_gym_amd as tg
    n = 1024
    cases = [("surface_follow-v0", dict(SURF_MODES), 3, 128, 3), ("object_balance-v0", dict(BAL_MODES), 2, 256, 3),
             ("object_push-v0", dict(PUSH_MODES, rand_init_orn=True, rand_obj_mass=True), 2, 128, 4)]
    for env_id, modes, act_dim, size, steps in cases:
        rng = np.random.default_rng(17)
        acts = rng.uniform(-0.25, 0.25, size=(steps, n, act_dim)).astype(np.float32)
        if env_id == "object_push-v0":
            acts[:] = 0.0                                    # TyRz with zero input: straight ahead at the maximum pushing speed
        outs = []
        for rep in range(2):
            venv = tg.make_vec(env_id, num_envs=n, max_steps=200, image_size=[size, size], env_modes=modes, seed=9, auto_reset=False)
            venv.reset()
            st0 = venv.get_state()
            for k in range(steps):
                obs, rew, done, _ = venv.step(acts[k])
            outs.append((obs["tactile"].copy(), rew.copy(), done.copy(), venv.get_state(), st0))
            venv.close()
        (img, rew, done, st, st0), (img2, rew2, done2, st2, _) = outs
        assert np.array_equal(img, img2) and np.array_equal(rew, rew2) and np.array_equal(done, done2), env_id
        assert np.array_equal(st["q"], st2["q"]), env_id
        assert np.isfinite(st["q"]).all() and np.isfinite(rew).all() and img.shape == (n, size, size, 1)
        contact = (img.reshape(n, -1) > 0).any(axis=1).mean()
        if env_id == "surface_follow-v0":
            assert np.abs(st["heights"]).max() <= 0.025 * 0.8660254 + 1e-12      # OpenSimplex 2-D range x height_range
            assert len(np.unique(st["heights"][:, 5, 7])) > n // 2               # per-env surfaces (noise2(0, 0) is 0 for every seed)
            assert contact > 0.5
        elif env_id == "object_balance-v0":
            gap = np.linalg.norm(st["body_pos"] - st0["body_pos"], axis=1)
            assert (gap < 0.1).all() and np.isfinite(st["body_rot"]).all()       # 3 steps: nobody has drifted 0.1 m yet
            assert np.abs(np.linalg.det(st["body_rot"]) - 1.0).max() < 1e-9      # orientation stays a rotation
        else:
            assert np.abs(st["body_pos"][:, 2] - 0.04).max() < 2e-3              # the cube rests on the table
            fwd = st["body_pos"][:, 1] - st0["body_pos"][:, 1]
            assert (fwd > 5e-4).mean() > 0.95                                    # pushed forward (world +y) in nearly every env
            assert ((st["goal_id"] >= 0) & (st["goal_id"] <= 10)).all()
            assert (st["obj_mass"] >= 0.4).all() and (st["obj_mass"] <= 0.8).all()
            assert contact > 0.9


@pytest.mark.gpu
def test_surface_follow_goal_env_matches_oracle():
    """surface_follow-v1 (goal variant, surface_follow_goal_env.py): 5-D actions, dense reward with the goal term, `extended_feature`
    = [tcp_pos, goal_pos] in the work frame; 4 envs vs 4 oracle envs, tactile_and_feature."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleSurfaceFollowGoalEnv
    modes = dict(SURF_MODES, observation_mode="tactile_and_feature")
    n = 4
    venv = tg.make_vec("surface_follow-v1", num_envs=n, max_steps=20, image_size=[128, 128], env_modes=modes, seed=21, auto_reset=False)
    assert venv.action_space.shape == (5,) and venv.observation_space["extended_feature"].shape == (6,)
    oracles = [OracleSurfaceFollowGoalEnv(seed=21 + i, max_steps=20, image_size=(128, 128), env_modes=modes) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    for i in range(n):
        assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) == 0
        assert np.abs(obs["extended_feature"][i] - ref[i]["extended_feature"]).max() < 1e-6
    rng = np.random.default_rng(22)
    for step in range(4):
        a = rng.uniform(-0.25, 0.25, size=(n, 5)).astype(np.float32)
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-9, (step, i)
            assert abs(rew[i] - rr) < 1e-6 and bool(done[i]) == rd
            assert np.abs(obs["extended_feature"][i] - ro["extended_feature"]).max() < 1e-6
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 2, (step, i)
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,mode,act_dim", [("surface_follow-v0", "yz", 1), ("surface_follow-v0", "yzRx", 2),
                                                 ("surface_follow-v1", "yz", 2), ("surface_follow-v1", "yzRx", 3)])
def test_surface_follow_1d_modes_match_oracle(env_id, mode, act_dim):
    """The 1-D surface modes (gen_heigtfield_simplex_1d, base_surface_env.py:339-357; goal direction = choice([-1, 1]) along y,
    :526-528): surface constant along x, per-mode action encodings, W_norm = 0 for "yz"; 4 envs vs 4 oracle envs."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleSurfaceFollowAutoEnv, OracleSurfaceFollowGoalEnv
    modes = dict(SURF_MODES, movement_mode=mode)
    Oracle = OracleSurfaceFollowAutoEnv if env_id.endswith("v0") else OracleSurfaceFollowGoalEnv
    n = 4
    venv = tg.make_vec(env_id, num_envs=n, max_steps=20, image_size=[128, 128], env_modes=modes, seed=61, auto_reset=False)
    assert venv.action_space.shape == (act_dim,)
    oracles = [Oracle(seed=61 + i, max_steps=20, image_size=(128, 128), env_modes=modes) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    st = venv.get_state()
    for i, o in enumerate(oracles):
        assert np.array_equal(st["heights"][i], o.heightfield_data) and np.ptp(st["heights"][i], axis=0).max() == 0.0
        assert np.abs(st["goal_pos"][i] - o.goal_pos_world).max() < 1e-12
        assert np.array_equal(obs["tactile"][i], ref[i]["tactile"])
    assert len({float(o.workframe_directions[1]) for o in oracles} | {-1.0, 1.0}) == 2
    rng = np.random.default_rng(62)
    for step in range(4):
        a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (step, i)
            assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 2, (step, i)
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,overrides,act_dim", [
    ("edge_follow-v0", dict(movement_mode="xyzRz"), 4),
    ("edge_follow-v0", dict(movement_mode="xy", arm_type="mg400"), 2),
    ("surface_follow-v0", dict(), 3),
    ("surface_follow-v1", dict(), 5)])
def test_tcp_position_control_matches_oracle(env_id, overrides, act_dim, edge_modes):
    """control_mode = TCP_position_control: the action is a work-frame pose delta (+-1 mm, +-1 deg), the target is
    clipped to the TCP limits, solved by inverse kinematics from the current joint state and tracked by POSITION_CONTROL motors
    through blocking_move(max_steps=10, constant_vel=None) (base_robot_arm.py:228-279, mg400.py:131-190, robot.py:188-260).
    4 envs vs 4 oracle envs: joints 1e-8 rad, reward 1e-5, image within 2 pixels."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleEdgeFollowEnv, OracleSurfaceFollowAutoEnv, OracleSurfaceFollowGoalEnv
    base = edge_modes if env_id.startswith("edge") else SURF_MODES
    modes = dict(base, control_mode="TCP_position_control", **overrides)
    Oracle = {"edge_follow-v0": OracleEdgeFollowEnv, "surface_follow-v0": OracleSurfaceFollowAutoEnv,
              "surface_follow-v1": OracleSurfaceFollowGoalEnv}[env_id]
    n = 4
    venv = tg.make_vec(env_id, num_envs=n, max_steps=20, image_size=[128, 128], env_modes=modes, seed=71, auto_reset=False)
    assert venv.action_space.shape == (act_dim,)
    oracles = [Oracle(seed=71 + i, max_steps=20, image_size=(128, 128), env_modes=modes) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    for i in range(n):
        assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 2
    rng = np.random.default_rng(72)
    moved = 0.0
    for step in range(6):
        a = rng.uniform(-0.3, 0.3, size=(n, act_dim)).astype(np.float32)      # beyond +-0.25: the clip of scale_actions is exercised
        q_before = venv.get_state()["q"].copy()
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        moved = max(moved, float(np.abs(st["q"] - q_before).max()))
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (step, i, np.abs(st["q"][i] - o.arm.q).max())
            assert np.abs(st["qd"][i] - o.arm.qd).max() < 1e-6, (step, i)
            assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 2, (step, i)
    assert moved > 1e-4                                                       # the arm did move
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["object_balance-v0", "object_push-v0"])
def test_tcp_position_control_with_free_body_matches_oracle(env_id):
    """TCP_position_control on the tasks with a free body: the blocking move runs the coupled arm + pole (P2P constraint) ticks /
    the arm + cube contact ticks.  4 envs vs 4 oracle envs; tolerances as in the velocity-control tests of the same envs."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectBalanceEnv, OracleObjectPushEnv
    push = env_id.startswith("object_push")
    modes = dict(PUSH_MODES if push else BAL_MODES, control_mode="TCP_position_control")
    if not push:
        modes["movement_mode"] = "xyRxRy"
    Oracle, act_dim = (OracleObjectPushEnv, 2) if push else (OracleObjectBalanceEnv, 4)
    n = 4
    venv = tg.make_vec(env_id, num_envs=n, max_steps=30, image_size=[128, 128], env_modes=modes, seed=81, auto_reset=False)
    assert venv.action_space.shape == (act_dim,)
    oracles = [Oracle(seed=81 + i, max_steps=30, image_size=(128, 128), env_modes=modes) for i in range(n)]
    venv.reset()
    for o in oracles:
        o.reset()
    rng = np.random.default_rng(82)
    for step in range(8):
        a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            pos, R = o.cube_pose() if push else o.body_pose()
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (step, i, np.abs(st["q"][i] - o.arm.q).max())
            assert np.abs(st["body_pos"][i] - pos).max() < 1e-7 and np.abs(st["body_rot"][i] - R).max() < 1e-7, (step, i)
            assert abs(rew[i] - rr) < 1e-6 and bool(done[i]) == rd
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 3, (step, i)
    venv.close()


@pytest.mark.gpu
def test_pipelined_shard_and_packed_outputs(edge_modes):
    """The device-resident rollout path bench.py times: TorchShard(pipelined=True) enqueues steps on a torch stream without a host wait.
    Same seeds and actions as the blocking path -> identical images / rewards / dones after 30 steps (auto-reset on); the packed output
    block [obs | pad | reward | done] (tg_get_packed_outputs) aliases the three per-field views."""
    import torch
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import TorchShard
    n = 128
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    acts = [torch.empty(n, 2, device="cuda").uniform_(-0.25, 0.25, generator=g) for _ in range(30)]
    outs = []
    for pipelined in (False, True):
        venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=12, image_size=[128, 128], env_modes=edge_modes, seed=3, auto_reset=True,
                           obs_mode="torch")
        shard = TorchShard(venv, pipelined=pipelined)
        ctx = torch.cuda.stream(shard.stream) if pipelined else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            shard.reset()
            hist = []
            for a in acts:
                obs, rew, done, _ = shard.step(a)
                hist.append((obs["tactile"].clone(), rew.clone(), done.clone()))      # ordered after the step on the same stream
            torch.cuda.synchronize()
        packed, off, foff = shard.packed()
        assert foff == -1 and off == (n * 128 * 128 + 15) // 16 * 16 and packed.numel() == off + 5 * n
        assert torch.equal(packed[:n * 128 * 128], obs["tactile"].reshape(-1))
        assert torch.equal(packed[off:off + 4 * n].view(torch.float32), rew) and torch.equal(packed[off + 4 * n:], done)
        outs.append([(t.cpu(), r.cpu(), d.cpu()) for t, r, d in hist])
        venv.close()
    assert any(bool(d.any()) for _, _, d in outs[0])                                   # episodes ended and were reset inside the steps
    for (t0, r0, d0), (t1, r1, d1) in zip(*outs):
        assert torch.equal(t0, t1) and torch.equal(r0, r1) and torch.equal(d0, d1)


@pytest.mark.gpu
@pytest.mark.parametrize("noise", ["none", "random"])
def test_surface_follow_noise_modes_match_oracle(noise):
    """surface_follow noise_mode "none" (flat surface) and "random" (gen_heigtfield_noisey, base_surface_env.py:302-317: one uniform
    draw per 2x2 block, evaluated in parallel from the env's RNG state): heights bit-exact, two consecutive episodes (the RNG stream
    moves past the block draws), 4 envs vs 4 oracle envs."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleSurfaceFollowAutoEnv
    modes = dict(SURF_MODES, noise_mode=noise)
    n = 4
    venv = tg.make_vec("surface_follow-v0", num_envs=n, max_steps=3, image_size=[128, 128], env_modes=modes, seed=31, auto_reset=False)
    oracles = [OracleSurfaceFollowAutoEnv(seed=31 + i, max_steps=3, image_size=(128, 128), env_modes=modes) for i in range(n)]
    rng = np.random.default_rng(32)
    for episode in range(2):
        obs = venv.reset()
        ref = [o.reset() for o in oracles]
        st = venv.get_state()
        for i, o in enumerate(oracles):
            assert np.array_equal(st["heights"][i], o.heightfield_data) and st["surf_zoff"][i] == o.surf_zoff
            assert np.abs(st["goal_pos"][i] - o.goal_pos_world).max() < 1e-12
            assert np.array_equal(obs["tactile"][i], ref[i]["tactile"])
        if noise == "random":
            assert len(np.unique(st["heights"][0])) > 500 and st["heights"].max() <= 0.005
        for step in range(3):
            a = rng.uniform(-0.25, 0.25, size=(n, 3)).astype(np.float32)
            a[:, 0] = np.abs(a[:, 0])
            obs, rew, done, _ = venv.step(a)
            st = venv.get_state()
            for i, o in enumerate(oracles):
                ro, rr, rd, _ = o.step(a[i])
                assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (episode, step, i)
                assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd
                assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 2, (episode, step, i)
    venv.close()


@pytest.mark.gpu
def test_surface_follow_sparse_reward_matches_oracle():
    """reward_mode "sparse" of surface_follow (surface_follow_auto_env.py:59-73): the dense reward is accumulated over the episode
    (starting with the reset pose, base_surface_env.py:640) and paid out on the step that reaches the goal.  Flat surface so the
    auto-driven TCP does reach the goal (150 mm at 1 mm per step); 3 envs vs 3 oracle envs until every env is done."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleSurfaceFollowAutoEnv
    modes = dict(SURF_MODES, noise_mode="none", reward_mode="sparse", tactile_sensor_name="tactip")
    n = 3
    venv = tg.make_vec("surface_follow-v0", num_envs=n, max_steps=200, image_size=[64, 64], env_modes=modes, seed=41, auto_reset=False)
    oracles = [OracleSurfaceFollowAutoEnv(seed=41 + i, max_steps=200, image_size=(64, 64), env_modes=modes) for i in range(n)]
    venv.reset()
    for o in oracles:
        o.reset()
    finished = np.zeros(n, dtype=bool)
    paid = np.zeros(n)
    a = np.zeros((n, 3), dtype=np.float32)
    for step in range(200):
        obs, rew, done, _ = venv.step(a)
        for i, o in enumerate(oracles):
            if finished[i]:
                continue
            ro, rr, rd, _ = o.step(a[i])
            assert abs(rew[i] - rr) <= 1e-5 * max(1.0, abs(rr)), (step, i, rew[i], rr)
            assert bool(done[i]) == rd, (step, i)
            if rd:
                finished[i], paid[i] = True, rr
        if finished.all():
            break
    assert finished.all() and step < 199 and (paid < 0).all()          # every env reached the goal and was paid its accumulated reward
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arm,sensor,obs_mode", [("mg400", "tactip", "tactile_and_feature"), ("mg400", "digitac", "tactile"), ("ur5", "digit", "tactile")])
def test_surface_follow_vertical_env_matches_oracle(arm, sensor, obs_mode):
    """surface_follow-v2 (surface_follow_vert_env.py + the vertical_simplex branches of base_surface_env.py): upright heightfield
    (rotated -90 deg about y), `forward` sensor, movement xRz (x and yaw from the agent, y auto-driven), reward -(10 surf + 3 cos);
    two episodes, 4 envs vs 4 oracle envs, plus the 20-d oracle observation."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleSurfaceFollowVertEnv
    modes = dict(movement_mode="xRz", control_mode="TCP_velocity_control", noise_mode="vertical_simplex", observation_mode=obs_mode,
                 reward_mode="dense", arm_type=arm, tactile_sensor_name=sensor)                   # surface_follow_vert_params.py
    n = 4
    venv = tg.make_vec("surface_follow-v2", num_envs=n, max_steps=4, image_size=[128, 128], env_modes=modes, seed=91, auto_reset=False)
    assert venv.action_space.shape == (2,)
    oracles = [OracleSurfaceFollowVertEnv(seed=91 + i, max_steps=4, image_size=(128, 128), env_modes=modes) for i in range(n)]
    rng = np.random.default_rng(92)
    seen = 0
    for episode in range(2):
        obs = venv.reset()
        ref = [o.reset() for o in oracles]
        st = venv.get_state()
        for i, o in enumerate(oracles):
            assert np.array_equal(st["heights"][i], o.heightfield_data) and np.ptp(st["heights"][i], axis=1).max() == 0.0   # varies along x only
            assert np.abs(st["goal_pos"][i] - o.goal_pos_world).max() < 1e-12
            assert st["reset_ticks"][i] == o.reset_ticks
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8
            assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 2
        for step in range(4):
            a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
            a[:, 0] = np.abs(a[:, 0])                       # push into the surface
            obs, rew, done, _ = venv.step(a)
            st = venv.get_state()
            oo = venv.oracle_obs()
            for i, o in enumerate(oracles):
                ro, rr, rd, _ = o.step(a[i])
                assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (episode, step, i)
                assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd
                assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 2, (episode, step, i)
                assert np.abs(oo[i] - o.oracle_obs()).max() < 1e-5, (episode, step, i)
                if "feature" in obs_mode:
                    assert np.abs(obs["extended_feature"][i] - o.extended_feature()).max() < 1e-6
                seen += int((ro["tactile"] > 0).any())
        assert done.all()
    assert seen > n * 4                                     # the sensor did touch the surface
    venv.close()


ROLL_MODES = dict(movement_mode="xy", control_mode="TCP_velocity_control", rand_init_obj_pos=True, rand_obj_size=True, rand_embed_dist=True,
                  observation_mode="tactile_and_feature", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")   # object_roll_params.py


@pytest.mark.gpu
@pytest.mark.parametrize("rand,control,mapping", [(True, "TCP_velocity_control", "wave"), (True, "TCP_velocity_control", "lane"),
                                                  (False, "TCP_velocity_control", "wave"), (True, "TCP_position_control", "wave"),
                                                  (True, "TCP_position_control", "lane")])
def test_object_roll_env_matches_oracle(rand, control, mapping):
    """object_roll-v0 (UR5 + flat TacTip, marble between the table and the tip's collision cylinder, soft tip contact, goal in the TCP
    frame): two consecutive episodes (the second Robot.reset runs with the marble of the first still in the world), 4 envs vs 4
    oracle envs.  Joints 1e-9 rad, marble pose 1e-8, reward 1e-6, images within 3 pixels, extended_feature and the 34-d oracle vector."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectRollEnv
    modes = dict(ROLL_MODES, rand_init_obj_pos=rand, rand_obj_size=rand, rand_embed_dist=rand, control_mode=control)
    n, steps = 4, 6
    venv = tg.make_vec("object_roll-v0", num_envs=n, max_steps=steps, image_size=[128, 128], env_modes=modes, seed=11, auto_reset=False,
                       contact_mapping=mapping)
    assert venv.action_space.shape == (2,) and venv.observation_space["extended_feature"].shape == (3,)
    oracles = [OracleObjectRollEnv(seed=11 + i, max_steps=steps, image_size=(128, 128), env_modes=modes) for i in range(n)]
    rng = np.random.default_rng(12)
    rolled = 0.0
    for episode in range(2):
        obs = venv.reset()
        ref = [o.reset() for o in oracles]
        st = venv.get_state()
        for i, o in enumerate(oracles):
            assert st["reset_ticks"][i] == o.reset_ticks and st["obj_mass"][i] == o.scaled_obj_radius and st["embed_dist"][i] == o.embed_dist
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-9
            assert np.abs(st["body_pos"][i] - o.ball_pose()[0]).max() < 1e-12
            assert np.abs(st["goal_pos"][i] - o.goal_pos_tcp).max() < 1e-15
            assert np.abs(obs["extended_feature"][i] - ref[i]["extended_feature"]).max() < 1e-7
            assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 3
        start = st["body_pos"].copy()
        for step in range(steps):
            a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
            obs, rew, done, _ = venv.step(a)
            st = venv.get_state()
            oo = venv.oracle_obs()
            for i, o in enumerate(oracles):
                ro, rr, rd, _ = o.step(a[i])
                pos, R = o.ball_pose()
                assert np.abs(st["q"][i] - o.arm.q).max() < 1e-9, (episode, step, i)
                assert np.abs(st["body_pos"][i] - pos).max() < 1e-8 and np.abs(st["body_rot"][i] - R).max() < 1e-8, (episode, step, i)
                assert abs(rew[i] - rr) < 1e-6 and bool(done[i]) == rd
                assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 3, (episode, step, i)
                assert st["contact_count"][i] == o.scene.n_contacts and list(st["contact_ids"][i]) == list(o.scene.contact_ids), (episode, step, i)
                ref_o, got_o = o.oracle_obs().copy(), oo[i].copy()
                for sl in (slice(3, 7), slice(16, 20)):          # orientations are quaternions: q and -q are the same rotation (the work
                    if np.dot(ref_o[sl], got_o[sl]) < 0:         # frame's roll of -pi puts a resting marble right on the +-pi branch cut)
                        got_o[sl] = -got_o[sl]
                assert np.abs(got_o - ref_o).max() < 2e-5, (episode, step, i, np.abs(got_o - ref_o).argmax())
        rolled = max(rolled, float(np.abs(st["body_pos"] - start)[:, :2].max()))
        assert done.all()
    if rand:
        assert rolled > (1e-3 if control == "TCP_velocity_control" else 1e-4)   # embed distances above the 1.75 mm skin-to-core gap: it rolls
    venv.close()


@pytest.mark.gpu
def test_object_roll_reset_onto_the_previous_episodes_marble():
    """Regression: with rand_obj_size the robot is reset while the PREVIOUS episode's (possibly twice as large) marble is still in the
    scene (object_roll_env.py:203-237 reloads it only afterwards); the rest pose can then put the tip's collision cylinder around the
    marble's centre - no contact by the closest-point rule (oracle: skipped), and the disabled rows must stay finite.  Envs 27 and 163
    of this seed hit it (NaN joints before the fix); the whole batch must stay finite and those two must match the oracle's reset."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectRollEnv
    modes = dict(ROLL_MODES, observation_mode="oracle")
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-0.25, 0.25, size=(256, 2)).astype(np.float32) for _ in range(5)]
    for mapping in ("wave", "lane"):
        venv = tg.make_vec("object_roll-v0", num_envs=256, max_steps=5, image_size=[64, 64], env_modes=modes, seed=3, contact_mapping=mapping)
        venv.reset()
        for a in acts:
            obs, _, done, _ = venv.step(a)
        assert done[[27, 163, 5]].all()        # (an env that met its goal earlier was reset then and is mid-episode now)
        st = venv.get_state()
        assert all(np.isfinite(v).all() for v in st.values() if v.dtype.kind == "f") and np.isfinite(obs["oracle"]).all()
        for i in (27, 163, 5):
            o = OracleObjectRollEnv(seed=3 + i, max_steps=5, image_size=(64, 64), env_modes=modes)
            o.reset()
            for a in acts:
                o.step(a[i])
            o.reset()
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-9, (mapping, i)
        venv.close()


@pytest.mark.gpu
def test_contact_indices_run_to_run_deterministic_full_size():
    """1024 object_push envs (BASELINE config 4's per-GPU shard), two independent contexts with the same seeds and actions: the contact
    sets (integer data) and the whole state are bit-identical run to run, step by step, and the contact sets are the physically
    expected ones (four table corners of the resting cube's bottom face, plus a tip-core vertex once the tip touches)."""
    import tactile_gym_amd as tg
    n, steps = 1024, 6
    a = tg.make_vec("object_push-v0", num_envs=n, max_steps=1000, image_size=[128, 128], env_modes=PUSH_MODES, seed=77, auto_reset=False)
    b = tg.make_vec("object_push-v0", num_envs=n, max_steps=1000, image_size=[128, 128], env_modes=PUSH_MODES, seed=77, auto_reset=False)
    a.reset(); b.reset()
    rng = np.random.default_rng(3)
    seen_tip = 0
    for step in range(steps):
        act = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        a.step(act); b.step(act)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa["contact_count"], sb["contact_count"]) and np.array_equal(sa["contact_ids"], sb["contact_ids"]), step
        assert np.array_equal(sa["q"], sb["q"]) and np.array_equal(sa["body_pos"], sb["body_pos"]) and np.array_equal(sa["body_rot"], sb["body_rot"])
        ids, cnt = sa["contact_ids"], sa["contact_count"]
        assert ((cnt >= 4) & (cnt <= 5)).all()                                   # the cube rests on four corners; the tip adds one
        table = np.sort(ids[:, :4], axis=1)
        # init orientation rpy (-pi, 0, pi/2): the bottom face is the +z face of the cube frame, i.e. the vertices with iz = 1 (odd ids)
        assert (table == np.array([1, 3, 5, 7])).all(), step
        tip = ids[np.arange(n), cnt - 1]
        assert ((cnt == 4) | (tip >= 8)).all()
        seen_tip += int((cnt == 5).sum())
    assert seen_tip > n                                                          # most envs are pushing by the end
    a.close(); b.close()


@pytest.mark.gpu
def test_full_size_properties_roll_and_vertical():
    """object_roll-v0 and surface_follow-v2 at 1024 envs: run-to-run determinism and domain invariants without the oracle: the marble
    stays on the table at its radius and rolls half as far as the tip that drives it (pure rolling between table and tip) in the envs
    whose embed distance reaches the tip's collision core; the upright surface varies along one axis only and is touched by most envs."""
    import tactile_gym_amd as tg
    n = 1024
    modes_roll = dict(ROLL_MODES, rand_init_obj_pos=False)
    outs = []
    for rep in range(2):
        venv = tg.make_vec("object_roll-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=modes_roll, seed=5, auto_reset=False)
        venv.reset()
        st0 = venv.get_state()
        a = np.tile(np.array([[0.25, 0.0]], dtype=np.float32), (n, 1))          # +x in the work frame at the maximum speed
        for k in range(5):
            obs, rew, done, _ = venv.step(a)
        outs.append((obs["tactile"].copy(), rew.copy(), venv.get_state(), st0))
        venv.close()
    (img, rew, st, st0), (img2, rew2, st2, _) = outs
    assert np.array_equal(img, img2) and np.array_equal(rew, rew2) and np.array_equal(st["q"], st2["q"]) and np.array_equal(st["body_pos"], st2["body_pos"])
    assert np.isfinite(st["body_pos"]).all() and np.abs(st["body_pos"][:, 2] - st["obj_mass"]).max() < 1e-4      # resting on the table
    assert (st["obj_mass"] >= 0.0025).all() and (st["obj_mass"] <= 0.005).all() and (st["embed_dist"] >= 0.0015).all()
    tip = np.linalg.norm(st["tcp_pos"][:, :2] - st0["tcp_pos"][:, :2], axis=1)
    ball = np.linalg.norm(st["body_pos"][:, :2] - st0["body_pos"][:, :2], axis=1)
    # the tip's collision core (cylinder cap) sits 1.75 mm behind the TCP: how far it reaches into the marble after the reset move
    # (blocking_move may stop one tick short of its target, robot.py:216-258, so the TCP height itself is used, not the nominal embed)
    reach = 2 * st["obj_mass"] - (st0["tcp_pos"][:, 2] + 0.00175)
    pressed = reach > 1.5e-4
    assert pressed.sum() > n // 2 and np.abs(ball[pressed] / tip[pressed] - 0.5).max() < 0.05
    assert (ball[reach < -1.5e-4] < 1e-6).all()                                 # not touched: does not move

    modes_v = dict(movement_mode="xRz", control_mode="TCP_velocity_control", noise_mode="vertical_simplex", observation_mode="tactile",
                   reward_mode="dense", arm_type="mg400", tactile_sensor_name="tactip")
    venv = tg.make_vec("surface_follow-v2", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=modes_v, seed=6, auto_reset=False)
    obs = venv.reset()
    st = venv.get_state()
    assert np.ptp(st["heights"], axis=2).max() == 0.0 and len(np.unique(st["heights"][:, 9, 0])) > n // 2
    assert set(np.unique(st["direction"][:, 1])) == {-1.0, 1.0} and (st["direction"][:, 0] == 0).all()
    for k in range(3):
        obs, rew, done, _ = venv.step(np.tile(np.array([[0.1, 0.0]], dtype=np.float32), (n, 1)))
    assert np.isfinite(rew).all() and (rew <= 0).all() and ((obs["tactile"].reshape(n, -1) > 0).any(axis=1).mean() > 0.9)
    venv.close()


def test_raster_division_is_correctly_rounded():
    """The depth interpolation divides with a reciprocal + residual corrections instead of the range-scaled expansion of `/`
    (DESIGN 5): on 2^26 operand pairs over the pixel-space exponent range every quotient must carry the bits of the IEEE division."""
    import ctypes
    from tactile_gym_amd import _capi as capi
    for seed in (1, 20240927):
        m = ctypes.c_int64(-1)
        assert 0 == (capi.test_lib().tg_selftest_division(1 << 26, seed, ctypes.byref(m)))
        assert m.value == 0
    # ... and where t_s_camera divides the clipped penetration by 0.05 (round 5): every float a penetration can be, exhaustively
    m = ctypes.c_int64(-1)
    assert 0 == (capi.test_lib().tg_selftest_penetration_division(ctypes.byref(m)))
    assert m.value == 0


def test_raster_edge_exclusion_never_hides_a_coverable_pixel():
    """The renders skip a record for a block / cell of pixels that its TRIANGLE provably cannot cover (edges_exclude_rect).
    2^24 pseudo-random triangle x rectangle cases per seed - image-sized, slivers, coordinates up to 1e4, vertices on pixel centres,
    heightfield-sized - with every pixel centre of the rectangle put through the pixel loops' own expressions: the rule never excludes a
    rectangle that holds a pixel whose three edge functions share a sign, and it does exclude most of the rectangles that hold none."""
    import ctypes
    from tactile_gym_amd import _capi as capi
    for seed in (3, 20260928):
        out = (ctypes.c_int64 * 6)(-1, -1, -1, -1, -1, -1)
        assert 0 == (capi.test_lib().tg_selftest_edge_exclusion(1 << 24, seed, out))
        violations, excluded, empty = out[0], out[1], out[2]
        assert violations == 0, (seed, violations)
        assert empty > (1 << 22) and excluded > 0.9 * empty, (seed, excluded, empty)      # the rule is not vacuous: it finds >= 90 % of the empty ones
        # the converse rule (round 5, edges_cover_rect): a rectangle called "wholly inside the triangle" never holds a pixel that fails the pixel
        # loops' coverage predicate, and the rule finds most of the rectangles whose every pixel passes
        cover_violations, called_covered, fully_covered = out[3], out[4], out[5]
        assert cover_violations == 0, (seed, cover_violations)
        assert fully_covered > 10000 and called_covered > 0.9 * fully_covered, (seed, called_covered, fully_covered)


def _run_bench(cmd, root, env, tag):
    """One bench.py child; on failure its whole stderr is kept under gpurun_out/ (a gpurun call brings it back).  That is how the cause of
    a rare SIGABRT of one-rank RCCL runs was found: the process group's watchdog thread polled an event of the caller's stream while
    tg_step was capturing its step graph on that stream (hipErrorCapturedEvent); the graph is now captured on a stream of its own."""
    import os, subprocess
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    if out.returncode != 0:
        open(os.path.join(root, "gpurun_out", f"bench_test_failure_{tag}.err"), "w").write(out.stderr)
    open(os.path.join(root, "gpurun_out", f"bench_test_last_{tag}.out"), "w").write(out.stdout)     # what a failed assertion below was looking at
    return out


@pytest.mark.parametrize("env_id,size,port,transport,payload", [("edge_follow-v0", 128, 29541, "collective", "auto"), ("object_push-v0", 128, 29542, "collective", "auto"),
                                                                ("object_balance-v0", 256, 29543, "collective", "auto"), ("edge_follow-v0", 128, 29544, "auto", "auto"),
                                                                ("object_push-v0", 128, 29545, "ipc", "tiles"), ("edge_follow-v0", 128, 29546, "collective", "tiles")])
def test_bench_launches_under_torchrun_on_the_rccl_gather_path(env_id, size, port, transport, payload):
    """The driver's multi-GPU command line with one rank: torch.distributed.run -> RCCL process group -> ShardedVecEnv's gather
    (forced although world_size is 1) -> one JSON line.  Guards the N > 1 launch path on a 1-GPU box for BASELINE configs 2 (edge_follow),
    4 (object_push: tactile_and_feature, the extended_feature block rides in the same packed message) and 5 (object_balance 256 x 256:
    a 16.8 MB payload per step per rank)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TG_BENCH_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "10", "--num-envs", "256",
           "--env", env_id, "--image-size", str(size), "--no-cpu-baseline", "--no-literal", "--transport", transport, "--payload", payload]
    out = _run_bench(cmd, root, env, f"{port}")
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert [l for l in out.stdout.splitlines() if l.strip()] == lines, "stdout must carry the JSON line only (RCCL's version banner goes to stderr)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["value"] > 0 and d["config"]["total_envs"] == 256
    assert d["roofline"]["frac"] > 0 and env_id in d["config"]["workload"]
    # the exchange really ran, on the transport asked for, and what rank 0 was handed is what the rank rendered
    ex = d["exchange"]
    assert d["rccl_ranks"] == 1 and ex["verified"] is True
    if transport == "auto":   # the faster verified one of the probe (with one rank and 256 envs the two are within a few per cent of each other)
        pr = ex["probe"]
        assert pr["ipc + tiles"]["verified"] is True and pr["collective + interior"]["verified"] is True
        faster = min(("ipc + tiles", "collective + interior"), key=lambda k: pr[k]["ms_per_step"])
        assert pr["chosen"] == f'{ex["transport"]} + {ex["payload"]}' and pr[pr["chosen"]]["ms_per_step"] <= 1.02 * pr[faster]["ms_per_step"]
    else:
        assert ex["transport"] == transport
        assert ex["payload"] == ("tiles" if payload == "tiles" else ("interior" if env_id != "object_push-v0" else "full"))
    if ex["payload"] == "tiles":
        assert ex["message_bytes_last"][0] < ex["message_bytes_capacity"]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run (the
    driver's N = 1 command was a plain `python3 bench.py`).  One GPU here, so the path is taken with one rank (TG_BENCH_SPAWN=1)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TG_BENCH_SPAWN="1", TG_BENCH_FORCE_COLLECTIVE="1")
    out = _run_bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5", "--num-envs", "128", "--no-cpu-baseline",
                      "--no-literal", "--no-companions"], root, env, "spawn")
    assert out.returncode == 0, out.stderr[-12000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert [l for l in out.stdout.splitlines() if l.strip()] == lines, "stdout must carry the JSON line only (RCCL's version banner goes to stderr)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["exchange"]["verified"] is True and d["value"] > 0


def test_bench_falls_back_to_the_rccl_gather_when_the_ipc_exchange_fails():
    """An ipc exchange that reports an error (flag timeouts, a wrong batch; injected here) does not cost the run its number: every rank
    agrees on the failure, the ipc slots are closed and the K steps are timed again through the RCCL gather; the line says so."""
    import json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TG_BENCH_SPAWN="1", TG_BENCH_FORCE_COLLECTIVE="1", TG_BENCH_INJECT_EXCHANGE_FAULT="1")
    out = _run_bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5", "--num-envs", "128", "--no-cpu-baseline",
                      "--no-literal", "--no-companions"], root, env, "fallback")
    assert out.returncode == 0, out.stderr[-12000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    ex = d["exchange"]
    assert ex["transport"] == "collective" and ex["verified"] is True and ex["fallback"]["from"].startswith("ipc") and "injected" in ex["fallback"]["why"]
    assert d["value"] > 0 and d["rccl_ranks"] == 1


def test_sample_actions_is_a_counter_based_uniform_box_sample(edge_modes):
    """tg_sample_actions = action_space.sample() for the batch: float32 in [min_action, max_action), a function of (seed, counter,
    element) only, flat histogram."""
    import torch
    import tactile_gym_amd as tg
    n = 4096
    venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=10, image_size=[128, 128], env_modes=edge_modes, seed=5, obs_mode="torch")
    a = torch.empty(n, 2, device="cuda", dtype=torch.float32)
    b = torch.empty_like(a)
    venv.sample_actions(a, 7, 1); venv.sample_actions(b, 7, 1); venv.sync()
    assert torch.equal(a, b)
    venv.sample_actions(b, 7, 2); venv.sync()
    assert not torch.equal(a, b)
    venv.sample_actions(b, 8, 1); venv.sync()
    assert not torch.equal(a, b)
    x = a.cpu().numpy().ravel()
    assert x.min() >= -0.25 and x.max() < 0.25 and abs(x.mean()) < 0.01 and abs(x.std() - 0.5 / np.sqrt(12.0)) < 0.005
    hist = np.histogram(x, bins=8, range=(-0.25, 0.25))[0]
    assert hist.min() > 0.85 * x.size / 8 and hist.max() < 1.15 * x.size / 8
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.05
    venv.reset(); venv.step_async(a); venv.step_wait()   # and a step takes them in place
    venv.close()


# ---------------------------------------------------------------------------------------------------------------- boundary
@pytest.mark.gpu
def test_single_env_gym_surface_matches_oracle(edge_modes):
    """BASELINE config 1's workload (edge_follow-v0, UR5 + TacTip, ONE env, random actions) on the HIP path through the reference's
    gym.Env surface: tg.make(id, max_steps, image_size, env_modes, show_gui, show_tactile) as sb3_helpers/rl_utils.py:49-56 calls it,
    seed / reset / step (old 4-tuple) / render / close, against the oracle env: observations bit-exact, reward 1e-6, done exact, and
    every returned array owned by the caller (a kept observation is not overwritten by later steps)."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleEdgeFollowEnv
    env = tg.make("edge_follow-v0", max_steps=12, image_size=[128, 128], env_modes=edge_modes, show_gui=False, show_tactile=False)
    assert env.seed(7) == [7]
    assert env.action_space.shape == (2,) and env.observation_space["tactile"].shape == (128, 128, 1)
    assert float(env.min_action) == -0.25 and float(env.max_action) == 0.25
    ora = OracleEdgeFollowEnv(seed=7, max_steps=12, image_size=(128, 128), env_modes=edge_modes)
    obs, ref = env.reset(), ora.reset()
    assert set(obs) == {"tactile"} and obs["tactile"].dtype == np.uint8 and obs["tactile"].shape == (128, 128, 1)
    assert np.array_equal(obs["tactile"], ref["tactile"])
    kept, kept_copy = obs["tactile"], obs["tactile"].copy()
    rng = np.random.default_rng(1)
    done = False
    for step in range(12):
        a = rng.uniform(-0.25, 0.25, 2).astype(np.float32)
        obs, rew, done, info = env.step(a)
        ro, rr, rd, _ = ora.step(a)
        assert isinstance(rew, float) and isinstance(done, bool) and info == {}
        assert np.array_equal(obs["tactile"], ro["tactile"]) and abs(rew - rr) < 1e-6 and done == rd, step
    assert done                                                   # max_steps reached: the caller resets (no auto-reset in a gym.Env)
    assert np.array_equal(kept, kept_copy)                        # the first observation is still what it was
    frame = env.render(mode="rgb_array")
    assert frame.dtype == np.uint8 and frame.ndim == 3 and frame.shape[2] == 3
    obs2 = env.reset()
    assert np.array_equal(obs2["tactile"], ora.reset()["tactile"])
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arm", ["ur5", "mg400"])
def test_edge_follow_oracle_observation_vector(edge_modes, arm):
    """observation_mode "oracle" of edge_follow (edge_follow_env.py:454-476): [tcp pos (work frame) 3, tcp linear velocity (work frame) 3,
    goal pos (work frame) 3, edge angle] float32 [N, 10], against the oracle's restatement after reset and after steps."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleEdgeFollowEnv
    modes = dict(edge_modes, observation_mode="oracle", arm_type=arm)
    n = 4
    venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=50, image_size=[64, 64], env_modes=modes, seed=3, auto_reset=False)
    assert venv.observation_space["oracle"].shape == (10,)
    oracles = [OracleEdgeFollowEnv(seed=3 + i, max_steps=50, image_size=(64, 64), env_modes=modes) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    rng = np.random.default_rng(2)
    for step in range(4):
        for i in range(n):
            assert obs["oracle"].dtype == np.float32 and np.abs(obs["oracle"][i] - ref[i]["oracle"]).max() < 2e-6, (step, i)
        a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        obs, _, _, _ = venv.step(a)
        ref = [o.step(a[i])[0] for i, o in enumerate(oracles)]
    venv.close()


@pytest.mark.gpu
def test_fresh_action_tensor_every_step_and_non_default_stream(edge_modes):
    """The documented device-resident usage env.step(policy(obs)) hands over a NEW action tensor on most steps, possibly on a
    non-default torch stream: results must equal the host-action path step for step (the step graph is not re-captured per pointer,
    the env runs on torch's current stream so reads / writes are ordered with the policy's kernels)."""
    import torch
    import tactile_gym_amd as tg
    n, steps = 256, 12
    a = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=edge_modes, seed=13)
    b = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=edge_modes, seed=13, obs_mode="torch")
    oa = a.reset()
    side = torch.cuda.Stream()
    rng = np.random.default_rng(4)
    with torch.cuda.stream(side):
        ob = b.reset()
        assert np.array_equal(oa["tactile"], ob["tactile"].cpu().numpy())
        for step in range(steps):
            act = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
            oa, ra, da, _ = a.step(act)
            t = torch.from_numpy(act).to("cuda", non_blocking=True) * 1.0      # a fresh tensor, produced on the side stream
            ob, rb, db, _ = b.step(t)
            img = ob["tactile"].clone()                                          # consumed on the same stream, no host sync in between
            side.synchronize()
            assert np.array_equal(oa["tactile"], img.cpu().numpy()), step
            assert np.array_equal(ra, rb) and np.array_equal(da, db)
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("stim", ["long_edge", "cube", "pole", "sphere"])
def test_backface_cull_leaves_images_bit_exact(stim):
    """The raster drops back faces of closed, outward-wound stimuli that lie wholly beyond the near plane (tg_raster.hip:back_facing);
    the oracle rasterises every triangle.  512 random poses per stimulus - grazing views, the stimulus poking through the skin, and
    poses that put part of it in front of the near plane or around the camera (where the cull must switch itself off): every image
    equals the oracle's bit for bit."""
    import os
    from oracle import minibullet as mb
    from tactile_gym_amd import hip_ops
    from tactile_gym_amd.robot_model import ASSETS, MeshDesc, SensorDesc
    if stim == "long_edge":
        z = np.load(os.path.join(ASSETS, "stimuli", "long_edge.npz"))
    else:
        z = np.load(os.path.join(ASSETS, "objects", f"{stim}.npz"))
    verts, tris = z["verts"].astype(np.float32), z["tris"].astype(np.int32)
    if stim == "sphere":
        verts = verts * np.float32(4.0)                     # a 1 cm marble: a few hundred pixels instead of a dozen
    sensor = SensorDesc("tactip", "standard", [128, 128])
    mesh = MeshDesc(verts, tris)
    rng = np.random.default_rng(17)
    n = 512
    xf = np.zeros((n, 12), np.float32)
    centre = 0.5 * (verts.min(0) + verts.max(0))
    for i in range(n):
        a = rng.normal(size=(3, 3))
        q, _ = np.linalg.qr(a)
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        dist = rng.uniform(-0.01, 0.075) if i % 4 == 0 else rng.uniform(0.04, 0.07)   # every 4th pose: near plane (0.01 m) or closer
        t = np.array([rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), -dist]) - q @ centre
        xf[i, :9], xf[i, 9:] = q.reshape(9), t
    got = hip_ops.render_tactile(sensor, mesh, xf)
    nonblank = 0
    for i in range(n):
        cur = sensor.nodef_dep.copy()
        mb.render_depth(verts, tris, xf[i], sensor.cam["fov"], sensor.cam["near"], sensor.cam["far"], 128, 128, cur)
        ref = mb.t_s_camera(cur, sensor.nodef_dep, sensor.nodef_gray, sensor.border_mask)
        assert np.array_equal(got[i], ref), (stim, i, int((got[i] != ref).sum()))
        nonblank += int((ref[sensor.border_mask == 0] > 0).any())
    assert nonblank > n // 3


@pytest.mark.gpu
@pytest.mark.parametrize("size,scale", [(64, 4.0), (256, 1.0), (256, 4.0), (128, 16.0)])
def test_scatter_raster_marble_sizes_and_large_triangles(size, scale):
    """k_render_scatter (triangle-parallel, LDS z-buffer of ds_min keys) on the 960-triangle marble at the other image sizes (64 x 64:
    one 64 x 64 tile; 256 x 256: four 128 x 128 tiles) and blown up until its triangles take the queued whole-wavefront fill: bit-exact
    against the oracle's sequential raster, poses through and in front of the near plane included."""
    import os
    from oracle import minibullet as mb
    from tactile_gym_amd import hip_ops
    from tactile_gym_amd.robot_model import ASSETS, MeshDesc, SensorDesc
    z = np.load(os.path.join(ASSETS, "objects", "sphere.npz"))
    verts, tris = z["verts"].astype(np.float32) * np.float32(scale), z["tris"].astype(np.int32)
    sensor = SensorDesc("tactip", "standard", [size, size])
    mesh = MeshDesc(verts, tris)
    rng = np.random.default_rng(size + int(scale))
    n = 96
    xf = np.zeros((n, 12), np.float32)
    for i in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        dist = rng.uniform(-0.01, 0.075) if i % 4 == 0 else rng.uniform(0.035, 0.07)
        xf[i, :9], xf[i, 9:] = q.reshape(9), np.array([rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), -dist])
    got = hip_ops.render_tactile(sensor, mesh, xf)
    nonblank = 0
    for i in range(n):
        cur = sensor.nodef_dep.copy()
        mb.render_depth(verts, tris, xf[i], sensor.cam["fov"], sensor.cam["near"], sensor.cam["far"], size, size, cur)
        ref = mb.t_s_camera(cur, sensor.nodef_dep, sensor.nodef_gray, sensor.border_mask)
        assert np.array_equal(got[i], ref), (size, scale, i, int((got[i] != ref).sum()))
        nonblank += int((ref[sensor.border_mask == 0] > 0).any())
    assert nonblank > n // 4


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,modes_name,mode", [("surface_follow-v1", "SURF", "tactile_and_feature"), ("surface_follow-v2", "VERT", "tactile_and_feature"),
                                                    ("edge_follow-v0", "EDGE", "oracle"), ("object_push-v0", "PUSH", "oracle"), ("surface_follow-v0", "SURF", "oracle")])
def test_terminal_observation_is_the_last_step_before_the_reset(env_id, modes_name, mode, edge_modes):
    """info["terminal_observation"] of an auto-reset env (SB3 convention) must be what a caller-reset env returns at that step: the
    extended_feature of surface_follow-v1 / -v2 and the oracle vector (both written by the step kernels before the reset, kept in
    device-side terminal buffers).  An auto-reset env against a twin without auto-reset, same seeds and actions, max_steps = 4."""
    import bench
    import tactile_gym_amd as tg
    modes = dict({"EDGE": edge_modes, "SURF": bench.SURF_MODES, "VERT": bench.VERT_MODES, "PUSH": bench.PUSH_MODES}[modes_name], observation_mode=mode)
    n = 8
    a_env = tg.make_vec(env_id, num_envs=n, max_steps=4, image_size=[64, 64], env_modes=modes, seed=21, auto_reset=True)
    b_env = tg.make_vec(env_id, num_envs=n, max_steps=4, image_size=[64, 64], env_modes=modes, seed=21, auto_reset=False)
    oa, ob = a_env.reset(), b_env.reset()
    key = "oracle" if mode == "oracle" else "extended_feature"
    assert np.array_equal(oa[key], ob[key])
    rng = np.random.default_rng(8)
    for step in range(4):
        act = rng.uniform(-0.25, 0.25, size=(n, a_env.act_dim)).astype(np.float32)
        oa, ra, da, infos = a_env.step(act)
        ob, rb, db, _ = b_env.step(act)
        assert np.array_equal(da, db) and np.allclose(ra, rb)
        for i in range(n):
            if da[i]:
                term = infos[i]["terminal_observation"]
                assert key in term and np.array_equal(term[key], ob[key][i]), (step, i)
                assert not np.array_equal(oa[key][i], ob[key][i])          # the returned observation is the post-reset one
            else:
                assert np.array_equal(oa[key][i], ob[key][i]), (step, i)
    assert da.all()
    if key == "extended_feature" and hasattr(a_env, "feature_host"):
        assert np.abs(a_env.feature_host() - oa[key]).max() < 2e-6    # device route == state read-back route
    a_env.close(); b_env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sensor,size", [("tactip", 128), ("digitac", 128), ("tactip", 64)])
def test_interior_payload_round_trip_is_bit_exact(edge_modes, sensor, size):
    """tg_pack_interior / tg_unpack_interior (the multi-GPU gather's interior-only payload): packing the pixels inside the border mask and
    restoring the constant ring gives back the observation batch bit for bit; also through ShardedVecEnv with one rank (forced)."""
    import warnings
    import torch
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import TorchShard
    modes = dict(edge_modes, tactile_sensor_name=sensor)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        venv = tg.make_vec("edge_follow-v0", num_envs=96, max_steps=50, image_size=[size, size], env_modes=modes, seed=4, obs_mode="torch")
    shard = TorchShard(venv)
    shard.reset()
    a = torch.zeros((96, 2), device="cuda")
    for _ in range(3):
        obs, _, _, _ = shard.step(a.uniform_(-0.25, 0.25))
    idx, template = shard.border_info()
    k = idx.numel()
    assert 0 < k <= size * size            # (DigiTac images have no border ring: k = H W, the payload is then the full image)
    packed = torch.empty((96, k), dtype=torch.uint8, device="cuda")
    shard.pack_interior(packed)
    full = torch.full((96, size * size), 7, dtype=torch.uint8, device="cuda")
    shard.unpack_interior(packed, full)
    torch.cuda.synchronize()
    ref = obs["tactile"].reshape(96, -1)
    assert torch.equal(packed, ref[:, idx]) and torch.equal(full, ref)
    assert (ref[:, idx] > 0).any()          # the interiors carry an imprint, not only zeros
    venv.close()


@pytest.mark.parametrize("env_id,overrides,act_dim,size", [
    ("edge_follow-v0", dict(movement_mode="xyRz", noise_mode="fixed_height", tactile_sensor_name="digit"), 3, 128),
    ("edge_follow-v0", dict(movement_mode="xyzRz", tactile_sensor_name="digitac"), 4, 128),
    ("edge_follow-v0", dict(movement_mode="xyz", arm_type="mg400", tactile_sensor_name="digit"), 3, 128),
    ("edge_follow-v0", dict(movement_mode="xy", arm_type="mg400", tactile_sensor_name="digitac", noise_mode="fixed_height"), 2, 64),
    ("edge_follow-v0", dict(movement_mode="xyz", noise_mode="fixed_height"), 3, 256),
    ("object_balance-v0", dict(movement_mode="xyz"), 3, 64),
    ("object_balance-v0", dict(movement_mode="RxRy", rand_gravity=False), 2, 64),
    ("object_balance-v0", dict(movement_mode="xyRxRy", rand_embed_dist=False), 4, 128),
])
def test_mode_matrix_matches_oracle(env_id, overrides, act_dim, size, edge_modes):
    """The env_modes the other tests leave out (edge_follow movement modes x sensors x arms, fixed_height; object_balance's velocity-control
    movement modes and its randomisation switches): reset + 4 random-action steps, 4 envs vs 4 oracle envs."""
    import warnings
    import tactile_gym_amd as tg
    from oracle import ref_env
    edge = env_id.startswith("edge")
    modes = dict(edge_modes if edge else BAL_MODES, **overrides)
    Oracle = ref_env.OracleEdgeFollowEnv if edge else ref_env.OracleObjectBalanceEnv
    n = 4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")     # digit 64 x 64 reference images are outside the pinned set (A14b); HIP and oracle share them
        venv = tg.make_vec(env_id, num_envs=n, max_steps=50, image_size=[size, size], env_modes=modes, seed=301, auto_reset=False)
        oracles = [Oracle(seed=301 + i, max_steps=50, image_size=(size, size), env_modes=modes) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    st = venv.get_state()
    for i, o in enumerate(oracles):
        assert st["reset_ticks"][i] == o.reset_ticks
        assert np.abs(st["q"][i] - o.arm.q).max() < 1e-7
        assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 3
    rng = np.random.default_rng(302)
    for step in range(4):
        a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-7, (step, i)
            if not edge:
                pos, R = o.body_pose()
                assert np.abs(st["body_pos"][i] - pos).max() < 1e-7 and np.abs(st["body_rot"][i] - R).max() < 1e-7, (step, i)
            assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd, (step, i)
            assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 3, (step, i)
    venv.close()


@pytest.mark.parametrize("env_id,arm,sensor,full", [("edge_follow-v0", "mg400", "tactip", False), ("edge_follow-v0", "ur5", "tactip", True),
                                                    ("surface_follow-v2", "mg400", "tactip", False)])
def test_arm_wave_mapping_matches_oracle(env_id, arm, sensor, full):
    """k_step_arm_wave (contact_mapping="wave" for the contact-free arm tasks: every tick a full tick on the env's own wavefront - lane-
    parallel dynamics, the motor pass as one linear map) against the oracle, like the lane kernel: joints 1e-9, images bit-exact, dones."""
    import tactile_gym_amd as tg
    from oracle import ref_env
    from tactile_gym_amd import registry
    cls = registry._resolve(registry.spec(env_id))
    modes = dict(cls.default_env_modes, arm_type=arm, tactile_sensor_name=sensor, observation_mode="tactile")
    ocls = {"edge_follow-v0": ref_env.OracleEdgeFollowEnv, "surface_follow-v2": ref_env.OracleSurfaceFollowVertEnv}[env_id]
    n = 5
    venv = tg.make_vec(env_id, num_envs=n, max_steps=50, image_size=[128, 128], env_modes=modes, seed=61, auto_reset=False, contact_mapping="wave",
                       pgs_full_sweeps=full)
    oracles = [ocls(seed=61 + i, max_steps=50, image_size=(128, 128), env_modes=modes) for i in range(n)]
    obs = venv.reset()
    ref = [o.reset() for o in oracles]
    rng = np.random.default_rng(3)
    for step in range(5):
        a = rng.uniform(-0.25, 0.25, size=(n, venv.act_dim)).astype(np.float32)
        obs, rew, done, _ = venv.step(a)
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-9, (step, i)
            assert abs(rew[i] - rr) < 1e-5 and bool(done[i]) == rd
            assert np.array_equal(obs["tactile"][i], ro["tactile"]), (step, i)
    venv.close()


def test_maximum_batch_size(edge_modes):
    """The largest context tg_create accepts (65 535 envs: the render launch carries the env index in grid.y): reset + 2 steps at 64 x 64,
    the first and the last eight envs equal 8-env contexts with the same seeds (batch-size independence at the edge of the grid), the tile
    payload of the whole batch round-trips, and 65 536 envs are refused with a message."""
    import torch
    import tactile_gym_amd as tg
    from tactile_gym_amd._capi import TactileGymHipError
    from tactile_gym_amd.parallel import TILE_REC, TorchShard
    n = 65535
    with pytest.raises(TactileGymHipError, match="65535"):
        tg.make_vec("edge_follow-v0", num_envs=n + 1, max_steps=10, image_size=[64, 64], env_modes=edge_modes)
    big = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=10, image_size=[64, 64], env_modes=edge_modes, seed=1000, obs_mode="torch", auto_reset=False)
    sh = TorchShard(big)
    sh.reset()
    acts = torch.from_numpy(np.random.default_rng(4).uniform(-0.25, 0.25, size=(2, n, 2)).astype(np.float32)).cuda()
    for k in range(2):
        obs, rew, done, _ = sh.step(acts[k])
    torch.cuda.synchronize()
    for lo in (0, n - 8):
        small = tg.make_vec("edge_follow-v0", num_envs=8, max_steps=10, image_size=[64, 64], env_modes=edge_modes, seed=1000 + lo, obs_mode="torch", auto_reset=False)
        ss = TorchShard(small)
        ss.reset()
        for k in range(2):
            o2, r2, d2, _ = ss.step(acts[k, lo:lo + 8].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(o2["tactile"], obs["tactile"][lo:lo + 8]) and torch.equal(r2, rew[lo:lo + 8]) and torch.equal(d2, done[lo:lo + 8])
        assert np.array_equal(small.get_state()["q"], big.get_state()["q"][lo:lo + 8])
        small.close()
    cap = 16 + TILE_REC * n * 16
    msg = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(4, dtype=torch.int32, device="cuda")
    sh.pack_tiles(msg.data_ptr(), counters)
    out = torch.zeros((n, 64 * 64), dtype=torch.uint8, device="cuda")
    sh.unpack_tiles(msg.data_ptr(), n, out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out.reshape(obs["tactile"].shape), obs["tactile"]) and 0 < int(msg[:4].view(torch.int32).item()) < n * 16
    big.close()


BALL_MODES = dict(BAL_MODES, object_mode="ball_on_plate")


@pytest.mark.parametrize("size,movement", [(128, "xy"), (64, "xyRxRy")])
def test_object_balance_ball_on_plate_matches_oracle(size, movement):
    """object_balance-v0 with object_mode "ball_on_plate" (object_balance_env.py:187-199, 241-260, 350-353, 393-401): the round plate on the
    point-to-point constraint and a ball rolling on it (sim_tick_body_ball == mb_step_body_ball, PARITY A39).  Two consecutive episodes -
    the second reset's blocking move runs with the ball wherever the first episode left it - 6 envs vs 6 oracle envs: joint angles
    1e-8 rad, plate pose and ball position 1e-7 (same recurrence in the Delassus form), reward / done exact, images <= 3 pixels."""
    import tactile_gym_amd as tg
    from oracle.ref_env import OracleObjectBalanceEnv
    n, steps = 6, 12
    modes = dict(BALL_MODES, movement_mode=movement)
    act_dim = {"xy": 2, "xyRxRy": 4}[movement]
    venv = tg.make_vec("object_balance-v0", num_envs=n, max_steps=steps, image_size=[size, size], env_modes=modes, seed=171, auto_reset=False)
    oracles = [OracleObjectBalanceEnv(seed=171 + i, max_steps=steps, image_size=(size, size), env_modes=modes) for i in range(n)]
    rng = np.random.default_rng(172)
    touched = 0
    for episode in range(2):
        obs = venv.reset()
        ref = [o.reset() for o in oracles]
        st = venv.get_state()
        for i, o in enumerate(oracles):
            assert st["gravity_z"][i] == o.gravity and st["embed_dist"][i] == o.embed_dist
            assert st["reset_ticks"][i] == o.reset_ticks
            assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8
            assert np.abs(st["body_pos"][i] - o.body_pose()[0]).max() < 1e-12
            assert np.array_equal(st["ball_pos"][i], np.array(o.ball.pos[:]))
            assert int((obs["tactile"][i] != ref[i]["tactile"]).sum()) <= 3
        for step in range(steps):
            a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
            obs, rew, done, _ = venv.step(a)
            st = venv.get_state()
            for i, o in enumerate(oracles):
                ro, rr, rd, _ = o.step(a[i])
                pos, R = o.body_pose()
                assert np.abs(st["q"][i] - o.arm.q).max() < 1e-8, (episode, step, i)
                assert np.abs(st["body_pos"][i] - pos).max() < 1e-7 and np.abs(st["body_rot"][i] - R).max() < 1e-7, (episode, step, i)
                assert np.abs(st["ball_pos"][i] - np.array(o.ball.pos[:])).max() < 1e-7, (episode, step, i)
                assert np.abs(st["ball_linvel"][i] - np.array(o.ball.linvel[:])).max() < 1e-6, (episode, step, i)
                assert abs(st["ball_impulse"][i] - o.ball.normal_impulse) < 1e-9, (episode, step, i)
                assert (st["ball_impulse"][i] > 0) == (o.ball.normal_impulse > 0)
                touched += int(o.ball.normal_impulse > 0)
                assert rew[i] == rr and bool(done[i]) == rd
                assert int((obs["tactile"][i] != ro["tactile"]).sum()) <= 3, (episode, step, i)
        assert done.all()
    assert touched > n * steps          # the ball was on the plate for most of the run
    venv.close()


def test_object_balance_ball_on_plate_refuses_wave():
    import tactile_gym_amd as tg
    with pytest.raises(ValueError):
        tg.make_vec("object_balance-v0", num_envs=2, max_steps=4, image_size=[64, 64], env_modes=BALL_MODES, seed=1, contact_mapping="wave")
