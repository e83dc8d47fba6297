"""GPU parity of the scene camera (the "visual" / "visuotactile" observation modes and render()), through the C ABI.

The HIP scene raster (tg_scene.hip) against oracle/minibullet.c mb_render_scene on the device's own joint angles / body pose:
BIT-EXACT rgb on every pixel (the eye<-frame transforms are rounded once to float32 from double-precision forward kinematics on both
sides; every float operation after that is written once and evaluated in the same order).  Parity with upstream's pixels is unpinned
(PARITY_ASSUMPTIONS A31-A33): upstream's renderer is the GL driver's.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spheres(env, st, i):
    """The env's translucent visuals from the device state, restated from the reference (what oracle/ref_env.py scene_spheres() lists for
    its own state): [(world centre, radius, rgb, alpha)]."""
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd import pb_math as pbm
    kind, cfg = env._cfg.env_kind, env._cfg
    red = (255.0, 0.0, 0.0)
    return [(st["tcp_pos"][i], 0.001, (229.5, 0.0, 51.0), 0.5)] + _task_spheres(env, st, i, kind, cfg, red)     # slot 0: the arm's TCP marker


def _task_spheres(env, st, i, kind, cfg, red):
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd import pb_math as pbm
    if kind == capi.ENV_EDGE_FOLLOW:                      # edge_follow_env.py:268-283: the edge's far end, at edge height
        a = st["edge_ang"][i]
        return [(np.array([cfg.stim_pos[0] + cfg.edge_len * np.cos(a), cfg.stim_pos[1] + cfg.edge_len * np.sin(a), cfg.stim_pos[2] + cfg.edge_height]), 0.01, red, 0.5)]
    if kind == capi.ENV_SURFACE_FOLLOW_AUTO:              # base_surface_env.py:560-576
        return [(st["goal_pos"][i], 0.01, red, 0.5)]
    if kind == capi.ENV_OBJECT_PUSH:                      # object_push_env.py:239-250, 276-282, 345-366
        wp = np.array([cfg.workframe_pos[k] for k in range(3)])
        wR = pbm.mat_from_quat(pbm.quat_from_euler(np.array([cfg.workframe_rpy[k] for k in range(3)])))
        gid, out = int(st["goal_id"][i]), []
        for k in range(cfg.traj_n_points):
            rgb = (0.0, 0.0, 255.0) if k == gid else (red if k < gid else (0.0, 255.0, 0.0))
            out.append((wp + wR @ np.array([st["traj"][i][0, k], st["traj"][i][1, k], 0.0]), 0.01, rgb, 0.5))
        return out
    if kind == capi.ENV_OBJECT_ROLL:                      # object_roll_env.py:258-284: the goal rides in the TCP frame
        R = pbm.mat_from_quat(pbm.quat_from_euler(st["tcp_rpy"][i]))
        return [(st["tcp_pos"][i] + R @ st["goal_pos"][i], 0.0025, red, 0.5)]
    return []                                             # object_balance: visualise_goal = False


def _oracle_images(env, envs):
    """mb_render_scene (+ mb_blend_spheres) for the listed envs of a TactileVecEnv at its current device state."""
    from oracle import minibullet as mb
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd.robot_model import BACKGROUND, LIGHT_DIR, load_tgmodel
    st = env.get_state()
    sc = env._scene
    arm = mb.Arm(load_tgmodel(env.env_modes["arm_type"], env._sensor.t_s_type, env._sensor.t_s_name))
    cam = env._scene_spec["camera"]
    view = mb.scene_view_matrix(cam[0], cam[1], cam[2], cam[3])
    kind = env._cfg.env_kind
    out = []
    for i in envs:
        frames = [(np.eye(3), np.zeros(3))] + arm.link_poses(st["q"][i])
        if kind == capi.ENV_SURFACE_FOLLOW_AUTO:        # the env's own heightfield, blue, in the frame of the surface
            from oracle.ref_env import heightfield_mesh
            from tactile_gym_amd import pb_math as pbm
            hv, ht = heightfield_mesh(st["heights"][i], env._cfg.surf_grid_scale, st["surf_zoff"][i])
            R = pbm.mat_from_quat(pbm.quat_from_euler(np.array([0.0, -np.pi / 2, 0.0]))) if env._cfg.noise_mode == capi.SNOISE["vertical_simplex"] else np.eye(3)
            frames.append((R, np.array([env._cfg.stim_pos[k] for k in range(3)])))
            verts = np.concatenate([sc.verts, hv]); tris = np.concatenate([sc.tris, ht + len(sc.verts)])
            tf = np.concatenate([sc.tri_frame, np.full(len(ht), len(frames) - 1, np.uint8)])
            rgb = np.concatenate([sc.tri_rgb, np.tile(np.array([0, 0, 255], np.uint8), (len(ht), 1))])
            out.append(mb.render_scene(verts, tris, tf, rgb, frames, view, LIGHT_DIR, cam[4], cam[5], cam[6], env.W, env.H, BACKGROUND,
                                       spheres=_spheres(env, st, i)))
            continue
        if kind == capi.ENV_EDGE_FOLLOW:
            a = st["edge_ang"][i]
            R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
            frames.append((R, np.array([env._cfg.stim_pos[k] for k in range(3)])))
        else:
            scale = st["obj_mass"][i] / env._cfg.roll_radius if kind == capi.ENV_OBJECT_ROLL else 1.0
            frames.append((st["body_rot"][i] * scale, st["body_pos"][i]))
        out.append(mb.render_scene(sc.verts, sc.tris, sc.tri_frame, sc.tri_rgb, frames, view, LIGHT_DIR, cam[4], cam[5], cam[6], env.W, env.H, BACKGROUND,
                                   spheres=_spheres(env, st, i)))
    return np.stack(out)


CASES = [("edge_follow-v0", "ur5", "tactip", 128, "visuotactile"), ("edge_follow-v0", "mg400", "digitac", 64, "visual"),
         ("edge_follow-v0", "ur5", "digit", 256, "visual"), ("object_push-v0", "ur5", "tactip", 128, "visuotactile_and_feature"),
         ("object_balance-v0", "ur5", "tactip", 256, "visuotactile"), ("object_roll-v0", "ur5", "tactip", 128, "visual"),
         ("surface_follow-v0", "ur5", "digit", 128, "visuotactile"), ("surface_follow-v1", "ur5", "tactip", 64, "visuotactile_and_feature"),
         ("surface_follow-v2", "mg400", "tactip", 128, "visual")]


@pytest.mark.parametrize("env_id,arm,sensor,size,mode", CASES)
def test_visual_observation_bit_exact(env_id, arm, sensor, size, mode):
    import warnings
    from tactile_gym_amd import registry
    cls = registry._resolve(registry.spec(env_id))
    modes = dict(cls.default_env_modes, arm_type=arm, tactile_sensor_name=sensor, observation_mode=mode)
    n = 6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")               # digit / digitac 64 x 64 reference images: STALE_REFERENCE_IMAGES warning
        env = cls.make_vec(n, max_steps=6, image_size=(size, size), env_modes=modes, seed=11, auto_reset=True)
    obs = env.reset()
    assert obs["visual"].shape == (n, size, size, 3) and obs["visual"].dtype == np.uint8
    assert ("tactile" in obs) == ("tactile" in mode) and ("extended_feature" in obs) == ("feature" in mode)
    assert np.array_equal(obs["visual"], _oracle_images(env, range(n)))
    rng = np.random.default_rng(5)
    for step in range(7):
        prev = _oracle_images(env, range(n)) if step == 5 else None
        a = rng.uniform(-0.25, 0.25, (n, env.act_dim)).astype(np.float32)
        obs, rew, done, infos = env.step(a)
        assert np.array_equal(obs["visual"], _oracle_images(env, range(n))), step
        if step == 5:                                   # max_steps = 6: every env finished, was reset, and its last image is kept
            assert done.all()
            term = np.stack([infos[i]["terminal_observation"]["visual"] for i in range(n)])
            assert term.shape == obs["visual"].shape and not np.array_equal(term, obs["visual"])
            # the terminal image is the scene after this step's physics, before the reset: the arm moved by one action from `prev`
            assert (term != prev).any() and (term != prev).mean() < 0.2
    frames = env.get_images()
    assert frames[0].shape == (size, 2 * size, 3) and np.array_equal(frames[0][:, :size], obs["visual"][0])
    # the translucent visuals are really in the picture: without them the oracle's image differs (except object_balance, which has none)
    from oracle import minibullet as mb
    keep, mb.lib().mb_blend_spheres = mb.lib().mb_blend_spheres, (lambda *a: None)
    try:
        bare = _oracle_images(env, range(n))
    finally:
        mb.lib().mb_blend_spheres = keep
    assert (bare != obs["visual"]).any() or env_id == "object_balance-v0", env_id       # (balance: only the TCP marker, 0.1 - 0.2 pixels across)
    env.close()


def test_render_frame_in_tactile_mode_and_single_env(edge_modes):
    """render() in a non-visual observation mode: [H, 2W, 3] (base_tactile_env.py:284-303), scene drawn on demand, stepping unchanged."""
    import tactile_gym_amd as tg
    env = tg.make("edge_follow-v0", max_steps=20, image_size=[128, 128], env_modes=edge_modes)
    env.seed(3)
    obs = env.reset()
    frame = env.render()
    assert frame.shape == (128, 256, 3) and frame.dtype == np.uint8
    assert np.array_equal(frame[:, 128:, 0], obs["tactile"][..., 0]) and np.array_equal(frame[:, 128:, 1], frame[:, 128:, 2])
    assert np.array_equal(frame[:, :128], _oracle_images(env._vec, [0])[0])
    env.step(np.array([0.2, -0.1], dtype=np.float32))
    frame2 = env.render()
    assert np.array_equal(frame2[:, :128], _oracle_images(env._vec, [0])[0]) and (frame2 != frame).any()
    env.close()


def test_visual_rate_smoke():
    """1024 envs, visual mode: steps run, images differ across envs (random edge angles), zero-copy torch view matches the host copy."""
    import torch
    from tactile_gym_amd import registry
    cls = registry._resolve(registry.spec("edge_follow-v0"))
    modes = dict(cls.default_env_modes, arm_type="ur5", tactile_sensor_name="tactip", observation_mode="visual")
    env = cls.make_vec(1024, max_steps=50, image_size=(128, 128), env_modes=modes, seed=0, obs_mode="torch")
    env.reset()
    a = torch.zeros((1024, env.act_dim), dtype=torch.float32, device="cuda")
    for _ in range(3):
        obs, _, _, _ = env.step(a)
    torch.cuda.synchronize()
    v = obs["visual"]
    assert v.shape == (1024, 128, 128, 3) and v.is_cuda
    host = env.visual_numpy()
    assert np.array_equal(v.cpu().numpy(), host)
    assert len({host[i].tobytes() for i in range(64)}) > 32
    assert np.array_equal(host[[0, 500, 1023]], _oracle_images(env, [0, 500, 1023]))
    env.close()
