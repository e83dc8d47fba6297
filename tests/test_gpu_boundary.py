"""What the reference's callers consume at the VecEnv boundary: per-env episode statistics - the Monitor wrapper
`make_vec_env(..., monitor_dir=...)` puts around every env (sb3_helpers/rl_utils.py:17-30, 59) - and construction through make_vec_env's
`vec_env_cls` hook."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
PUSH = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
            observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")
BAL = dict(movement_mode="xy", control_mode="TCP_velocity_control", object_mode="pole", rand_gravity=True, rand_embed_dist=True,
           observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
SURF = dict(movement_mode="xyzRxRy", control_mode="TCP_velocity_control", noise_mode="simplex", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="digit")


@pytest.mark.parametrize("env_id,cls,modes,act_dim,size", [("edge_follow-v0", "OracleEdgeFollowEnv", EDGE, 2, 64), ("surface_follow-v0", "OracleSurfaceFollowAutoEnv", SURF, 3, 64),
                                                           ("object_balance-v0", "OracleObjectBalanceEnv", BAL, 2, 64), ("object_push-v0", "OracleObjectPushEnv", PUSH, 2, 128)])
def test_episode_info_equals_summed_rewards_of_the_oracle(env_id, cls, modes, act_dim, size):
    """info["episode"] = {"r", "l", "t"} on done (Monitor / VecMonitor semantics), for all four task families (each has its own step-kernel
    epilogue): r = the sum of the rewards of the episode that just ended - equal to the sum of what step() returned (float32 terms added in
    double) and to the oracle env's summed rewards (1e-4: the oracle's terms are doubles) - l = its length; two consecutive episodes per env
    (the second starts from an auto-reset), and a caller's reset() in mid-episode drops the running sum."""
    import tactile_gym_amd as tg
    from oracle import ref_env
    n, max_steps = 6, 5
    venv = tg.make_vec(env_id, num_envs=n, max_steps=max_steps, image_size=[size, size], env_modes=modes, seed=70, auto_reset=True)
    oracles = [getattr(ref_env, cls)(seed=70 + i, max_steps=max_steps, image_size=(size, size), env_modes=modes) for i in range(n)]
    venv.reset()
    for o in oracles:
        o.reset()
    rng = np.random.default_rng(1)
    mine, ref, lens, episodes = np.zeros(n), np.zeros(n), np.zeros(n, int), 0
    for step in range(2 * max_steps + 2):
        a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
        obs, rew, done, infos = venv.step(a)
        for i, o in enumerate(oracles):
            _, rr, rd, _ = o.step(a[i])
            mine[i] += float(rew[i]); ref[i] += rr; lens[i] += 1
            vs_oracle = env_id != "object_push-v0"      # push: the first goal advance of an episode is a knife edge (PARITY A29; followed in
            assert bool(done[i]) == rd or not vs_oracle, (step, i)   # tests/test_gpu_parity.py), so its sums are checked against step()'s rewards only
            if done[i]:
                ep = infos[i]["episode"]
                assert set(ep) == {"r", "l", "t"} and ep["l"] == lens[i] and ep["t"] >= 0.0
                assert abs(ep["r"] - mine[i]) <= 1e-5 * max(1.0, abs(mine[i])), (env_id, step, i, ep, mine[i])
                assert not vs_oracle or abs(ep["r"] - ref[i]) <= 1e-4 * max(1.0, abs(ref[i])), (env_id, step, i, ep, ref[i])
                assert "terminal_observation" in infos[i]
                mine[i] = ref[i] = 0.0; lens[i] = 0; episodes += 1
                o.reset()
            elif rd and not vs_oracle:
                o.reset()
            else:
                assert "episode" not in infos[i]
    assert episodes >= 2 * n
    # a caller's reset() in the middle of an episode starts the sums again
    venv.reset()
    for o in oracles:
        o.reset()
    tot = np.zeros(n)
    for step in range(max_steps):
        a = rng.uniform(-0.25, 0.25, size=(n, act_dim)).astype(np.float32)
        _, rew, done, infos = venv.step(a)
        tot += rew
    assert done.all() and all(abs(infos[i]["episode"]["r"] - tot[i]) <= 1e-5 * max(1.0, abs(tot[i])) and infos[i]["episode"]["l"] == max_steps for i in range(n))
    venv.close()


def test_make_vec_env_with_hipvecenv_equals_make_vec():
    """make_vec_env(env_id, n_envs, seed, env_kwargs, vec_env_cls=tg.HipVecEnv) - the reference's make_training_envs with one token
    changed (sb3_helpers/rl_utils.py:17-30) - builds the same batch as tg.make_vec(..., seed=seed): identical states and images."""
    import tactile_gym_amd as tg
    from test_host_cpu import sb3_like_make_vec_env
    kw = dict(max_steps=50, image_size=[64, 64], env_modes=EDGE)
    a = sb3_like_make_vec_env("edge_follow-v0", n_envs=5, seed=9, env_kwargs=kw, vec_env_cls=tg.HipVecEnv)
    b = tg.make_vec("edge_follow-v0", num_envs=5, seed=9, **kw)
    assert a.num_envs == 5 and type(a) is type(b)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa["tactile"], ob["tactile"])
    acts = np.random.default_rng(0).uniform(-0.25, 0.25, size=(3, 5, 2)).astype(np.float32)
    for k in range(3):
        ra, rb = a.step(acts[k]), b.step(acts[k])
        assert np.array_equal(ra[0]["tactile"], rb[0]["tactile"]) and np.array_equal(ra[1], rb[1])
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa["q"], sb["q"]) and np.array_equal(sa["edge_ang"], sb["edge_ang"])
    a.close(); b.close()


@pytest.mark.parametrize("env_id,modes,size,n", [("edge_follow-v0", EDGE, 128, 96), ("object_push-v0", PUSH, 128, 32), ("object_balance-v0", BAL, 256, 8)])
def test_tile_download_hands_out_the_batch_the_full_copy_hands_out(env_id, modes, size, n):
    """`set_obs_transfer("tiles")`: the numpy observations of a rollout with resets (device pack -> pinned copy of exactly the records ->
    libtg_host.so rebuilding one of five persistent host buffers) equal the plain whole-batch copy byte for byte at every step, the ring
    buffers stay untouched for three further steps, and fewer bytes cross PCIe than the batch holds (edge / push; the pole's plate fills
    the view, so object_balance ships everything plus the record headers)."""
    import tactile_gym_amd as tg
    venv = tg.make_vec(env_id, num_envs=n, max_steps=7, image_size=[size, size], env_modes=modes, seed=21, auto_reset=True)
    venv.set_obs_transfer("tiles")
    rng = np.random.default_rng(2)
    obs = venv.reset()
    held = []
    for step in range(20):
        a = rng.uniform(-0.25, 0.25, size=(n, venv.act_dim)).astype(np.float32)
        obs, rew, done, infos = venv.step(a)
        got = obs["tactile"] if isinstance(obs, dict) else obs
        dl, venv._tile_download = venv._tile_download, None
        full = venv.tactile_numpy()
        venv._tile_download = dl
        assert got.shape == full.shape and np.array_equal(got, full), (env_id, step, int((got != full).sum()))
        held.append((got, full.copy()))
        for g, f in held[-4:]:
            assert np.array_equal(g, f)            # the last four batches handed out are still what they were
        if env_id != "object_balance-v0":
            assert dl.last_bytes < full.size
    assert len({g.ctypes.data for g, _ in held}) == 5          # five ring buffers, reused in turn
    venv.set_obs_transfer("full")
    obs, _, _, _ = venv.step(a)
    venv.close()


@pytest.mark.parametrize("env_id", ["edge_follow-v0", "object_balance-v0"])
def test_step_random_equals_sample_actions_then_step(env_id):
    """tg_step_random (the uniform policy's draw inside the step's own launch - k_step, and since round 5 object_balance's k_step_body_wave, one
    wavefront per env with a two-level election for the counter - device-resident draw counter) = tg_sample_actions(seed, k) followed by
    tg_step on it: same actions, observations, rewards, dones, also across a restart of the counter and with auto-resets in the rollout."""
    import torch
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import TorchShard
    n = 64 if env_id == "edge_follow-v0" else 96          # 96 workgroups: the two-level election (more than 2 x 32)
    modes = EDGE if env_id == "edge_follow-v0" else BAL
    mk = lambda: tg.make_vec(env_id, num_envs=n, max_steps=9, image_size=[128, 128], env_modes=modes, seed=5, auto_reset=True, obs_mode="torch")   # noqa: E731
    a, b = mk(), mk()
    sa, sb = TorchShard(a), TorchShard(b)
    sa.reset(); sb.reset()
    buf = torch.empty(n, a.act_dim, device="cuda")
    draw = 0
    for k in range(25):
        restart = k in (0, 11)
        if k == 11:
            draw = 100                                      # jump the stream: the next draw is 101 on both sides
        oa, ra, da, _ = sa.step_random(77, draw, restart=restart)
        draw += 1
        b.sample_actions(buf, 77, draw)
        ob, rb, db, _ = sb.step(buf)
        torch.cuda.synchronize()
        assert torch.equal(a.actions_torch(), buf), k
        assert torch.equal(oa["tactile"], ob["tactile"]) and torch.equal(ra, rb) and torch.equal(da, db), k
    a.close(); b.close()


def test_terminal_observations_of_finished_envs_equal_the_terminal_batch():
    """numpy VecEnv.step_wait hands info["terminal_observation"] of the finished envs from tg_copy_obs_rows (their images only, round 5) - the same
    bytes as those envs' rows of the whole terminal batch (tg_copy_obs_tactile(terminal=1)), with episodes that end in different steps, in the
    default and in the tile transfer mode; the scene camera's terminal images likewise."""
    import ctypes as C
    import os
    import tactile_gym_amd as tg
    n = 40
    for modes, tiles in ((EDGE, False), (EDGE, True), (EDGE, "cap4"), (dict(EDGE, observation_mode="visuotactile"), False)):
        venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=9, image_size=[128, 128], env_modes=modes, seed=4, auto_reset=True)
        if tiles:
            # round 6: in tile mode the finished envs' ids, episode statistics and terminal images ride along with the observation fetch
            # (tg_pack_done_rows, up to 32 envs per step); "cap4": more envs finish than the block holds - the step falls back to the copies
            if tiles == "cap4":
                os.environ["TG_DONE_ROWS_CAP"] = "4"
            try:
                venv.set_obs_transfer("tiles")
            finally:
                os.environ.pop("TG_DONE_ROWS_CAP", None)
            assert venv._tile_download.done_cap == (4 if tiles == "cap4" else 32)
        venv.reset()
        rng = np.random.default_rng(2)
        m = np.zeros(n, np.uint8); m[::3] = 1
        for k in range(4):
            venv.step(rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32))
        venv.reset(m)                                           # a third of the envs now finish four steps later than the rest
        seen = 0
        for k in range(24):
            obs, rew, done, infos = venv.step(rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32))
            if done.any():
                whole = venv.tactile_numpy(True)
                vis = venv.visual_numpy(True) if "visual" in obs else None
                assert not done.all()
                ret, ln = np.zeros(n, np.float32), np.zeros(n, np.int32)
                venv._L.tg_copy_episode_stats(venv._ctx, ret.ctypes.data_as(C.POINTER(C.c_float)), ln.ctypes.data_as(C.POINTER(C.c_int32)))
                if tiles:
                    rode_along = venv._tile_download.done_rows is not None
                    assert rode_along == (int(done.sum()) <= venv._tile_download.done_cap), (k, int(done.sum()))
                for i in np.flatnonzero(done):
                    t = infos[i]["terminal_observation"]
                    assert infos[i]["episode"]["l"] == int(ln[i]) and abs(infos[i]["episode"]["r"] - float(ret[i])) < 1e-6, (k, i)
                    assert t["tactile"].shape == whole[i].shape and np.array_equal(t["tactile"], whole[i]), (k, i)
                    if vis is not None:
                        assert np.array_equal(t["visual"], vis[i]), (k, i)
                    seen += 1
                for i in np.flatnonzero(~done):
                    assert "terminal_observation" not in infos[i]
        assert seen >= 2 * n
        venv.close()
