"""Soak run on the MI355X box: every env at 1024 envs with its randomisations on, auto-reset, random actions; the state must stay finite and
the episode statistics plausible.  python tools/soak.py [steps]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
import tactile_gym_amd as tg

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
CASES = [("edge_follow-v0", bench.MODES, 200, 2), ("surface_follow-v0", bench.SURF_MODES, 200, 3), ("surface_follow-v2", bench.VERT_MODES, 200, 2),
         ("object_balance-v0", bench.BAL_MODES, 250, 2), ("object_balance-v0", dict(bench.BAL_MODES, object_mode="ball_on_plate"), 250, 2), ("object_push-v0", dict(bench.PUSH_MODES, rand_init_orn=True, rand_obj_mass=True), 120, 2),
         ("object_roll-v0", bench.ROLL_MODES, 60, 2)]
only = sys.argv[2] if len(sys.argv) > 2 else None       # e.g. ball_on_plate: run the cases whose modes mention it
for env_id, modes, max_steps, act_dim in CASES:
    if only and only not in env_id and only not in str(modes.values()):
        continue
    v = tg.make_vec(env_id, num_envs=1024, max_steps=max_steps, image_size=[64, 64], env_modes=modes, seed=123, obs_mode="numpy")
    v.reset()
    rng = np.random.default_rng(0)
    n_done, rsum, t0 = 0, 0.0, time.time()
    k = steps if env_id not in ("object_push-v0",) else min(steps, 200)
    for s in range(k):
        obs, rew, done, _ = v.step(rng.uniform(-0.25, 0.25, (1024, act_dim)).astype(np.float32))
        n_done += int(done.sum()); rsum += float(rew.sum())
        assert np.isfinite(rew).all(), (env_id, s)
        if s % 50 == 49 or s == k - 1:
            st = v.get_state()
            bad = {kk: int((~np.isfinite(a.reshape(1024, -1).astype(float))).any(1).sum()) for kk, a in st.items() if a.dtype.kind == "f"}
            assert not any(bad.values()), (env_id, s, {kk: b for kk, b in bad.items() if b})
    st = v.get_state()
    print(f"{env_id} {modes.get('object_mode', '')}: {k} steps ok, {n_done} episodes finished, mean reward {rsum / (1024 * k):.4f}, |qd| max {np.abs(st['qd']).max():.3f}, "
          f"tactile mean {obs['tactile'].mean():.2f}, {time.time() - t0:.1f} s", flush=True)
    v.close()
