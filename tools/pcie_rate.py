"""Rate of the host-buffer path: VecEnv.step() with numpy actions in and numpy observations / rewards / dones out (PCIe both ways)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, tactile_gym_amd as tg
from bench import MODES
n = 1024
v = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=MODES, seed=1, auto_reset=True)
if "--tiles" in sys.argv:
    v.set_obs_transfer("tiles")       # only the 16 x 16 tiles that changed cross PCIe (host_tiles.py)
v.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-0.25, 0.25, size=(64, n, 2)).astype(np.float32)
stag = "--staggered" in sys.argv      # the episodes out of phase (tools/desync_rate.py): ~5 envs finish in every step, as in an RL run, and the
if stag:                              # caller reads their terminal observations the way SB3's rollout collection does
    phase = rng.integers(0, 200, size=n)
    for k in range(200):
        v.step(acts[k % 64])
        m = (phase == k).astype(np.uint8)
        if m.any():
            v.reset(m)
for k in range(10):
    v.step(acts[k])
t = time.perf_counter()
K = 100
ndone = 0
for k in range(K):
    obs, rew, done, info = v.step(acts[k % 64])
    if stag:
        for i in np.flatnonzero(done):
            ndone += 1
            _ = info[i]["terminal_observation"]
dt = time.perf_counter() - t
print(("episodes out of phase (%d finished in %d steps, terminal observations read), " % (ndone, K) if stag else "") +
      ("tile download, " if "--tiles" in sys.argv else "") + f"host-buffer path: {n * K / dt:.0f} env-steps/s, {1e3 * dt / K:.3f} ms/step, obs {obs['tactile'].shape} {obs['tactile'].dtype}")
if "--tiles" in sys.argv:
    d = v._tile_download
    print(f"  per fetch: pack + copy + sync {1e3 * d.t_device / d.calls:.3f} ms, host rebuild {1e3 * d.t_host / d.calls:.3f} ms, {d.last_bytes} bytes in the last message")
v.close()
