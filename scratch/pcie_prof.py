import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, tactile_gym_amd as tg
from bench import MODES
n = 1024
v = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=MODES, seed=1, auto_reset=True)
v.reset()
acts = np.random.default_rng(0).uniform(-0.25, 0.25, size=(64, n, 2)).astype(np.float32)
for k in range(10): v.step(acts[k])
pr = cProfile.Profile(); pr.enable()
for k in range(100): v.step(acts[k % 64])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
