import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, tactile_gym_amd as tg
from bench import SURF_MODES
v = tg.make_vec("surface_follow-v0", num_envs=1024, max_steps=200, image_size=[128,128], env_modes=SURF_MODES, seed=1, obs_mode="torch")
v.reset()
a = torch.empty(1024, 3, device="cuda")
for _ in range(30):
    v.step_async(a.uniform_(-0.25, 0.25)); v.sync()
v.close()
