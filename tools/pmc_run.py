import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, tactile_gym_amd as tg
from bench import MODES
v = tg.make_vec("edge_follow-v0", num_envs=1024, max_steps=200, image_size=[128,128], env_modes=MODES, seed=1, obs_mode="torch")
v.reset()
a = torch.empty(1024, 2, device="cuda")
for _ in range(40):
    v.step_async(a.uniform_(-0.25, 0.25)); v.sync()
v.close()
