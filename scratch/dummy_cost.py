import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, tactile_gym_amd as tg
from tactile_gym_amd.parallel import TorchShard
from bench import MODES
for ar in (True, False):
    v = tg.make_vec("edge_follow-v0", num_envs=1024, max_steps=100000, image_size=[128,128], env_modes=MODES, seed=1, obs_mode="torch", auto_reset=ar)
    sh = TorchShard(v, pipelined=True)
    a = torch.empty(1024, 2, device="cuda")
    with torch.cuda.stream(sh.stream):
        sh.reset()
        for _ in range(30): sh.step(a.uniform_(-0.25, 0.25))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(300): sh.step(a.uniform_(-0.25, 0.25))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("auto_reset", ar, "ms/step", round(1e3 * dt / 300, 4))
    v.close()
