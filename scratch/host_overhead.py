"""Host enqueue cost per component of the sharded step (1 rank, forced collective)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import tactile_gym_amd as tg
from tactile_gym_amd.parallel import ShardedVecEnv, TorchShard
from bench import MODES
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
n = 1024
venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=MODES, seed=1, auto_reset=True, obs_mode="torch")
shard = TorchShard(venv, pipelined=True)
env = ShardedVecEnv(shard, dist, overlap=True, force_collective=True)
act = torch.empty(n, 2, device="cuda")
T = {}
def tic(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
with torch.cuda.stream(shard.stream):
    env.reset()
    for _ in range(20):
        env.step(act.uniform_(-0.25, 0.25))
    torch.cuda.synchronize()
    K = 200
    for it in range(K):
        t0 = time.perf_counter(); a = act.uniform_(-0.25, 0.25); tic("uniform", t0)
        t0 = time.perf_counter(); obs, rew, done, _ = shard.step(a); tic("lib_step", t0)
        slot = env._tick & 1
        t0 = time.perf_counter()
        if env._pending[slot] is not None:
            env._pending[slot].wait(); env._pending[slot] = None
        tic("wait_slot", t0)
        t0 = time.perf_counter(); env._pack(slot, obs["tactile"], rew, done); tic("pack", t0)
        t0 = time.perf_counter(); env._pending[slot] = env._start_gather(slot, True); tic("gather", t0)
        env._tick += 1
        prev = slot ^ 1
        t0 = time.perf_counter()
        if env._pending[prev] is not None:
            env._pending[prev].wait(); env._pending[prev] = None
        tic("wait_prev", t0)
        t0 = time.perf_counter(); out = env._unpack(prev) if it > 0 else None; tic("unpack", t0)
        if it % 50 == 49:
            t0 = time.perf_counter(); torch.cuda.synchronize(); tic("drain", t0)
    torch.cuda.synchronize()
for k, v in T.items():
    print(f"{k:10s} {1e6 * v / K:8.1f} us/step")
dist.destroy_process_group()
