"""Multi-GPU sharding: envs are independent, so rank r owns envs [r*n_local, (r+1)*n_local) and the only exchange is
one gather of (tactile obs, reward, done) to rank 0 per step — the MI355X-native replacement for
SubprocVecEnv's per-step pickled pipes (reference sb3_helpers/rl_utils.py:17-30).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).
The gather is a direct many-to-one exchange: every peer sends its shard to rank 0 over its own xGMI link, so the
7 links into rank 0 work concurrently (a ring would serialise the 16-64 MiB shards on one link).
"""
import numpy as np


class ShardedVecEnv:
    """Wraps this rank's local env shard (anything with reset()/step() returning per-shard tensors/arrays).

    `local` must expose num_envs, reset() -> {"tactile": tensor[n,H,W,1]}, step(a) -> (obs, reward, done, info) with
    torch tensors (device tensors under nccl, CPU tensors under gloo).  Rank 0's step() returns the gathered
    [world * n] batch; other ranks return their local shard (what an actor-only rank needs)."""

    def __init__(self, local, dist=None, root=0):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist, self.local, self.root = torch, dist, local, root
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_local = local.num_envs
        self.num_envs = self.n_local * self.world
        self._bufs = {}

    def env_slice(self):
        return slice(self.rank * self.n_local, (self.rank + 1) * self.n_local)

    def scatter_actions(self, actions_all):
        """Every rank is given (or rank 0 broadcasts) the full [N, act_dim] action batch; keep this rank's block."""
        t = actions_all
        if self.world > 1:
            self.dist.broadcast(t, src=self.root)
        return t[self.env_slice()]

    def _gather(self, name, t):
        if self.world == 1:
            return t
        t = t.contiguous()
        if self.rank == self.root:
            key = (name, tuple(t.shape), t.dtype, t.device)
            if key not in self._bufs:
                full = self.torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
                self._bufs[key] = (full, [full[i] for i in range(self.world)])
            full, views = self._bufs[key]
            self.dist.gather(t, gather_list=views, dst=self.root)
            return full.reshape((self.world * t.shape[0],) + tuple(t.shape[1:]))
        self.dist.gather(t, gather_list=None, dst=self.root)
        return t

    def reset(self):
        obs = self.local.reset()
        return {k: self._gather("obs_" + k, v) for k, v in obs.items()}

    def step(self, local_actions):
        obs, rew, done, info = self.local.step(local_actions)
        obs = {k: self._gather("obs_" + k, v) for k, v in obs.items()}
        return obs, self._gather("rew", rew), self._gather("done", done), info


class TorchShard:
    """Adapter: a TactileVecEnv (obs_mode='torch') presented with torch reward/done tensors, no host copies."""

    def __init__(self, venv):
        self.venv, self.num_envs = venv, venv.num_envs

    def reset(self):
        self.venv.reset()
        return {"tactile": self.venv.tactile_torch()}

    def step(self, actions):
        self.venv.step_async(actions)
        self.venv.sync()
        rew, done = self.venv.reward_done_torch()
        return {"tactile": self.venv.tactile_torch()}, rew, done, {}
