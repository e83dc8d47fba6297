"""world_size-2 gloo test of the env sharding + gather-to-rank-0 exchange (parallel.ShardedVecEnv) on CPU.

The local shard here is the CPU oracle presented as torch tensors (the HIP shard needs a GPU); what is under test is
the multi-process logic: block partition, per-env seeds = seed + global index, gather order, broadcast of actions."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

MODES = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
             reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
N_LOCAL, SEED, STEPS = 2, 40, 2


class OracleShard:
    """CPU stand-in for TorchShard: n_local oracle envs with seeds seed + global index."""

    def __init__(self, rank, n_local, seed):
        from oracle.ref_env import OracleEdgeFollowEnv
        self.num_envs = n_local
        self.envs = [OracleEdgeFollowEnv(seed=seed + rank * n_local + i, max_steps=200, image_size=(64, 64), env_modes=MODES)
                     for i in range(n_local)]

    def border_info(self):
        """What TorchShard.border_info() hands out (the TacTip ring is 40 % of the image: payload "auto" ships the interior only)."""
        e = self.envs[0]
        mask = torch.from_numpy(e.border_mask.reshape(-1).astype(np.uint8))
        gray = torch.from_numpy(e.nodef_gray.reshape(-1).astype(np.uint8))
        return torch.nonzero(mask != 1).reshape(-1), torch.where(mask == 1, gray, torch.zeros_like(gray))

    def tile_template(self):
        """What TorchShard.tile_template() hands out: the untouched sensor's image (zero inside, the pasted ring outside)."""
        _, ring = self.border_info()
        return ring

    def _obs(self, dicts):
        obs = {"tactile": torch.from_numpy(np.stack([d["tactile"] for d in dicts]))}
        if self.visual:                             # visuotactile: the scene camera's rgb image rides in the same message
            obs["visual"] = torch.from_numpy(np.stack([e.visual_image() for e in self.envs]))
        return obs

    visual = False

    def reset(self):
        return self._obs([e.reset() for e in self.envs])

    def step(self, actions):
        outs = [e.step(actions[i].numpy()) for i, e in enumerate(self.envs)]
        return (self._obs([o[0] for o in outs]),
                torch.tensor([o[1] for o in outs], dtype=torch.float32), torch.tensor([o[2] for o in outs], dtype=torch.uint8), {})


class PackedOracleShard(OracleShard):
    """Like the HIP shard (TorchShard.packed / tg_get_packed_outputs): the step's outputs also live in one byte block
    [obs | pad to 16 | reward f32 | done u8], which ShardedVecEnv ships with a single copy."""

    def _block(self, obs, rew, done):
        nb = obs["tactile"].numel()
        off = (nb + 15) & ~15
        blk = torch.zeros(off + 4 * rew.numel() + done.numel(), dtype=torch.uint8)
        blk[:nb] = obs["tactile"].reshape(-1)
        blk[off:off + 4 * rew.numel()] = rew.view(torch.uint8)
        blk[off + 4 * rew.numel():] = done
        self._packed = (blk, off)

    def reset(self):                                 # the library's block exists from tg_create on
        obs = super().reset()
        self._block(obs, torch.zeros(self.num_envs), torch.zeros(self.num_envs, dtype=torch.uint8))
        return obs

    def step(self, actions):
        obs, rew, done, info = super().step(actions)
        self._block(obs, rew, done)
        return obs, rew, done, info

    def packed(self):
        return self._packed


PUSH_MODES = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
                  observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")


class PushOracleShard:
    """BASELINE config 4's shard (object_push-v0, tactile_and_feature): the observation dict carries `extended_feature` f32[n, 12]
    (object_push_env.py:611-629) next to the tactile image.  packed=True mimics the library's output block
    [obs | pad 16 | reward | done | pad 4 | feature] (tg_get_packed_outputs + tg_get_packed_feature)."""

    def __init__(self, rank, n_local, seed, packed):
        from oracle.ref_env import OracleObjectPushEnv
        self.num_envs, self._use_packed = n_local, packed
        self.envs = [OracleObjectPushEnv(seed=seed + rank * n_local + i, max_steps=1000, image_size=(128, 128), env_modes=PUSH_MODES)
                     for i in range(n_local)]

    @staticmethod
    def _obs(dicts):
        return {"tactile": torch.from_numpy(np.stack([d["tactile"] for d in dicts])),
                "extended_feature": torch.from_numpy(np.stack([d["extended_feature"] for d in dicts]).astype(np.float32))}

    def reset(self):
        obs = self._obs([e.reset() for e in self.envs])
        self._block(obs, torch.zeros(self.num_envs), torch.zeros(self.num_envs, dtype=torch.uint8))
        return obs

    def step(self, actions):
        outs = [e.step(actions[i].numpy()) for i, e in enumerate(self.envs)]
        obs = self._obs([o[0] for o in outs])
        rew = torch.tensor([o[1] for o in outs], dtype=torch.float32)
        done = torch.tensor([o[2] for o in outs], dtype=torch.uint8)
        self._block(obs, rew, done)
        return obs, rew, done, {}

    def _block(self, obs, rew, done):
        nb = obs["tactile"].numel()
        off = (nb + 15) & ~15
        off_f = (off + 4 * rew.numel() + done.numel() + 3) & ~3
        blk = torch.zeros(off_f + 4 * obs["extended_feature"].numel(), dtype=torch.uint8)
        blk[:nb] = obs["tactile"].reshape(-1)
        blk[off:off + 4 * rew.numel()] = rew.view(torch.uint8)
        blk[off + 4 * rew.numel():off + 4 * rew.numel() + done.numel()] = done
        blk[off_f:] = obs["extended_feature"].reshape(-1).view(torch.uint8)
        self._packed = (blk, off, off_f)

    def border_info(self):
        """What TorchShard.border_info() hands out, from the oracle's copy of the sensor constants."""
        e = self.envs[0]
        mask = torch.from_numpy(e.border_mask.reshape(-1).astype(np.uint8))
        gray = torch.from_numpy(e.nodef_gray.reshape(-1).astype(np.uint8))
        return torch.nonzero(mask != 1).reshape(-1), torch.where(mask == 1, gray, torch.zeros_like(gray))

    def tile_template(self):
        return self.border_info()[1]

    def __getattr__(self, name):
        if name == "packed" and self.__dict__.get("_use_packed"):
            return lambda: self._packed
        raise AttributeError(name)


def _push_worker(rank, world, port, out_path, overlap, packed, payload="full"):
    import torch.distributed as dist
    from tactile_gym_amd.parallel import ShardedVecEnv
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = ShardedVecEnv(PushOracleShard(rank, N_LOCAL, SEED, packed), dist, overlap=overlap, payload=payload)
    obs = env.reset()
    assert env.transport == "collective" and env.payload == ("full" if payload == "auto" else payload)   # auto: the DigiTac ring is under 10 % -> full
    assert obs["extended_feature"].shape == ((world if rank == 0 else 1) * N_LOCAL, 12)
    gen = torch.Generator().manual_seed(11)
    for _ in range(3):
        acts = (torch.rand(world * N_LOCAL, 2, generator=gen) - 0.5) * 0.5 if rank == 0 else torch.zeros(world * N_LOCAL, 2)
        obs, rew, done, _ = env.step(env.scatter_actions(acts))
    if overlap:
        last = env.flush()
        if rank == 0:
            obs, rew, done = last
    if rank == 0:
        torch.save({"tactile": obs["tactile"].clone(), "feature": obs["extended_feature"].clone(), "rew": rew.clone(), "done": done.clone()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,packed,payload", [(False, False, "full"), (True, True, "full"), (True, True, "interior"), (False, False, "auto"),
                                                    (True, True, "tiles"), (False, False, "tiles")])
def test_tactile_and_feature_reaches_rank0_world2(tmp_path, overlap, packed, payload):
    """SURVEY 8e / BASELINE config 4: `extended_feature f32[N/R, 12]` travels to rank 0 in the same per-step message as the images.
    payload "interior" / "auto": only the pixels inside the sensor's border mask are shipped, rank 0 restores the constant ring - the
    gathered images must equal the full ones bit for bit."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_push_worker, args=(world, port, out, overlap, packed, payload), nprocs=world, join=True)
    got = torch.load(out)
    ref = PushOracleShard(0, world * N_LOCAL, SEED, False)
    ref.reset()
    gen = torch.Generator().manual_seed(11)
    for _ in range(3):
        obs, rew, done, _ = ref.step((torch.rand(world * N_LOCAL, 2, generator=gen) - 0.5) * 0.5)
    assert got["feature"].shape == (world * N_LOCAL, 12) and torch.equal(got["feature"], obs["extended_feature"])
    assert torch.equal(got["tactile"], obs["tactile"]) and torch.allclose(got["rew"], rew) and torch.equal(got["done"], done)


def _worker(rank, world, port, out_path, overlap=False, packed=False, payload="auto", visual=False):
    import torch.distributed as dist
    from tactile_gym_amd.parallel import ShardedVecEnv
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = (PackedOracleShard if packed else OracleShard)(rank, N_LOCAL, SEED)
    shard.visual = visual
    env = ShardedVecEnv(shard, dist, overlap=overlap, payload=payload)
    assert env.num_envs == world * N_LOCAL and env.env_slice() == slice(rank * N_LOCAL, (rank + 1) * N_LOCAL)
    obs = env.reset()
    assert env.payload == ("interior" if payload == "auto" else payload)      # TacTip: "auto" picks the interior payload
    assert ("visual" in obs) == visual
    gen = torch.Generator().manual_seed(7)
    hist = [obs["tactile"].clone()]
    vis = None
    for _ in range(STEPS):
        acts = (torch.rand(world * N_LOCAL, 2, generator=gen) - 0.5) * 0.5 if rank == 0 else torch.zeros(world * N_LOCAL, 2)
        local = env.scatter_actions(acts)            # rank 0's batch is broadcast, every rank keeps its block
        obs, rew, done, _ = env.step(local)
        hist.append(obs["tactile"].clone())
        assert ("visual" in obs) == visual           # the same keys from reset() and step(), on every rank
    if overlap:                                       # pipelined: step k returned the batch of step k-1; flush() hands over the last one
        last = env.flush()
        if rank == 0:
            hist = [hist[0]] + hist[2:] + [last[0]["tactile"].clone()]
            obs, rew, done = last
    if rank == 0:
        assert obs["tactile"].shape == (world * N_LOCAL, 64, 64, 1) and rew.shape == (world * N_LOCAL,)
        torch.save({"obs": torch.stack(hist), "rew": rew, "done": done, "visual": obs["visual"].clone() if visual else None}, out_path)
    else:
        assert obs["tactile"].shape == (N_LOCAL, 64, 64, 1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,packed,payload,visual", [(False, False, "full", False), (True, False, "auto", False), (True, True, "auto", False),
                                                           (True, True, "full", False), (False, True, "interior", False), (True, False, "tiles", False),
                                                           (False, True, "tiles", False), (True, False, "interior", True), (False, False, "tiles", True)])
def test_shard_and_gather_world2(tmp_path, overlap, packed, payload, visual):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(world, port, out, overlap, packed, payload, visual), nprocs=world, join=True)
    got = torch.load(out)
    # single-process reference: the same 4 envs with seeds SEED..SEED+3 stepped with the same actions
    ref = OracleShard(0, world * N_LOCAL, SEED)
    ref.visual = visual
    obs = ref.reset()
    gen = torch.Generator().manual_seed(7)
    assert torch.equal(got["obs"][0], obs["tactile"])
    for k in range(STEPS):
        acts = (torch.rand(world * N_LOCAL, 2, generator=gen) - 0.5) * 0.5
        obs, rew, done, _ = ref.step(acts)
        assert torch.equal(got["obs"][k + 1], obs["tactile"])       # gather order = global env index
    assert torch.allclose(got["rew"], rew) and torch.equal(got["done"], done)
    if visual:                                      # visuotactile over several ranks: the rgb observation arrives with every step
        assert got["visual"].shape == (world * N_LOCAL, 64, 64, 3) and torch.equal(got["visual"], obs["visual"])


def test_tile_payload_round_trip_torch():
    """The tile payload's definition (parallel.torch_pack_tiles / torch_unpack_tiles, what the HIP kernels are checked against on the GPU):
    lossless on random images with a few live tiles, on an all-template batch (zero records) and on an all-live batch."""
    from tactile_gym_amd.parallel import TILE_REC, torch_pack_tiles, torch_unpack_tiles
    g = torch.Generator().manual_seed(3)
    n, H, W = 5, 64, 128
    tmpl = torch.randint(0, 256, (H * W,), dtype=torch.uint8, generator=g)
    cap = 16 + TILE_REC * n * (H // 16) * (W // 16)
    for mode in ("sparse", "none", "all"):
        img = tmpl.reshape(1, H, W).repeat(n, 1, 1).clone()
        if mode == "sparse":
            img[1, 17, 33] ^= 1; img[3, 63, 127] ^= 255; img[3, 0, 0] ^= 7; img[4, 16:32, 16:32] = 9
        elif mode == "all":
            img = img ^ 1
        msg = torch.zeros(cap, dtype=torch.uint8)
        count = torch_pack_tiles(torch, img.unsqueeze(-1), tmpl, msg)
        assert count == {"none": 0, "all": n * 32}.get(mode, count) and (mode != "sparse" or 3 <= count <= 5)
        out = torch.zeros(n, H * W, dtype=torch.uint8)
        torch_unpack_tiles(torch, msg, tmpl, n, H, W, out)
        assert torch.equal(out.reshape(n, H, W), img)
