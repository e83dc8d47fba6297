"""Multi-GPU sharding: envs are independent, so rank r owns envs [r*n_local, (r+1)*n_local) and the only exchange is
one gather of (tactile obs, reward, done) to rank 0 per step — the MI355X-native replacement for
SubprocVecEnv's per-step pickled pipes (reference sb3_helpers/rl_utils.py:17-30).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).
The gather is a direct many-to-one exchange: every peer sends its shard to rank 0 over its own xGMI link, so the
7 links into rank 0 work concurrently (a ring would serialise the 16-64 MiB shards on one link).
"""
from . import _capi as capi

import numpy as np


class ShardedVecEnv:
    """Wraps this rank's local env shard (anything with reset()/step() returning per-shard tensors/arrays).

    `local` must expose num_envs, reset() -> {"tactile": tensor[n,H,W,1]}, step(a) -> (obs, reward, done, info) with
    torch tensors (device tensors under nccl, CPU tensors under gloo).  Rank 0's step() returns the gathered
    [world * n] batch; other ranks return their local shard (what an actor-only rank needs).

    Per step there is ONE collective: tactile obs (uint8), reward (float32) and done (uint8) are packed into one byte
    buffer per rank and gathered together (three small-latency collectives per 0.15 ms step would cost as much as the step).

    overlap=True ("overlap gather of step t with simulate of step t+1, double-buffered obs"): step() snapshots this
    rank's results into one of two staging buffers, starts the gather asynchronously and returns; rank 0 is handed the batch of
    the PREVIOUS step (complete by then), i.e. the learner side runs one step behind the simulators, and flush() waits for the
    last gather and returns its batch.  With overlap=False every step() returns its own gathered batch (synchronous VecEnv)."""

    def __init__(self, local, dist=None, root=0, overlap=False, force_collective=False, payload="auto"):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist, self.local, self.root = torch, dist, local, root
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_local = local.num_envs
        self.num_envs = self.n_local * self.world
        # force_collective: run the packed gather even with one rank (a 1-GPU box can then exercise the RCCL path end to end)
        self._solo = self.world == 1 and not force_collective
        self.overlap = bool(overlap) and not self._solo
        self._bufs = {}
        self._stage, self._full, self._views, self._pending = [None, None], [None, None], [None, None], [None, None]
        self._tick, self._layout, self._last = 0, None, None
        # payload: what the tactile part of the per-step message carries.  "full": every pixel.  "interior": only the 4-pixel words that hold
        # a pixel inside the sensor's border mask - the border ring of a TacTip image is a constant paste of the reference image
        # (tactile_sensor.py:291-292), which rank 0 fills in from its own copy of that constant: 62 % of the bytes of a 128 x 128 TacTip image
        # (61 % at 256 x 256) cross xGMI.  Two library kernels do the work (tg_pack_interior on the sender: 6.6 us per 1024 images,
        # tg_unpack_interior on rank 0: 9.3 us per 1024 images, measured on an MI355X); with one rank (edge_follow, 1024 envs, 128 x 128,
        # TG_BENCH_FORCE_COLLECTIVE=1) a step costs 0.062 ms without a gather, 0.097 ms with the full payload, 0.103 ms with the interior one,
        # i.e. it pays as soon as a link is in the way (16.8 MB per peer per step at ~76 GB/s per direction is ~0.22 ms).  "auto" (default):
        # interior when the shard exposes its border (`border_info()`) and the ring is at least 10 % of the image, else full.
        self._interior = None
        if payload not in ("auto", "full", "interior"):
            raise ValueError(f"payload {payload!r}")
        info = local.border_info() if (payload != "full" and hasattr(local, "border_info")) else None
        if payload == "interior" and info is None:
            raise ValueError("payload='interior' needs a shard with border_info() (border paste on)")
        if info is not None and not self._solo:
            idx, template = info
            if payload == "interior" or idx.numel() <= 0.9 * template.numel():
                self._interior = (idx, template)
        self._obs_full = [None, None]

    def env_slice(self):
        return slice(self.rank * self.n_local, (self.rank + 1) * self.n_local)

    def scatter_actions(self, actions_all):
        """Every rank is given (or rank 0 broadcasts) the full [N, act_dim] action batch; keep this rank's block."""
        t = actions_all
        if self.world > 1:
            self.dist.broadcast(t, src=self.root)
        return t[self.env_slice()]

    def _gather(self, name, t):
        if self._solo:
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

    # ---- packed exchange: [tactile bytes | pad to 16 | reward f32 | done u8 | pad to 4 | extended_feature f32[n][K] (if any)] per rank
    def _pack(self, slot, obs, rew, done):
        torch = self.torch
        tac, feat = obs["tactile"], obs.get("extended_feature")
        packed = self.local.packed() if hasattr(self.local, "packed") else None   # the library's own contiguous output block
        n = tac.shape[0]
        nb_t, nb_r, nb_d = tac.numel(), rew.numel() * 4, done.numel()
        fw_obs = int(feat.shape[1]) if feat is not None else 0
        if packed is not None:                       # (block, reward offset[, feature offset or -1]): the library's layout rules
            off_r = packed[1]
            off_f = packed[2] if len(packed) > 2 and packed[2] is not None else -1
            total = packed[0].numel()
            fw = (total - off_f) // (4 * n) if off_f >= 0 else 0       # the block's feature rows may be wider than the observation's
        else:
            off_r = (nb_t + 15) & ~15
            off_f = ((off_r + nb_r + nb_d + 3) & ~3) if feat is not None else -1
            fw = fw_obs
            total = off_f + 4 * n * fw if feat is not None else off_r + nb_r + nb_d
        shift = 0                                    # interior payload: the block after the tactile part moves up by this many bytes
        if self._interior is not None:
            k_int = int(self._interior[0].numel())
            shift = off_r - ((n * k_int + 15) & ~15)
            nb_t, off_r, total = n * k_int, off_r - shift, (total - shift + 15) & ~15   # (peer blocks stay 16-byte aligned on rank 0)
            off_f = off_f - shift if off_f >= 0 else -1
        if self._layout is None:
            assert tac.dtype == torch.uint8 and rew.dtype == torch.float32 and done.dtype == torch.uint8
            assert feat is None or (feat.dtype == torch.float32 and off_f >= 0 and fw >= fw_obs)
            assert total >= off_r + nb_r + nb_d and shift >= 0
            self._layout = (tuple(tac.shape), nb_t, off_r, nb_r, nb_d, off_f, fw, fw_obs)
        if self._stage[slot] is None:
            self._stage[slot] = torch.zeros(total, dtype=torch.uint8, device=tac.device)
            if self.rank == self.root:
                self._full[slot] = torch.empty((self.world, total), dtype=torch.uint8, device=tac.device)
                self._views[slot] = [self._full[slot][i] for i in range(self.world)]
        st = self._stage[slot]
        if self._interior is not None:
            if hasattr(self.local, "pack_interior"):
                self.local.pack_interior(st[:nb_t].view(n, -1))                    # one library kernel
            else:
                torch.index_select(tac.reshape(n, -1), 1, self._interior[0], out=st[:nb_t].view(n, -1))   # the pixels inside the border mask
            if packed is not None:
                rest = packed[0].numel() - (off_r + shift)
                st[off_r:off_r + rest].copy_(packed[0][off_r + shift:])            # reward | done | pad | feature, as laid out by the library
        elif packed is not None:
            st.copy_(packed[0])                                                   # one device copy
        else:
            st[:nb_t].copy_(tac.reshape(-1))
        if packed is None:
            st[off_r:off_r + nb_r].view(torch.float32).copy_(rew.reshape(-1))
            st[off_r + nb_r:off_r + nb_r + nb_d].copy_(done.reshape(-1))
            if feat is not None:
                st[off_f:off_f + 4 * n * fw].view(torch.float32).reshape(n, fw)[:, :fw_obs].copy_(feat)
        if st.is_cuda and not getattr(self.local, "pipelined", False):
            # the sources alias the env library's device buffers, which the next step's kernels (on the library's own stream) overwrite:
            # the snapshot must have been taken before step() returns (a 17 MB device copy, ~6 us).  A pipelined shard runs the
            # library on the current stream, where the copy is ordered before the next step by the stream itself.
            torch.cuda.current_stream(st.device).synchronize()
        return st

    def _start_gather(self, slot, async_op):
        st = self._stage[slot]
        if self.rank == self.root:
            return self.dist.gather(st, gather_list=self._views[slot], dst=self.root, async_op=async_op)
        return self.dist.gather(st, gather_list=None, dst=self.root, async_op=async_op)

    def _unpack(self, slot):
        torch = self.torch
        shape, nb_t, off_r, nb_r, nb_d, off_f, fw, fw_obs = self._layout
        full = self._full[slot]
        if self._interior is not None:               # scatter the interiors into images whose border ring is already in place
            idx, template = self._interior
            if self._obs_full[slot] is None:
                self._obs_full[slot] = template.reshape(1, -1).repeat(self.world * shape[0], 1).contiguous()
            img = self._obs_full[slot]
            if hasattr(self.local, "unpack_interior"):
                for r in range(self.world):      # one kernel per peer block (contiguous views, no staging copy): every pixel written once
                    self.local.unpack_interior(full[r, :nb_t].view(shape[0], -1), img[r * shape[0]:(r + 1) * shape[0]])
            else:
                img.index_copy_(1, idx, full[:, :nb_t].reshape(self.world * shape[0], -1))
            obs = {"tactile": img.reshape((self.world * shape[0],) + shape[1:])}
        else:
            obs = {"tactile": full[:, :nb_t].reshape((self.world * shape[0],) + shape[1:])}
        rew = full[:, off_r:off_r + nb_r].contiguous().view(torch.float32).reshape(-1)
        done = full[:, off_r + nb_r:off_r + nb_r + nb_d].reshape(-1)
        if fw_obs:   # config 4's tactile_and_feature observation (object_push_env.py:611-629) reaches rank 0 in the same message
            feat = full[:, off_f:off_f + 4 * shape[0] * fw].contiguous().view(torch.float32).reshape(self.world * shape[0], fw)
            obs["extended_feature"] = feat[:, :fw_obs]
        return obs, rew, done

    def step(self, local_actions):
        obs, rew, done, info = self.local.step(local_actions)
        if self._solo:
            return obs, rew, done, info
        slot = self._tick & 1
        if self._pending[slot] is not None:          # this staging buffer's previous gather (two steps ago) must be complete
            self._pending[slot].wait()
            self._pending[slot] = None
        self._pack(slot, obs, rew, done)
        if not self.overlap:
            self._start_gather(slot, False)
            self._tick += 1
            return self._unpack(slot) + (info,) if self.rank == self.root else (obs, rew, done, info)
        self._pending[slot] = self._start_gather(slot, True)
        self._tick += 1
        prev = slot ^ 1
        if self.rank != self.root:
            return obs, rew, done, info
        if self._tick == 1:                          # nothing gathered yet: hand back the local shard's view of step 0
            return obs, rew, done, info
        if self._pending[prev] is not None:
            self._pending[prev].wait()
            self._pending[prev] = None
        return self._unpack(prev) + (info,)

    def flush(self):
        """overlap=True: wait for the outstanding gathers; rank 0 gets the gathered batch of the last step."""
        if self._solo or not self.overlap or self._tick == 0:
            return None
        for k in (0, 1):
            if self._pending[k] is not None:
                self._pending[k].wait()
                self._pending[k] = None
        last = (self._tick - 1) & 1
        return self._unpack(last) if self.rank == self.root else None


class TorchShard:
    """Adapter: a TactileVecEnv (obs_mode='torch') presented with torch reward/done tensors, no host copies.

    pipelined=False: step() returns after the step has finished on the device (step_async + sync), like VecEnv.step_wait.
    pipelined=True: the library is put on a torch stream (`self.stream`) and step() only ENQUEUES the step; the returned tensors
    are valid for work enqueued on that stream afterwards (a policy forward pass, the packed gather), the usual CUDA-stream contract.
    The host then runs ahead of the device instead of idling through every step, so per-step launch overhead is hidden; the caller
    must do its torch work under `with torch.cuda.stream(shard.stream)` and synchronise before reading results on the host."""

    def __init__(self, venv, pipelined=False):
        self.venv, self.num_envs, self.pipelined, self.stream = venv, venv.num_envs, bool(pipelined), None
        if self.pipelined:
            import torch
            self.stream = torch.cuda.Stream(device=venv.tactile_torch().device)
            venv.sync()
            venv.set_stream(self.stream.cuda_stream)

    def packed(self):
        return self.venv.packed_torch()

    def pack_interior(self, dst):
        """This shard's current observations, interior pixels only, into the uint8 device tensor `dst` [n, K] (tg_pack_interior: one kernel
        on the library's stream)."""
        import ctypes as C
        capi.check(self.venv._L.tg_pack_interior(self.venv._ctx, C.c_void_p(dst.data_ptr())))

    def unpack_interior(self, src, dst):
        """Interiors `src` uint8 [m, K] -> full images `dst` uint8 [m, H*W] with the border ring restored (tg_unpack_interior)."""
        import ctypes as C
        m = int(src.shape[0])
        for lo in range(0, m, 32768):
            hi = min(m, lo + 32768)
            capi.check(self.venv._L.tg_unpack_interior(self.venv._ctx, C.c_void_p(src[lo:hi].data_ptr()), hi - lo, C.c_void_p(dst[lo:hi].data_ptr())))

    def border_info(self):
        """(flat indices of the pixels inside the border mask, the constant image of the border ring) as device tensors, or None when the
        border paste is off (the ring then carries rendered values)."""
        import torch
        sd = self.venv._sensor
        if sd.struct.turn_off_border:
            return None
        dev = self.venv.tactile_torch().device
        mask = torch.from_numpy(sd.border_mask.reshape(-1).astype("uint8")).to(dev)
        gray = torch.from_numpy(sd.nodef_gray.reshape(-1).astype("uint8")).to(dev)    # the truncating uint8 cast of tactile_sensor.py:291-292
        # the payload is made of the 4-pixel words that hold at least one interior pixel, padded to a multiple of 4 words (tg_pack_interior)
        words = torch.nonzero((mask.reshape(-1, 4) != 1).any(dim=1)).reshape(-1)
        if words.numel() % 4:
            words = torch.cat([words, words[-1:].repeat(4 - words.numel() % 4)])
        idx = (words.reshape(-1, 1) * 4 + torch.arange(4, device=words.device)).reshape(-1)
        return idx, torch.where(mask == 1, gray, torch.zeros_like(gray))

    def _obs(self):
        obs = {"tactile": self.venv.tactile_torch()}
        if getattr(self.venv, "_visual", False):
            obs["visual"] = self.venv.visual_torch()
        if "feature" in self.venv.observation_mode and self.venv.feature_dim:
            obs["extended_feature"] = self.venv.feature_torch()
        return obs

    def reset(self):
        self.venv.reset()
        return self._obs()

    def step(self, actions):
        self.venv.step_async(actions)
        if not self.pipelined:
            self.venv.sync()
        rew, done = self.venv.reward_done_torch()
        return self._obs(), rew, done, {}
