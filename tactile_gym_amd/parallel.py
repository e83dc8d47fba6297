"""Multi-GPU sharding: envs are independent, so rank r owns envs [r*n_local, (r+1)*n_local) and the only exchange is
one message of (tactile obs, reward, done[, extended_feature, visual]) to rank 0 per step — the MI355X-native replacement
for SubprocVecEnv's per-step pickled pipes (reference sb3_helpers/rl_utils.py:17-30).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).
The exchange is many-to-one: every peer reaches rank 0 over its own xGMI link, so the 7 links into rank 0 work concurrently (a ring
would serialise the 16-64 MiB shards on one link).

payload (what the tactile part of a rank's message carries):
  "full"      every pixel.
  "interior"  only the 4-pixel words that hold a pixel inside the sensor's border mask; rank 0 restores the constant ring
              (tactile_sensor.py:291-292): 62 % of the bytes of a TacTip image.
  "tiles"     only the 16 x 16 tiles that differ from the untouched sensor's image (zero inside, the ring outside): lossless, 9 % of the
              bytes of an edge_follow batch, 22 % object_balance at 256 x 256, 48 % surface_follow, 73 % object_push.  Variable length.
  "auto"      ipc transport: tiles; collective transport: interior when the ring is at least 10 % of the image, else full.
transport (how it travels):
  "collective"  torch.distributed: one gather of the fixed-size message per step (full / interior).  A tile message has a length only the
                sender's device knows, so it goes as an exact-size send / recv after a small fixed gather that carries the counts
                (reward | done | feature | visual | count): two host reads per step, kept for completeness and for the CPU tests.
  "ipc"         rank 0 owns the receive slots (tg_ipc_alloc); every peer's pack kernel stores straight into its slot of rank 0's HBM
                (tg_ipc_open), ordered by stream-side flags (tg_flag_set / tg_flag_wait): no staging copy, no host in the loop, and the
                variable-length tile message costs nothing extra.  Needs one process per GPU on one node.
  "auto"        ipc when the set-up handshake succeeds on every rank, else collective.
"""
import ctypes as C
import os

from . import _capi as capi

TILE_MAGIC = 0x54475431
TILE_REC = 272


def _align(x, a):
    return (x + a - 1) // a * a


# ---- the tile payload restated with torch ops: what the CPU (gloo) tests run, and the definition the kernels are checked against
def torch_pack_tiles(torch, tac, tmpl, dst):
    """tac uint8 [n, H, W(, 1)], tmpl uint8 [H*W] -> message in dst (uint8, capacity bytes); returns the record count."""
    n, H, W = int(tac.shape[0]), int(tac.shape[1]), int(tac.shape[2])
    TH, TW = H // 16, W // 16
    T = TH * TW
    t = tac.reshape(n, TH, 16, TW, 16).permute(0, 1, 3, 2, 4).reshape(n * T, 256)
    tm = tmpl.reshape(TH, 16, TW, 16).permute(0, 2, 1, 3).reshape(T, 256)
    live = (t != tm.repeat(n, 1)).any(dim=1)
    ids = torch.nonzero(live).reshape(-1)
    count = int(ids.numel())
    rec = torch.zeros(count, TILE_REC, dtype=torch.uint8, device=tac.device)
    rec[:, :4] = ids.to(torch.int32).reshape(-1, 1).contiguous().view(torch.uint8)
    rec[:, 16:] = t[ids]
    dst[:16] = torch.tensor([count, n, T, TILE_MAGIC], dtype=torch.int32, device=tac.device).view(torch.uint8)
    dst[16:16 + TILE_REC * count] = rec.reshape(-1)
    return count


def torch_unpack_tiles(torch, src, tmpl, n, H, W, dst):
    """message in src -> dst uint8 [n, H*W]."""
    TH, TW = H // 16, W // 16
    T = TH * TW
    hdr = src[:16].contiguous().view(torch.int32)
    count = int(hdr[0])
    assert int(hdr[3]) == TILE_MAGIC and int(hdr[2]) == T and 0 <= count <= n * T
    rec = src[16:16 + TILE_REC * count].reshape(count, TILE_REC)
    ids = rec[:, :4].contiguous().view(torch.int32).reshape(-1).to(torch.int64)
    tm = tmpl.reshape(TH, 16, TW, 16).permute(0, 2, 1, 3).reshape(T, 256)
    tiles = tm.repeat(n, 1).clone()
    tiles[ids] = rec[:, 16:]
    dst.copy_(tiles.reshape(n, TH, TW, 16, 16).permute(0, 1, 3, 2, 4).reshape(n, H * W))


class _DevArray:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class _IpcSlots:
    """Rank 0's receive slots and the flags beside them, shared with the peers by IPC handle.  Layout of the allocation:
    [flags: 4096 B | slot 0: world x msg | slot 1: world x msg]; flags (uint32, one per 64 bytes): ready[slot][rank] at word
    16 * (slot * world + rank), consumed[slot][rank] at word 16 * (2 * world + slot * world + rank), handshake[rank] behind them."""

    FLAG_BYTES = 16384

    def __init__(self, torch, dist, rank, world, root, msg_bytes, device, timeout_ms):
        self.torch, self.dist, self.rank, self.world, self.root, self.msg, self.device = torch, dist, rank, world, root, msg_bytes, device
        self.timeout_ms = int(timeout_ms)
        self.L = capi.lib()
        assert 16 * 4 * (5 * world + 1) <= self.FLAG_BYTES, "too many ranks for the flag block"
        total = self.FLAG_BYTES + 2 * world * msg_bytes
        self.base, self.owner, self.failed, self._fp = C.c_void_p(), rank == root, None, {}
        handle = (C.c_uint8 * 64)()
        payload = [None]
        if self.owner:
            try:
                capi.check(self.L.tg_ipc_alloc(total, C.byref(self.base), handle))
                payload = [bytes(handle)]
                self.uncached = bool(self.L.tg_ipc_alloc_was_uncached())
                if not self.uncached:
                    import warnings
                    warnings.warn("tg_ipc_alloc: the runtime refused uncached device memory for the receive slots; they are plain hipMalloc memory, and "
                                  "only the set-up handshake vouches for peer stores becoming visible to rank 0's reads (exchange_info()['receive_slots_uncached'])")
            except Exception as e:  # noqa: BLE001 - the peers must still be told (they wait in the broadcast)
                self.failed, self.base = e, C.c_void_p()
        dist.broadcast_object_list(payload, src=root)
        if not self.owner:
            if payload[0] is None:
                self.failed = RuntimeError("rank 0 could not allocate the receive slots")
            else:
                try:
                    h = (C.c_uint8 * 64).from_buffer_copy(payload[0])
                    capi.check(self.L.tg_ipc_open(h, C.byref(self.base)))
                except Exception as e:  # noqa: BLE001
                    self.failed, self.base = e, C.c_void_p()
        self.err = torch.zeros(1, dtype=torch.int32, device=device)       # timeouts of this rank's waits (bit = flag lane)
        self.local = torch.as_tensor(_DevArray(self.base.value, total), device=device) if (self.owner and self.failed is None) else None

    uncached = None                                   # rank 0: whether the receive slots are uncached device memory (None elsewhere)
    fixed_stream = None                               # a pipelined shard pins its stream: no per-call lookup (ShardedVecEnv sets it)

    def _stream(self):
        if self.fixed_stream is not None:
            return self.fixed_stream
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def slot_ptr(self, slot, r):
        return self.base.value + self.FLAG_BYTES + (slot * self.world + r) * self.msg

    def slot_tensor(self, slot):                      # rank 0 only: uint8 [world, msg]
        off = self.FLAG_BYTES + slot * self.world * self.msg
        return self.local[off:off + self.world * self.msg].view(self.world, self.msg)

    def _flag_ptr(self, kind, slot, r):
        key = (kind, slot, r)
        p = self._fp.get(key)
        if p is None:
            word = 16 * ({"ready": 0, "consumed": 2 * self.world, "hello": 4 * self.world}[kind] + slot * self.world + r)
            p = self._fp[key] = C.c_void_p(self.base.value + 4 * word)
        return p

    def set(self, kind, slot, r, n, value):
        capi.check(self.L.tg_flag_set(self._stream(), self._flag_ptr(kind, slot, r), n, 16, value & 0xFFFFFFFF))

    def wait(self, kind, slot, r, n, value):
        capi.check(self.L.tg_flag_wait(self._stream(), self._flag_ptr(kind, slot, r), n, 16, value & 0xFFFFFFFF,
                                       C.c_void_p(self.err.data_ptr()), self.timeout_ms))

    def copy(self, dst_ptr, src_tensor):
        capi.check(self.L.tg_copy_bytes(self._stream(), C.c_void_p(dst_ptr), C.c_void_p(src_tensor.data_ptr()), src_tensor.numel() * src_tensor.element_size()))

    def copy2(self, dst1, src1, dst2, src2):
        capi.check(self.L.tg_copy_bytes2(self._stream(), C.c_void_p(dst1), C.c_void_p(src1.data_ptr()), src1.numel() * src1.element_size(),
                                         C.c_void_p(dst2), C.c_void_p(src2.data_ptr()), src2.numel() * src2.element_size()))

    def handshake(self, timeout_ms=5000):
        """Every peer stores a pattern into its slot and raises its flag; rank 0 waits (bounded) and checks the bytes."""
        torch = self.torch
        keep, self.timeout_ms = self.timeout_ms, int(timeout_ms)
        oks = [None] * self.world
        self.dist.all_gather_object(oks, self.failed is None)      # every rank holds a mapping, or nobody goes on
        if not all(oks):
            self.timeout_ms = keep
            return False
        pat = torch.arange(16, dtype=torch.uint8, device=self.device) + 16 * self.rank + 1
        ok = True
        if not self.owner:
            self.copy(self.slot_ptr(0, self.rank), pat)
            self.set("hello", 0, self.rank, 1, 1)
        else:
            for r in range(self.world):
                if r != self.root:
                    self.wait("hello", 0, r, 1, 1)
            torch.cuda.current_stream(self.device).synchronize()
            got = self.slot_tensor(0)[:, :16].cpu()
            for r in range(self.world):
                if r != self.root and not torch.equal(got[r], (torch.arange(16, dtype=torch.uint8) + 16 * r + 1)):
                    ok = False
            self.slot_tensor(0)[:, :16].zero_()
        torch.cuda.current_stream(self.device).synchronize()
        self.timeout_ms = keep
        oks = [None] * self.world
        self.dist.all_gather_object(oks, bool(ok and int(self.err.item()) == 0))
        if not all(oks):
            self.failed = self.failed or RuntimeError("a pattern stored by a peer did not arrive in rank 0's slot within the time limit")
        self.err.zero_()
        return all(oks)

    def poll(self):
        """The error word without a synchronisation: an asynchronous copy into pinned memory is started on the exchange's stream and the value the
        PREVIOUS polls delivered is looked at - a timeout shows up one or two polls after it happened, at the cost of a 4-byte copy."""
        if self._err_host is None:
            self._err_host = self.torch.zeros(1, dtype=self.torch.int32).pin_memory()
        e = int(self._err_host[0])
        if e:
            raise RuntimeError(f"rank {self.rank}: exchange flags timed out after {self.timeout_ms} ms (lanes 0x{e & 0xFFFFFFFF:x}): a partner rank is missing or stuck")
        self._err_host.copy_(self.err, non_blocking=True)

    _err_host = None

    def check(self):
        e = int(self.err.item())
        if e:
            raise RuntimeError(f"rank {self.rank}: exchange flags timed out after {self.timeout_ms} ms (lanes 0x{e & 0xFFFFFFFF:x}): a partner rank is missing or stuck")

    def close(self):
        if getattr(self, "_closed", False):
            return
        self._closed = True
        self.torch.cuda.synchronize(self.device)
        self.local = None
        if not self.owner and self.base.value:
            self.L.tg_ipc_close(self.base)
        self.dist.barrier()                           # every mapping is gone before the owner frees
        if self.owner and self.base.value:
            self.L.tg_ipc_free(self.base)
        self.base = C.c_void_p()


class ShardedVecEnv:
    """Wraps this rank's local env shard (anything with reset()/step() returning per-shard tensors/arrays).

    `local` must expose num_envs, reset() -> {"tactile": tensor[n,H,W,1], ...}, step(a) -> (obs, reward, done, info) with
    torch tensors (device tensors under nccl / ipc, CPU tensors under gloo).  Rank 0's step() returns the gathered
    [world * n] batch; other ranks return their local shard (what an actor-only rank needs).

    Per step there is ONE message per rank: [tactile part | pad to 16 | reward f32 | done u8 | pad to 4 | extended_feature f32[n][K] |
    pad to 16 | visual u8[n][H][W][3]] (the last two when the observation has them).

    overlap=True ("overlap gather of step t with simulate of step t+1, double-buffered obs"): step() snapshots this
    rank's results into one of two slots, starts the exchange asynchronously and returns; rank 0 is handed the batch of
    the PREVIOUS step (complete by then), i.e. the learner side runs one step behind the simulators, and flush() waits for the
    last exchange and returns its batch.  With overlap=False every step() returns its own gathered batch (synchronous VecEnv).
    A batch handed out stays valid until the next step() / reset() call.

    READ-ONLY on rank 0 with the ipc transport (exchange_info()["rank0_draws_into_batch"]): rank 0's block of each of the two alternating
    gathered batches IS the env library's render target (tg_set_obs_targets) - the block raster's persistent, incrementally updated image buffer,
    of which only the changed 16 x 16 blocks are rewritten per step - so the gathered tactile batch must not be modified in place (copy before
    augmenting; until round 5 that block was a copy).  venv.set_obs_guard(True) checks it per target."""

    def __init__(self, local, dist=None, root=0, overlap=False, force_collective=False, payload="auto", transport="collective", timeout_ms=20000):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist, self.local, self.root = torch, dist, local, root
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_local = local.num_envs
        self.num_envs = self.n_local * self.world
        # force_collective: run the exchange even with one rank (a 1-GPU box can then exercise the RCCL path end to end)
        self._solo = self.world == 1 and not force_collective
        self.overlap = bool(overlap) and not self._solo
        if payload not in ("auto", "full", "interior", "tiles"):
            raise ValueError(f"payload {payload!r}")
        if transport not in ("auto", "collective", "ipc"):
            raise ValueError(f"transport {transport!r}")
        self._want_payload, self._want_transport, self._timeout_ms = payload, transport, timeout_ms
        self.payload = self.transport = None         # decided with the first message (needs the observation's shapes)
        self._interior = self._tiles = None
        self._bufs, self._ipc, self._lay = {}, None, None
        self._stage, self._full, self._views, self._pending, self._obs_full = [None, None], [None, None], [None, None], [None, None], [None, None]
        self._tick, self._handed = 0, 0              # messages started / the index of the newest message rank 0 has unpacked
        self._last_out = None                        # (obs, reward, done) of that message
        self._counters = None
        self._last_counts = None
        self._prev_ids = [None, None]
        self._root_cache, self._out_cache, self._tac_view = [None, None], [None, None], [None, None]
        self._captured = set()                       # step kinds whose graph capture has been made safe (_quiesce_before_capture)

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

    # ------------------------------------------------------------------ message layout and set-up (first message)
    def _setup(self, obs):
        torch = self.torch
        tac, feat, vis = obs["tactile"], obs.get("extended_feature"), obs.get("visual")
        assert tac.dtype == torch.uint8
        dev, n = tac.device, int(tac.shape[0])
        H, W = int(tac.shape[1]), int(tac.shape[2])
        # transport
        want = self._want_transport
        if want in ("ipc", "auto") and (dev.type != "cuda" or not hasattr(self.local, "raw")):
            if want == "ipc":
                raise ValueError("transport='ipc' needs a device-resident shard (TorchShard)")
            want = "collective"
        # payload
        payload = self._want_payload
        info = self.local.border_info() if hasattr(self.local, "border_info") else None
        tile_ok = hasattr(self.local, "tile_template") and H % 16 == 0 and W % 16 == 0 and tac[0].numel() == H * W
        if payload == "interior" and info is None:
            raise ValueError("payload='interior' needs a shard with border_info() (border paste on)")
        if payload == "tiles" and not tile_ok:
            raise ValueError("payload='tiles' needs a shard with tile_template() and image sides that are multiples of 16")
        if payload == "auto":
            if want in ("ipc", "auto") and tile_ok:
                payload = "tiles"
            elif info is not None and info[0].numel() <= 0.9 * info[1].numel():
                payload = "interior"
            else:
                payload = "full"
        packed = self.local.packed() if hasattr(self.local, "packed") else None   # the library's own contiguous output block
        nb_full = tac.numel()
        if packed is not None:                       # (block, reward offset[, feature offset or -1]): the library's layout rules
            lib_r = int(packed[1])
            lib_f = int(packed[2]) if len(packed) > 2 and packed[2] is not None else -1
            rest_bytes = packed[0].numel() - lib_r
            f_in_rest = lib_f - lib_r if lib_f >= 0 else -1
            fw = (packed[0].numel() - lib_f) // (4 * n) if lib_f >= 0 else 0       # the block's feature rows may be wider than the observation's
        else:
            lib_r = -1
            f_in_rest = _align(5 * n, 4) if feat is not None else -1
            fw = int(feat.shape[1]) if feat is not None else 0
            rest_bytes = f_in_rest + 4 * n * fw if feat is not None else 5 * n
        fw_obs = int(feat.shape[1]) if feat is not None else 0
        assert feat is None or (feat.dtype == torch.float32 and f_in_rest >= 0 and fw >= fw_obs)
        if payload == "interior":
            self._interior = info
            img_cap = n * int(info[0].numel())
        elif payload == "tiles":
            self._tiles = (self.local.tile_template(), H, W)
            img_cap = 16 + TILE_REC * n * (H // 16) * (W // 16)
        else:
            img_cap = nb_full
        off_rest = _align(img_cap, 16)
        off_vis = _align(off_rest + rest_bytes, 16)
        vis_bytes = vis.numel() if vis is not None else 0
        off_cnt = _align(off_vis + vis_bytes, 16)    # tiles: a copy of the tile header, so that the fixed-size tail carries the count
        total = off_cnt + (16 if payload == "tiles" else 0)
        self._lay = dict(n=n, H=H, W=W, tac_shape=tuple(tac.shape), nb_full=nb_full, img_cap=img_cap, off_rest=off_rest, rest_bytes=rest_bytes,
                         lib_r=lib_r, f_in_rest=f_in_rest, fw=fw, fw_obs=fw_obs, off_vis=off_vis, vis_bytes=vis_bytes,
                         vis_shape=tuple(vis.shape) if vis is not None else None, off_cnt=off_cnt, total=total, dev=dev)
        self.payload = payload
        if payload == "tiles" and dev.type == "cuda":
            self._counters = torch.zeros(4, dtype=torch.int32, device=dev)
        # ipc set-up (collective on failure when "auto"); every rank takes the same branch
        self.transport = "collective"
        if want in ("ipc", "auto"):
            ipc = _IpcSlots(torch, self.dist, self.rank, self.world, self.root, total, dev, self._timeout_ms)
            if ipc.handshake():                       # the same verdict on every rank
                self._ipc, self.transport = ipc, "ipc"
                if getattr(self.local, "pipelined", False) and getattr(self.local, "stream", None) is not None:
                    ipc.fixed_stream = C.c_void_p(self.local.stream.cuda_stream)
            else:
                why = ipc.failed
                ipc.close()
                if want == "ipc":
                    raise RuntimeError(f"transport='ipc': the set-up handshake failed ({why})")
        if self.transport == "collective" and self._want_payload == "auto" and payload == "tiles":
            # the tile payload was chosen for the ipc transport: without it fall back to the fixed-size choice
            self._tiles = None
            self._want_payload = "interior" if (info is not None and info[0].numel() <= 0.9 * info[1].numel()) else "full"
            self._want_transport = "collective"
            self._counters = None
            return self._setup(obs)
        for slot in (0, 1):
            if self.transport == "ipc":
                if self.rank == self.root:
                    self._full[slot] = self._ipc.slot_tensor(slot)
                    self._stage[slot] = self._full[slot][self.root]
            else:
                self._stage[slot] = torch.zeros(total, dtype=torch.uint8, device=dev)
                if self.rank == self.root:
                    self._full[slot] = torch.zeros((self.world, total), dtype=torch.uint8, device=dev)
        if self.transport == "ipc" and self.rank != self.root:
            self._scratch = torch.zeros(max(rest_bytes, 16), dtype=torch.uint8, device=dev) if packed is None else None
        # Rank 0 draws its own shard STRAIGHT into its block of the gathered batch (round 5): the two alternating batches are the env library's
        # render targets 1 and 2 (tg_set_obs_targets: each with its own changed-block record and step graphs), selected before every step by
        # the slot its message will use.  Until round 4 the shard was drawn into the library's buffer and copied over (16.8 MB per step at
        # 1024 envs: the one-rank ipc + tiles step 57.5 us against 43 without an exchange).
        self._direct = False
        venv = getattr(self.local, "venv", None)
        if (self.transport == "ipc" and self.rank == self.root and self.payload == "tiles" and hasattr(self.local, "unpack_tiles_multi")
                and venv is not None and hasattr(venv, "set_obs_targets") and os.environ.get("TG_NO_DIRECT_BATCH") is None):
            venv.sync()
            venv.set_obs_targets([self._batch_buffer(s)[self.root * n:(self.root + 1) * n].data_ptr() for s in (0, 1)])
            self._direct = True
            # these two targets' step graphs can only be captured now that the batches exist, i.e. under the process group: the first steps
            # on them go through the quiesce below even on a primed shard (once, in reset(), outside any timed region; no collective runs
            # between the steps of the ipc transport, so the watchdog has nothing to poll while the two captures happen)
            self._captured.clear()
            self._quiesce_always = True

    # ------------------------------------------------------------------ sender side
    def _rest_tensor(self, obs, rew, done):
        """[reward f32 | done u8 | pad | feature f32[n][fw]] as one uint8 tensor: the library's own block when there is one."""
        torch, L = self.torch, self._lay
        packed = self.local.packed() if hasattr(self.local, "packed") else None
        if packed is not None:
            return packed[0][L["lib_r"]:]
        n = L["n"]
        rest = torch.zeros(L["rest_bytes"], dtype=torch.uint8, device=L["dev"])
        rest[:4 * n].view(torch.float32).copy_(rew.reshape(-1))
        rest[4 * n:5 * n].copy_(done.reshape(-1))
        feat = obs.get("extended_feature")
        if feat is not None:
            rest[L["f_in_rest"]:].view(torch.float32).reshape(n, L["fw"])[:, :L["fw_obs"]].copy_(feat)
        return rest

    def _pack_local(self, st, obs, rew, done):
        """This rank's message into the torch tensor `st` (uint8 [total]; a staging buffer, or rank 0's own receive slot)."""
        torch, L = self.torch, self._lay
        tac, n = obs["tactile"], self._lay["n"]
        if self.payload == "interior":
            dst = st[:L["img_cap"]].view(n, -1)
            if hasattr(self.local, "pack_interior"):
                self.local.pack_interior(dst)                                           # one library kernel
            else:
                torch.index_select(tac.reshape(n, -1), 1, self._interior[0], out=dst)   # the pixels inside the border mask
        elif self.payload == "tiles":
            if hasattr(self.local, "pack_tiles"):
                self.local.pack_tiles(st.data_ptr(), self._counters)
            else:
                torch_pack_tiles(torch, tac, self._tiles[0], st)
            st[L["off_cnt"]:L["off_cnt"] + 16].copy_(st[:16])
        else:
            st[:L["nb_full"]].copy_(tac.reshape(-1))
        st[L["off_rest"]:L["off_rest"] + L["rest_bytes"]].copy_(self._rest_tensor(obs, rew, done))
        if L["vis_bytes"]:
            st[L["off_vis"]:L["off_vis"] + L["vis_bytes"]].copy_(obs["visual"].reshape(-1))

    def _pack_remote(self, slot, obs, rew, done):
        """ipc transport, a peer: the message goes straight into this rank's slot of rank 0's memory (library kernels on raw pointers)."""
        L, ipc = self._lay, self._ipc
        dst = ipc.slot_ptr(slot, self.rank)
        if self.payload == "interior":
            self.local.pack_interior_ptr(dst)
        rest = self._rest_tensor(obs, rew, done)
        if rest.data_ptr() % 16:                      # 16-byte aligned ends are wanted; the library's block is, an assembled one is too
            rest = rest.clone()
        if self.payload == "tiles" and rest.numel() <= (1 << 20):
            self.local.pack_tiles(dst, self._counters, tail=rest, tail_offset=L["off_rest"])   # images and the block behind them: one launch
            rest = None
        elif self.payload == "tiles":
            self.local.pack_tiles(dst, self._counters)
        elif self.payload != "interior":
            ipc.copy(dst, obs["tactile"].reshape(-1))
        if rest is not None:
            ipc.copy(dst + L["off_rest"], rest)
        if L["vis_bytes"]:
            ipc.copy(dst + L["off_vis"], obs["visual"].reshape(-1))

    def _sync_if_unpipelined(self, dev):
        if dev.type == "cuda" and not getattr(self.local, "pipelined", False):
            # the sources alias the env library's device buffers, which the next step's kernels (on the library's own stream) overwrite:
            # the snapshot must have been taken before step() returns.  A pipelined shard runs the library on the current stream, where
            # the copy is ordered before the next step by the stream itself.
            self.torch.cuda.current_stream(dev).synchronize()

    def _send(self, t, obs, rew, done, async_op):
        """Start message t (1, 2, ...) of this rank; returns what has to be waited for before its slot is reused / unpacked."""
        torch, dist, L = self.torch, self.dist, self._lay
        slot, root = t & 1, self.rank == self.root
        if self.transport == "ipc":
            ipc = self._ipc
            if root:
                fused_flag = t > 2 and self.payload == "tiles" and hasattr(self.local, "unpack_tiles_multi") and not L["vis_bytes"]
                if t > 2 and not fused_flag:              # the batch of message t - 2 has been consumed: its slot may be overwritten
                    ipc.set("consumed", slot, 0, self.world, t - 2)
                if self.payload == "tiles" and hasattr(self.local, "unpack_tiles_multi"):
                    # rank 0's own images are not packed and unpacked: they go straight into their block of the batch (one copy), and
                    # only what rides behind the image part goes into its slot (where _receive reads reward / done / ... of every rank)
                    c = self._root_cache[slot]
                    if c is None or c[4] != obs["tactile"].data_ptr():     # the library's buffers never move: pointers are taken once per slot
                        n = L["n"]
                        rest = self._rest_tensor(obs, rew, done)
                        c = self._root_cache[slot] = (C.c_void_p(self._batch_buffer(slot)[self.root * n:(self.root + 1) * n].data_ptr()),
                                                      C.c_void_p(obs["tactile"].data_ptr()), obs["tactile"].numel(),
                                                      C.c_void_p(self._stage[slot][L["off_rest"]:].data_ptr()), obs["tactile"].data_ptr(),
                                                      C.c_void_p(rest.data_ptr()), rest.numel(), hasattr(self.local, "packed"))
                    if not c[7]:                                           # an assembled block is a new tensor every step
                        rest = self._rest_tensor(obs, rew, done)
                        c = c[:5] + (C.c_void_p(rest.data_ptr()), rest.numel(), False)
                        self._keep = rest
                    n_img = 0 if c[0].value == c[1].value else c[2]        # drawn in place (render target = this block): only the small block moves
                    if fused_flag:                                         # the "consumed" signal rides in the copy's launch (one dependent launch less per step)
                        capi.check(ipc.L.tg_copy_bytes2_flag(ipc._stream(), c[0], c[1], n_img, c[3], c[5], c[6], ipc._flag_ptr("consumed", slot, 0), self.world, 16,
                                                             (t - 2) & 0xFFFFFFFF))
                    else:
                        capi.check(ipc.L.tg_copy_bytes2(ipc._stream(), c[0], c[1], n_img, c[3], c[5], c[6]))
                    if L["vis_bytes"]:
                        ipc.copy(self._stage[slot][L["off_vis"]:].data_ptr(), obs["visual"].reshape(-1))
                else:
                    self._pack_local(self._stage[slot], obs, rew, done)
            else:
                ipc.wait("consumed", slot, self.rank, 1, t - 2)
                self._pack_remote(slot, obs, rew, done)
                ipc.set("ready", slot, self.rank, 1, t)
            self._sync_if_unpipelined(L["dev"])
            return None
        st = self._stage[slot]
        self._pack_local(st, obs, rew, done)
        self._sync_if_unpipelined(L["dev"])
        if self.payload != "tiles":
            views = [self._full[slot][i] for i in range(self.world)] if root else None
            return dist.gather(st, gather_list=views, dst=self.root, async_op=async_op)
        # tiles over collectives: a small fixed gather of everything behind the records (it carries every rank's count), then exact-size
        # send / recv of the records
        tail = st[L["off_rest"]:]
        views = [self._full[slot][i][L["off_rest"]:] for i in range(self.world)] if root else None
        dist.gather(tail, gather_list=views, dst=self.root)
        works = []
        if root:
            counts = self._full[slot][:, L["off_cnt"]:L["off_cnt"] + 4].contiguous().view(torch.int32).reshape(-1).cpu().tolist()
            self._last_counts = counts
            for r in range(self.world):
                nb = 16 + TILE_REC * int(counts[r])
                if r == self.root:
                    self._full[slot][r][:nb].copy_(st[:nb])
                else:
                    works.append(dist.irecv(self._full[slot][r][:nb], src=r))
        else:
            count = int(st[L["off_cnt"]:L["off_cnt"] + 4].view(torch.int32).item())
            works.append(dist.isend(st[:16 + TILE_REC * count], dst=self.root))
        if not async_op:
            for w in works:
                w.wait()
            return None
        return works

    @staticmethod
    def _wait(work):
        if work is None:
            return
        for w in (work if isinstance(work, list) else [work]):
            w.wait()

    def _batch_buffer(self, slot):
        """rank 0: the uint8 [world * n, H * W] images of slot `slot` (interior / tile payloads are unpacked into it)."""
        if self._obs_full[slot] is None:
            L = self._lay
            self._obs_full[slot] = self.torch.zeros((self.world * L["n"], L["H"] * L["W"]), dtype=self.torch.uint8, device=L["dev"])
            if self.payload == "interior":
                self._obs_full[slot].copy_(self._interior[1].reshape(1, -1).expand(self.world * L["n"], -1))
            elif self.payload == "tiles":             # starts as the template everywhere; unpacking restores only what the last message touched
                self._obs_full[slot].copy_(self._tiles[0].reshape(1, -1).expand(self.world * L["n"], -1))
                tiles = L["n"] * (L["H"] // 16) * (L["W"] // 16)
                self._prev_ids[slot] = self.torch.zeros(self.world * (tiles + 1), dtype=self.torch.int32, device=L["dev"])
        return self._obs_full[slot]

    # ------------------------------------------------------------------ rank 0: message t -> batch
    def _receive(self, t):
        torch, L = self.torch, self._lay
        slot = t & 1
        if self.transport == "ipc":                           # one wave per contiguous run of peers, lane r polls ready[slot][r]
            if self.root > 0:
                self._ipc.wait("ready", slot, 0, self.root, t)
            if self.root < self.world - 1:
                self._ipc.wait("ready", slot, self.root + 1, self.world - 1 - self.root, t)
        full, n, w = self._full[slot], L["n"], self.world
        shape = L["tac_shape"]
        tv = self._tac_view[slot]
        if self.payload == "full":
            if tv is None:                                    # one rank: a view of the slot; several: the images are strided across the
                tv = full[:, :L["nb_full"]].reshape((w * n,) + shape[1:])     # messages, and reshape gathers them (a copy per step)
                if w == 1:
                    self._tac_view[slot] = tv
            obs = {"tactile": tv}
        elif self.payload == "tiles" and self.transport == "ipc" and hasattr(self.local, "unpack_tiles_multi"):
            img = self._batch_buffer(slot)                    # rank 0's own block was copied in by _send; the peers' messages in two launches
            if w > 1:
                self.local.unpack_tiles_multi(full.data_ptr(), L["total"], w, self.root, n, img.data_ptr(), self._prev_ids[slot].data_ptr())
            if tv is None:
                tv = self._tac_view[slot] = img.reshape((w * n,) + shape[1:])
            obs = {"tactile": tv}
        else:
            img = self._batch_buffer(slot)
            for r in range(w):
                blk = img[r * n:(r + 1) * n]
                if self.payload == "interior":
                    src = full[r, :L["img_cap"]].view(n, -1)
                    if hasattr(self.local, "unpack_interior"):
                        self.local.unpack_interior(src, blk)          # one kernel per peer block: every pixel written once
                    else:
                        blk.index_copy_(1, self._interior[0], src)
                elif hasattr(self.local, "unpack_tiles"):
                    self.local.unpack_tiles(full[r].data_ptr(), n, blk.data_ptr())
                else:
                    torch_unpack_tiles(torch, full[r], self._tiles[0], n, L["H"], L["W"], blk)
            if tv is None:
                tv = self._tac_view[slot] = img.reshape((w * n,) + shape[1:])
            obs = {"tactile": tv}
        oc = self._out_cache[slot]
        if oc is None:
            # views of fixed buffers, built once per slot; what is strided across the ranks' messages (reward, done, feature) is gathered
            # into contiguous outputs by one small copy each per step
            o = L["off_rest"]
            rew_src = full[:, o:o + 4 * n].view(torch.float32)                       # [world, n], row stride = message size
            done_src = full[:, o + 4 * n:o + 5 * n]
            rew_out = torch.empty((w, n), dtype=torch.float32, device=L["dev"]) if w > 1 else None
            done_out = torch.empty((w, n), dtype=torch.uint8, device=L["dev"]) if w > 1 else None
            feat_src = feat_out = None
            if L["fw_obs"]:   # config 4's tactile_and_feature observation (object_push_env.py:611-629) reaches rank 0 in the same message
                f = o + L["f_in_rest"]
                feat_src = full[:, f:f + 4 * n * L["fw"]].view(torch.float32).reshape(w, n, L["fw"]) if w == 1 else \
                    full[:, f:f + 4 * n * L["fw"]].view(torch.float32).unflatten(1, (n, L["fw"]))
                feat_out = torch.empty((w, n, L["fw"]), dtype=torch.float32, device=L["dev"]) if w > 1 else None
            vis_src = full[:, L["off_vis"]:L["off_vis"] + L["vis_bytes"]] if L["vis_bytes"] else None
            oc = self._out_cache[slot] = (None, rew_src, rew_out, done_src, done_out, feat_src, feat_out, vis_src)
        _, rew_src, rew_out, done_src, done_out, feat_src, feat_out, vis_src = oc
        if w > 1:
            rew_out.copy_(rew_src); done_out.copy_(done_src)
            rew, done = rew_out.reshape(-1), done_out.reshape(-1)
        else:
            rew, done = rew_src.reshape(-1), done_src.reshape(-1)
        if feat_src is not None:
            if w > 1:
                feat_out.copy_(feat_src)
            obs["extended_feature"] = (feat_out if w > 1 else feat_src).reshape(w * n, L["fw"])[:, :L["fw_obs"]]
        if vis_src is not None:                               # (a view with one rank, gathered by reshape with several)
            obs["visual"] = vis_src.reshape((w * n,) + L["vis_shape"][1:])
        self._handed = t
        self._last_out = (obs, rew, done)
        return obs, rew, done

    # ------------------------------------------------------------------ VecEnv surface
    def reset(self):
        self._aim()
        obs = self.local.reset()
        if self._solo:
            return obs
        if self._lay is None:
            self._setup(obs)
        if self.transport != "ipc":
            out = {k: self._gather("obs_" + k, v) for k, v in obs.items()}
            self._quiesce_before_capture("step", "random")   # here rather than in the first step: outside any timed region
            return out
        # ipc: the reset observations travel like a step's message (zero reward / done), synchronously
        n = self._lay["n"]
        zr = self.torch.zeros(n, dtype=self.torch.float32, device=self._lay["dev"])
        zd = self.torch.zeros(n, dtype=self.torch.uint8, device=self._lay["dev"])
        self._drain()
        self._tick += 1
        self._send(self._tick, obs, zr, zd, False)
        out = self._receive(self._tick)[0] if self.rank == self.root else obs
        self._quiesce_before_capture("step", "random")
        return out

    def _drain(self):
        for k in (0, 1):
            self._wait(self._pending[k])
            self._pending[k] = None

    def _quiesce_before_capture(self, *kinds):
        """The first step of each kind makes the library capture its step graph (tg_step / tg_step_random).  While a HIP stream is capturing,
        hipEventQuery from ANOTHER thread can fail with hipErrorCapturedEvent, and torch's RCCL process group has such a thread: its watchdog
        polls the end events of the collectives still in its list every 100 ms, and a poll that fell into the few hundred microseconds of a
        capture aborted the process (2 of ~100 one-rank bench runs, all ranks would go down with it).  The ordering that cannot race is
        TorchShard.prime() before the process group is created (bench.py does that): then this is a no-op.  The fallback for a shard that was
        not primed: finish all device work, then give the watchdog three periods to retire the completed collectives - it polls nothing while
        its list is empty.  reset() does it for both kinds (outside any timed region); a step without a reset before it for its own kind."""
        todo = [k for k in kinds if k not in self._captured]
        if not todo or self._solo:
            return
        self._captured.update(todo)
        if not capi.step_graphs_enabled():
            return                                   # round 6: steps are enqueued launch by launch - the library captures nothing, there is nothing to race with
        if getattr(self.local, "primed", False) and not getattr(self, "_quiesce_always", False):
            return                                   # TorchShard.prime(): the graphs were captured before the process group existed - nothing to wait for
        if not (getattr(self.local, "raw", False) and self.torch.cuda.is_available()):
            return                                   # a host-side shard (the gloo tests): nothing is captured
        import time
        self.torch.cuda.synchronize()
        time.sleep(0.3)

    def _aim(self):
        """Rank 0, direct mode: the next message is t = tick + 1 and lands in slot t & 1 - the library draws this step's images there."""
        if getattr(self, "_direct", False):
            self.local.venv.select_obs_target(1 + ((self._tick + 1) & 1))

    def step_random(self, seed, first_draw=0, restart=False):
        """step(action_space.sample()) on every rank's shard (TorchShard.step_random: the draw inside the step's graph), then the exchange of step().
        Every rank passes its own seed."""
        self._quiesce_before_capture("random")
        self._aim()
        return self._exchange(*self.local.step_random(seed, first_draw, restart))

    def step(self, local_actions):
        self._quiesce_before_capture("step")
        self._aim()
        return self._exchange(*self.local.step(local_actions))

    def _exchange(self, obs, rew, done, info):
        if self._solo:
            return obs, rew, done, info
        if self._lay is None:
            self._setup(obs)
        self._tick += 1
        t = self._tick
        slot = t & 1
        self._wait(self._pending[slot])              # this slot's previous message (two steps ago) must be complete
        self._pending[slot] = None
        root = self.rank == self.root
        if not self.overlap:
            self._send(t, obs, rew, done, False)
            out = self._receive(t) + (info,) if root else (obs, rew, done, info)
            if self._ipc is not None:
                # the synchronous path (VecEnv.step_wait semantics): the caller reads this batch next, so the flag waits that delivered it are
                # checked now - a missing or stuck peer raises here instead of stale slots being unpacked step after step
                self.torch.cuda.current_stream(self._lay["dev"]).synchronize()
                self._ipc.check()
            return out
        self._pending[slot] = self._send(t, obs, rew, done, True)
        if self._ipc is not None and (t & 15) == 0:
            self._ipc.poll()                          # overlap: no synchronisation on the step path, a timeout surfaces within ~32 steps (and at flush / close)
        if not root or self._handed >= t - 1 or t == 1:   # nothing newer to hand out yet
            if root and self._last_out is not None and self._handed == t - 1:
                # right after an ipc reset(): message t - 1 IS the reset's batch, handed out by reset() already.  It is handed out again (its zero
                # reward / done with it) rather than the local shard, so that rank 0 sees [world * n] batches at every step
                return self._last_out + (info,)
            return obs, rew, done, info
        self._wait(self._pending[slot ^ 1])
        self._pending[slot ^ 1] = None
        return self._receive(t - 1) + (info,)

    def flush(self):
        """overlap=True: wait for the outstanding messages; rank 0 gets the gathered batch of the last step."""
        if self._solo or not self.overlap or self._tick == 0:
            return None
        self._drain()
        out = self._receive(self._tick) if self.rank == self.root else None
        if self._ipc is not None:
            self.torch.cuda.current_stream(self._lay["dev"]).synchronize()
            self._ipc.check()
        return out

    def exchange_info(self):
        """What travelled: payload / transport in use, the message capacity per rank and, for the tile payload, the bytes of the newest
        message per rank (rank 0, after a flush: read from the tile headers)."""
        if self._lay is None:
            return None
        L = self._lay
        out = {"payload": self.payload, "transport": self.transport, "message_bytes_capacity": L["total"],
               "full_payload_bytes": _align(L["nb_full"], 16) + L["total"] - L["off_rest"]}
        if self._ipc is not None and self._ipc.uncached is not None:
            out["receive_slots_uncached"] = self._ipc.uncached
        if self.rank == self.root:
            out["rank0_draws_into_batch"] = bool(getattr(self, "_direct", False))   # its shard is rendered in place (tg_set_obs_targets), not copied
        if self.payload == "tiles" and self.rank == self.root and self._handed:
            hdr = self._full[self._handed & 1][:, :4].contiguous().view(self.torch.int32).reshape(-1).cpu().tolist()
            if self.transport == "ipc" and hasattr(self.local, "unpack_tiles_multi"):
                hdr[self.root] = 0                        # rank 0's own images never become a message (copied straight into the batch)
            out["tile_records_last_message"] = hdr
            out["message_bytes_last"] = [16 + TILE_REC * int(c) + L["total"] - L["off_rest"] for c in hdr]
        elif self.payload != "tiles":
            out["message_bytes_last"] = [L["total"]] * self.world
        what = {"full": "every pixel", "interior": "interior pixels only, border ring restored on rank 0",
                "tiles": "only the 16x16 tiles that differ from the untouched sensor's image"}[self.payload]
        how = ("stored by each peer's pack kernel straight into rank 0's IPC-mapped receive slot over its own xGMI link, stream-side flags"
               if self.transport == "ipc" else "one packed RCCL gather to rank 0" + (" + exact-size send / recv of the tile records" if self.payload == "tiles" else ""))
        out["what"] = f"one message per rank per step (obs u8: {what}; reward f32, done u8), {how}, overlapped with the next step's simulation"
        return out

    def close(self):
        if getattr(self, "_direct", False):
            self._direct = False
            try:
                self.local.venv.sync()
                self.local.venv.set_obs_targets([])       # the library goes back to its own buffer before the batches are freed
            except Exception:  # noqa: BLE001
                pass
        if self._ipc is not None:
            self._drain()
            self._full = [None, None]
            self._stage = [None, None]
            ipc, self._ipc = self._ipc, None
            self.torch.cuda.current_stream(self._lay["dev"]).synchronize()
            e = int(ipc.err.item())
            ipc.close()
            if e:                                     # a timeout nobody has been told about yet (no flush, fewer than 16 steps since it happened)
                raise RuntimeError(f"rank {self.rank}: exchange flags timed out (lanes 0x{e & 0xFFFFFFFF:x}): a partner rank was missing or stuck; "
                                   f"batches handed out since then held stale slots")


class TorchShard:
    """Adapter: a TactileVecEnv (obs_mode='torch') presented with torch reward/done tensors, no host copies.

    pipelined=False: step() returns after the step has finished on the device (step_async + sync), like VecEnv.step_wait.
    pipelined=True: the library is put on a torch stream (`self.stream`) and step() only ENQUEUES the step; the returned tensors
    are valid for work enqueued on that stream afterwards (a policy forward pass, the packed gather), the usual CUDA-stream contract.
    The host then runs ahead of the device instead of idling through every step, so per-step launch overhead is hidden; the caller
    must do its torch work under `with torch.cuda.stream(shard.stream)` and synchronise before reading results on the host."""

    raw = True     # device resident: the ipc transport can address this shard's buffers

    def __init__(self, venv, pipelined=False):
        self.venv, self.num_envs, self.pipelined, self.stream, self._stream_ptr = venv, venv.num_envs, bool(pipelined), None, None
        if self.pipelined:
            import torch
            self.stream = torch.cuda.Stream(device=venv.tactile_torch().device)
            venv.sync()
            venv.set_stream(self.stream.cuda_stream)

    def packed(self):
        return self.venv.packed_torch()

    def _cur_stream(self):
        if self.pipelined:                            # the shard's own stream, fixed for its lifetime
            if self._stream_ptr is None:
                self._stream_ptr = C.c_void_p(self.stream.cuda_stream)
            return self._stream_ptr
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.venv.tactile_torch().device).cuda_stream)

    def pack_interior(self, dst):
        """This shard's current observations, interior pixels only, into the uint8 device tensor `dst` [n, K] (tg_pack_interior: one kernel
        on the library's stream)."""
        capi.check(self.venv._L.tg_pack_interior(self.venv._ctx, C.c_void_p(dst.data_ptr())))

    def pack_interior_ptr(self, dst_ptr):
        capi.check(self.venv._L.tg_pack_interior(self.venv._ctx, C.c_void_p(dst_ptr)))

    def unpack_interior(self, src, dst):
        """Interiors `src` uint8 [m, K] -> full images `dst` uint8 [m, H*W] with the border ring restored (tg_unpack_interior)."""
        m = int(src.shape[0])
        for lo in range(0, m, 32768):
            hi = min(m, lo + 32768)
            capi.check(self.venv._L.tg_unpack_interior(self.venv._ctx, C.c_void_p(src[lo:hi].data_ptr()), hi - lo, C.c_void_p(dst[lo:hi].data_ptr())))

    def tile_template(self):
        """uint8 [H*W] device tensor: the image of the untouched sensor (zero inside, the pasted ring outside) tiles are compared with."""
        if not hasattr(self, "_tmpl"):
            import torch
            p = C.c_void_p()
            capi.check(self.venv._L.tg_get_tile_template(self.venv._ctx, C.byref(p)))
            self._tmpl = torch.as_tensor(_DevArray(p.value, self.venv.H * self.venv.W), device=self.venv.tactile_torch().device)
        return self._tmpl

    def pack_tiles(self, dst_ptr, counters, tail=None, tail_offset=0):
        """This shard's current observations as a tile message at the raw device address `dst_ptr` (tg_pack_tiles, current torch stream);
        `counters`: a zeroed int32 device tensor of at least 2 elements in local memory; `tail`: a small uint8 tensor the same launch copies
        to dst_ptr + tail_offset."""
        v = self.venv
        capi.check(v._L.tg_pack_tiles(self._cur_stream(), C.c_void_p(v.tactile_torch().data_ptr()), C.c_void_p(self.tile_template().data_ptr()),
                                      v.num_envs, v.H, v.W, C.c_void_p(dst_ptr), C.c_void_p(counters.data_ptr()),
                                      C.c_void_p(tail.data_ptr() if tail is not None else None), tail.numel() if tail is not None else 0, tail_offset))

    def unpack_tiles_multi(self, src_ptr, stride, n_ranks, skip_rank, n_images, dst_ptr, prev_ids_ptr=None):
        """The tile messages of n_ranks ranks (`stride` bytes apart) -> their blocks of the batch, two launches; skip_rank's block is left
        alone; prev_ids_ptr: the destination's list of last-live tiles (only those get the template back)."""
        v = self.venv
        capi.check(v._L.tg_unpack_tiles_multi(self._cur_stream(), C.c_void_p(src_ptr), stride, n_ranks, skip_rank, C.c_void_p(self.tile_template().data_ptr()),
                                              n_images, v.H, v.W, C.c_void_p(dst_ptr), C.c_void_p(prev_ids_ptr)))

    def unpack_tiles(self, src_ptr, n_images, dst_ptr):
        v = self.venv
        capi.check(v._L.tg_unpack_tiles(self._cur_stream(), C.c_void_p(src_ptr), C.c_void_p(self.tile_template().data_ptr()), n_images, v.H, v.W,
                                        C.c_void_p(dst_ptr)))

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

    def prime(self, device_actions=None):
        """Make the library capture its step graphs NOW (tg_step's two slots - host actions / the caller's device tensor - and tg_step_random's),
        by one reset and one step of each kind.  Meant to be called BEFORE the process group exists: a capture that happens while torch's RCCL
        watchdog thread may poll an event is what aborted one-rank runs in rounds 3-4 (ShardedVecEnv._quiesce_before_capture waited three
        watchdog periods instead: a timing workaround).  With the graphs captured up front no step of the rollout captures anything, whatever the
        watchdog does - an ordering, not a delay.  `device_actions`: the float32 [n, act_dim] CUDA tensor the rollout will pass to step()
        (tg_step pins the first device pointer it sees to its in-place graph)."""
        import contextlib
        import numpy as np
        import torch
        v = self.venv
        ctx = torch.cuda.stream(self.stream) if self.pipelined else contextlib.nullcontext()
        with ctx:
            self.reset()
            self.step(np.zeros((v.num_envs, v.act_dim), dtype=np.float32))          # slot 0: the context's own action buffer
            if device_actions is not None:
                device_actions.zero_()
                self.step(device_actions)                                            # slot 1: this tensor, in place
            self.step_random(0, 0, restart=True)                                    # the random-action graph
        v.sync()
        self.primed = True
        return self

    def step(self, actions):
        self.venv.step_async(actions)
        if not self.pipelined:
            self.venv.sync()
        rew, done = self.venv.reward_done_torch()
        return self._obs(), rew, done, {}

    def step_random(self, seed, first_draw=0, restart=False):
        """step(action_space.sample()) with the draw inside the step's graph (TactileVecEnv.step_random_async)."""
        self.venv.step_random_async(seed, first_draw, restart)
        if not self.pipelined:
            self.venv.sync()
        rew, done = self.venv.reward_done_torch()
        return self._obs(), rew, done, {}
