"""Tile-sparse observation download for the numpy VecEnv boundary (opt in: `venv.set_obs_transfer("tiles")`).

What the reference's sb3_helpers consume is a numpy batch per `VecEnv.step_wait()` (sb3_helpers/rl_utils.py:17-30, stable-baselines3's
rollout collection).  The plain path copies the whole uint8 batch device -> host every step (16.8 MB for 1024 x 128 x 128: the copy IS the
step, 2.1 M env-steps/s).  Here the device packs only the 16 x 16 tiles that differ from the untouched sensor's image (`tg_pack_tiles`, the
payload of the multi-GPU exchange), that message crosses PCIe into pinned memory, and `libtg_host.so` (host/tg_host_tiles.c, plain C)
rebuilds the batch in one of five persistent host buffers: the tiles that buffer's previous frame had live get the template back, the new
records land.  Lossless; the arrays handed out are five ring buffers, untouched for the next three calls (the restore half of a buffer's next
rebuild starts on the pool's workers one call before the buffer is filled again), like `copy_obs=False`.
Needs torch (device buffers, the pinned staging area, the copy); there is no fallback: a missing library or torch raises."""
import ctypes as C
import os

import numpy as np

from . import _capi as capi
from .parallel import TILE_REC, TorchShard

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libtg_host.so")
_host = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.isfile(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                               f"`tactile_gym_amd/host/build.sh` (gcc).")
        L = C.CDLL(HOST_LIB_PATH)
        L.tg_host_unpack_tiles.restype = C.c_int64
        L.tg_host_unpack_tiles.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.tg_host_fill_template.restype = C.c_int32
        L.tg_host_fill_template.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.tg_host_pool_create.restype = C.c_void_p
        L.tg_host_pool_create.argtypes = [C.c_int32]
        L.tg_host_pool_destroy.argtypes = [C.c_void_p]
        L.tg_host_pool_threads.restype = C.c_int32
        L.tg_host_pool_threads.argtypes = [C.c_void_p]
        L.tg_host_unpack_tiles_mt.restype = C.c_int64
        L.tg_host_unpack_tiles_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_int64)]
        L.tg_host_restore_begin.restype = C.c_int32
        L.tg_host_restore_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
        L.tg_host_pool_wait.argtypes = [C.c_void_p]
        L.tg_host_checksum.restype = C.c_uint64
        L.tg_host_checksum.argtypes = [C.c_void_p, C.c_int64]
        _host = L
    return _host


class HostTileBatch:
    """One persistent host batch [n, H, W] and the list of its tiles that differ from the template; `apply(msg)` brings it to the frame a
    tile message describes.  Pure host code (numpy + libtg_host.so): what the CPU tests drive."""

    ERRORS = {-1: "bad argument", -2: "bad message header", -3: "message shorter than its record count says", -4: "tile id out of range"}

    def __init__(self, tmpl, n, H, W, out=None, pool=None):
        self.pool = pool                     # a tg_host_pool (HostPool.handle) shared by the ring's buffers, or None: the calling thread only
        self.n, self.H, self.W = int(n), int(H), int(W)
        assert self.H % 16 == 0 and self.W % 16 == 0
        self.tmpl = np.ascontiguousarray(np.asarray(tmpl, dtype=np.uint8).reshape(-1))
        assert self.tmpl.size == self.H * self.W
        self.batch = out if out is not None else np.empty((self.n, self.H, self.W), dtype=np.uint8)
        assert self.batch.flags["C_CONTIGUOUS"] and self.batch.size == self.n * self.H * self.W and self.batch.dtype == np.uint8
        self.prev = np.zeros(self.n * (self.H // 16) * (self.W // 16), dtype=np.int32)
        self.n_prev = C.c_int64(0)
        if host_lib().tg_host_fill_template(self.tmpl.ctypes.data, self.n, self.H, self.W, self.batch.ctypes.data) != 0:
            raise RuntimeError("tg_host_fill_template failed")

    def restore_begin(self):
        """Start bringing this buffer back to the untouched-sensor image on the pool's workers (the restore half of its next rebuild); the
        caller goes on.  `restore_end()` before the buffer is used again."""
        rc = host_lib().tg_host_restore_begin(self.pool, self.tmpl.ctypes.data, self.n, self.H, self.W, self.batch.ctypes.data, self.prev.ctypes.data,
                                              self.n_prev.value)
        if rc < 0:
            raise RuntimeError(f"tg_host_restore_begin: {self.ERRORS.get(int(rc), rc)}")
        self._restoring = True

    def restore_end(self):
        if getattr(self, "_restoring", False):
            host_lib().tg_host_pool_wait(self.pool)
            self.n_prev.value = 0
            self._restoring = False

    def apply(self, msg):
        self.restore_end()
        msg = np.ascontiguousarray(msg)
        rc = host_lib().tg_host_unpack_tiles_mt(self.pool, msg.ctypes.data, msg.nbytes, self.tmpl.ctypes.data, self.n, self.H, self.W,
                                                self.batch.ctypes.data, self.prev.ctypes.data, C.byref(self.n_prev))
        if rc < 0:
            raise RuntimeError(f"tg_host_unpack_tiles: {self.ERRORS.get(int(rc), rc)}")
        return int(rc)


class HostPool:
    """The rebuild's thread pool (libtg_host.so: tg_host_pool_create): `threads` includes the calling thread; thread t owns the images
    [t n / P, (t + 1) n / P).  TG_HOST_THREADS overrides; 1 = no pool."""

    def __init__(self, threads=None):
        if threads is None:
            threads = int(os.environ.get("TG_HOST_THREADS", "4"))
        self.handle = host_lib().tg_host_pool_create(int(threads)) if threads > 1 else None
        self.threads = host_lib().tg_host_pool_threads(self.handle) if self.handle else 1

    def close(self):
        if self.handle:
            host_lib().tg_host_pool_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class TileDownload:
    """The device -> host leg: pack on the device, one pinned copy of the header and the records, rebuild on the host."""

    RING = 5      # a hand-out stays untouched for three further calls; the fifth buffer is the one whose restore runs ahead (see fetch)

    def __init__(self, venv):
        import torch
        self.torch, self.venv = torch, venv
        self.shard = TorchShard(venv)
        n, H, W = venv.num_envs, venv.H, venv.W
        if H % 16 or W % 16:
            raise ValueError("obs_transfer='tiles' needs image sides that are multiples of 16")
        dev = venv.tactile_torch().device
        T = (H // 16) * (W // 16)
        cap = 16 + TILE_REC * n * T
        self.cap = (cap + 15) & ~15
        self.counters = torch.zeros(4, dtype=torch.int32, device=dev)
        # Zero copy (round 5): the pack kernel stores the header, the records and the small block behind the images (reward | done | ...) STRAIGHT
        # into this pinned host buffer - device-visible at its own address - so a fetch is one launch and one synchronisation; until round 4 it
        # was the launch, three copies (message, rewards, dones) and the synchronisation, each copy a DMA command of its own behind the kernel.
        packed, obs_bytes, _ = venv.packed_torch()
        # [reward f32[n] | done u8[n]] of tg_get_packed_outputs' layout - NOT the extended_feature block behind them (with it the tail
        # passed tg_pack_tiles' 1 MiB limit at ~19.7 k envs of a 12-float feature and every fetch raised; the feature has its own copy, feature_numpy)
        self.tail = packed[obs_bytes:obs_bytes + 5 * n]
        if self.tail.numel() > (1 << 20):
            raise ValueError(f"obs_transfer='tiles': {n} envs need a {self.tail.numel()}-byte reward / done block, tg_pack_tiles carries at most 1 MiB")
        self.host_pk = torch.empty(self.cap + self.tail.numel(), dtype=torch.uint8).pin_memory()
        self.host_np = self.host_pk.numpy()
        self.rew_host = self.host_np[self.cap:self.cap + 4 * n].view(np.float32)
        self.done_host = self.host_np[self.cap + 4 * n:self.cap + 5 * n]
        self.zero_copy = os.environ.get("TG_TILES_ZERO_COPY", "1") != "0"      # A/B switch for the measurement: "0" packs on the device and copies
        # Round 6: what the finished envs' infos need - their ids, episode statistics and terminal images - rides along under the same synchronisation
        # (tg_pack_done_rows: one launch, stores into this pinned block); until then step_wait made two more blocking round trips per step in which any
        # env finished, i.e. nearly every step of an RL run.  DONE_CAP envs per step; a step with more falls back to the copies.
        self.done_rows = None
        self.done_cap = int(os.environ.get("TG_DONE_ROWS_CAP", "32"))
        if self.zero_copy and venv._cfg.auto_reset and self.done_cap > 0 and (H * W) % 16 == 0:
            nb = C.c_int64()
            capi.check(venv._L.tg_done_rows_bytes(venv._ctx, self.done_cap, C.byref(nb)))
            self.host_done = torch.zeros(int(nb.value), dtype=torch.uint8).pin_memory()
            d = self.host_done.numpy()
            cap = self.done_cap
            off = (16 + 12 * cap + 15) & ~15
            self._dr = dict(hdr=d[:16].view(np.uint32), ids=d[16:16 + 4 * cap].view(np.int32), ret=d[16 + 4 * cap:16 + 8 * cap].view(np.float32),
                            len=d[16 + 8 * cap:16 + 12 * cap].view(np.int32), rows=d[off:off + cap * H * W].reshape(cap, H, W, 1))
        else:
            self.host_done = None
        if not self.zero_copy:
            self.pk = torch.zeros(self.cap + self.tail.numel(), dtype=torch.uint8, device=dev)
        tmpl = self.shard.tile_template().cpu().numpy()
        self.pool = HostPool()
        self.ring = [HostTileBatch(tmpl, n, H, W, pool=self.pool.handle) for _ in range(self.RING)]
        self.i = -1
        self.last_bytes = 0
        self._last_count = 0
        self.t_device = self.t_host = 0.0      # seconds spent in fetch(): pack + copy + synchronise / host rebuild
        self.calls = 0
        self.rd_fresh = False

    def close(self):
        """Wait for the restore job that fetch() leaves running on the pool's workers, stop the workers, drop the buffers (without
        this a dropped TileDownload relied on attribute destruction order to keep workers from writing into freed arrays)."""
        if getattr(self, "pool", None) is not None:
            for hb in self.ring:
                try:
                    hb.restore_end()
                except Exception:  # noqa: BLE001
                    pass
            self.pool.close()
            self.pool = None
        self.ring = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def fetch(self):
        """The current observation batch as uint8 [n, H, W, 1] (one of the ring buffers)."""
        import time
        torch, v = self.torch, self.venv
        t0 = time.perf_counter()
        stream = torch.cuda.current_stream(self.tail.device)
        self.done_rows = None
        if self.zero_copy:
            if self.host_done is not None:               # (on the context's own - blocking - stream, ahead of the pack on torch's legacy default stream,
                                                         #  which waits for it: the one synchronisation covers both)
                capi.check(v._L.tg_pack_done_rows(v._ctx, C.c_void_p(self.host_done.data_ptr()), self.done_cap))
            self.shard.pack_tiles(self.host_pk.data_ptr(), self.counters, tail=self.tail, tail_offset=self.cap)
            stream.synchronize()
            if self.host_done is not None:
                k = int(self._dr["hdr"][0])
                if k <= self.done_cap and int(self._dr["hdr"][3]) == 0x74674452:
                    self.done_rows = (self._dr["ids"][:k].copy(), self._dr["ret"][:k].copy(), self._dr["len"][:k].copy(), self._dr["rows"][:k].copy())
            count = int(self.host_np[:4].view(np.int32)[0])
            nb = 16 + TILE_REC * count
        else:
            self.shard.pack_tiles(self.pk.data_ptr(), self.counters, tail=self.tail, tail_offset=self.cap)
            guess = min(self.cap, 16 + TILE_REC * (self._last_count + self._last_count // 4 + 64))
            self.host_pk[:guess].copy_(self.pk[:guess], non_blocking=True)
            self.host_pk[self.cap:].copy_(self.pk[self.cap:], non_blocking=True)
            stream.synchronize()
            count = int(self.host_np[:4].view(np.int32)[0])
            nb = 16 + TILE_REC * count
            if nb > guess:
                self.host_pk[guess:nb].copy_(self.pk[guess:nb], non_blocking=True)
                stream.synchronize()
        self.rd_fresh = True
        t1 = time.perf_counter()
        self._last_count = count
        self.i = (self.i + 1) % self.RING
        hb = self.ring[self.i]
        hb.apply(self.host_np[:nb])
        self.last_bytes = nb
        t2 = time.perf_counter()
        # the buffer the NEXT call fills was handed out four calls ago - its contract ("untouched for the next three calls") has run out: the pool's
        # workers restore its live tiles now, under the next device step, and that call only scatters
        if self.pool.handle:
            self.ring[(self.i + 1) % self.RING].restore_begin()
        self.t_device += t1 - t0; self.t_host += t2 - t1; self.calls += 1
        return hb.batch.reshape(v.num_envs, v.H, v.W, 1)
