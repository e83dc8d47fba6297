"""Device-resident vectorised tactile env: the SB3 `VecEnv` surface over the HIP C ABI.

Replaces `make_vec_env(env_id, n_envs, vec_env_cls=SubprocVecEnv)` (reference sb3_helpers/rl_utils.py:17-30): instead
of N OS processes each running one PyBullet server, all N envs live in one process on one MI355X; `step_async`
enqueues the step kernels on a HIP stream and `step_wait` synchronises, mirroring SubprocVecEnv's async split.

Observations follow the reference layout: dict of arrays, tactile image uint8 [N, H, W, 1]
(base_tactile_env.py:200-210, 247-282).  With `obs_mode="numpy"` (default, what SB3 wrappers such as VecFrameStack /
VecTransposeImage expect) they are copied to host; with `obs_mode="torch"` the tactile observation is returned as a
zero-copy `torch.uint8` CUDA(HIP) tensor aliasing the library's device buffer.
"""
import ctypes as C
import os
import time
import warnings

import numpy as np

from . import _capi as capi
from . import spaces


def _optional_base(module, name):
    """`module.name` when it can be imported, else `object`: the classes below are real gym.Env / stable_baselines3 VecEnv subclasses
    wherever those packages exist (the reference's callers wrap envs with Monitor, VecFrameStack, VecTransposeImage:
    sb3_helpers/rl_utils.py:17-35, 49-68) and plain duck types where they do not (the build image has neither)."""
    try:
        import importlib
        return getattr(importlib.import_module(module), name)
    except Exception:  # noqa: BLE001 - absent, or broken by its own missing dependencies
        return object


_VecEnvBase = _optional_base("stable_baselines3.common.vec_env.base_vec_env", "VecEnv")
_GymEnvBase = _optional_base("gym", "Env")
if _GymEnvBase is object:
    _GymEnvBase = _optional_base("gymnasium", "Env")


class _DevArray:
    """Minimal __cuda_array_interface__ carrier for a raw device pointer owned by the HIP library."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class _ReadOnlyInfo(dict):
    """The info dict of an env that has nothing to report in a step, shared by all such envs, steps and contexts (lazy_info).  SB3's VecEnv
    contract hands every env a dict of its own; a wrapper or callback that WRITES into the info of a running env would, with a shared plain
    dict, leak its key into every env and every later step - so writing raises and names the switch that restores per-env dicts."""
    __slots__ = ()

    def _ro(self, *a, **k):
        raise TypeError("this info dict is shared by every env that has nothing to report (lazy_info); call venv.set_lazy_info(False) to get a "
                        "fresh dict per env and step")
    __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = __ior__ = _ro


_EMPTY_INFO = _ReadOnlyInfo()      # (TactileVecEnv.step_wait)


class MonitorCsv:
    """stable_baselines3.common.monitor's file format (what its load_results reads): `#{json header}`, then a csv with columns r,l,t."""

    EXT = "monitor.csv"

    def __init__(self, monitor_dir, t_start, env_id=None, name="tactile_gym_hip"):
        import json
        import os
        os.makedirs(monitor_dir, exist_ok=True)
        self.path = os.path.join(monitor_dir, f"{name}.{self.EXT}")
        self._f = open(self.path, "w", newline="\n")
        self._f.write("#" + json.dumps({"t_start": t_start, "env_id": env_id}) + "\n")
        self._f.write("r,l,t\n")
        self._f.flush()

    def write(self, ep):
        self._f.write(f"{ep['r']},{ep['l']},{ep['t']}\n")
        self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None


class TactileVecEnv(_VecEnvBase):
    """N environments stepped by libtactile_gym_hip.so: a stable_baselines3 VecEnv (a subclass when SB3 is importable, the same API as a
    duck type otherwise)."""

    metadata = {"render.modes": ["rgb_array"]}

    def __init__(self, cfg, robot, sensor_desc, mesh_desc, observation_mode="tactile", obs_mode="numpy", seed=None, act_dim=None,
                 oracle_dim=10, feature_dim=0, copy_obs=True, scene_spec=None, guard_spec=None):
        self._L = capi.lib()
        self.num_envs = int(cfg.num_envs)
        self._cfg, self._robot, self._sensor, self._mesh = cfg, robot, sensor_desc, mesh_desc
        self.observation_mode = observation_mode
        if observation_mode not in ("oracle", "tactile", "visual", "visuotactile", "tactile_and_feature", "visual_and_feature",
                                    "visuotactile_and_feature"):
            raise SystemExit(f"Incorrect observation mode specified: {observation_mode}")  # base_tactile_env.py:264
        self._visual = "visual" in observation_mode or "visuo" in observation_mode
        # scene_spec = {"arm_type", "camera": (target, distance, yaw, pitch, fov, near, far), "body_rgb"}: what get_visual_obs needs
        # (base_tactile_env.py:212-245); None: this env's scene is not built (the surface envs' textured heightfield)
        self._scene_spec, self._scene = scene_spec, None
        if self._visual and scene_spec is None:
            raise NotImplementedError("visual (RGB scene camera) observations are not built for this env")
        self.obs_mode = obs_mode
        # copy_obs=True (default): every observation batch handed out is an array of its own, like the reference's fresh arrays
        # (base_tactile_env.py:247-282).  copy_obs=False: the device -> host copy lands in one of four rotating host buffers, valid for
        # the next three steps - for consumers that copy on arrival anyway (SB3 rollout / replay buffers, VecFrameStack).
        self.copy_obs = bool(copy_obs)
        self._pinned_stream = False      # TorchShard(pipelined=True) owns the stream choice
        self._ctx = C.c_void_p()
        capi.check(self._L.tg_create(C.byref(cfg), C.byref(robot), C.byref(sensor_desc.struct),
                                     C.byref(mesh_desc.struct) if mesh_desc is not None else None, C.byref(self._ctx)))
        self.H, self.W = sensor_desc.struct.image_h, sensor_desc.struct.image_w
        self.ndof = robot.ndof
        self.act_dim = act_dim if act_dim is not None else {0: 2, 1: 3, 2: 3, 3: 4}[cfg.movement_mode]
        self.action_space = spaces.Box(low=cfg.min_action, high=cfg.max_action, shape=(self.act_dim,), dtype=np.float32)
        obs_spaces = {}
        if "oracle" in observation_mode:
            obs_spaces["oracle"] = spaces.Box(low=-np.inf, high=np.inf, shape=(oracle_dim,), dtype=np.float32)
            capi.check(self._L.tg_enable_oracle_obs(self._ctx))   # written by every step / reset; the step's own copy is the terminal observation
        if "tactile" in observation_mode:
            obs_spaces["tactile"] = spaces.Box(low=0, high=255, shape=(self.H, self.W, 1), dtype=np.uint8)
        if self._visual:
            obs_spaces["visual"] = spaces.Box(low=0, high=255, shape=(self.H, self.W, 3), dtype=np.uint8)   # rgb_image_size = image_size
            self._set_scene(every_step=True)
        if "feature" in observation_mode:
            obs_spaces["extended_feature"] = spaces.Box(low=-np.inf, high=np.inf, shape=(feature_dim,), dtype=np.float32)
        self.feature_dim, self._oracle_dim = feature_dim, oracle_dim
        self.observation_space = spaces.Dict(obs_spaces)
        if _VecEnvBase is not object:     # SB3's constructor records num_envs / spaces (and render_mode in recent versions)
            try:
                _VecEnvBase.__init__(self, self.num_envs, self.observation_space, self.action_space)
            except TypeError:             # an SB3 whose VecEnv.__init__ takes other arguments: the attributes above are what it would set
                pass
        self._actions = np.zeros((self.num_envs, self.act_dim), dtype=np.float32)
        self._reward = np.zeros(self.num_envs, dtype=np.float32)
        self._done = np.zeros(self.num_envs, dtype=np.uint8)
        self._obs_host = np.zeros((self.num_envs, self.H, self.W, 1), dtype=np.uint8)
        self._term_host = None
        self._closed = False
        self._views = {}
        self._t_start = time.time()                 # Monitor's t_start (info["episode"]["t"])
        self._lazy_info, self._monitor = True, None
        self._obs_guard, self._guard_sum = False, {}
        self._ep_ret = np.zeros(self.num_envs, dtype=np.float32)
        self._ep_len = np.zeros(self.num_envs, dtype=np.int32)
        self._rebinds = 0
        # broadphase guard (broadphase.py; tg_set_broadphase): guard_spec = {"arm_type", "t_s_core", "edge" | "obj" | "ball_radius", "every_step"};
        # every_step None = the env's default: on for the contact envs (a 3 us launch in a 1.9 ms step), off elsewhere (check_broadphase() on demand).
        # TG_BROADPHASE_GUARD=1 / 0 forces it on every step / off for every env (the device test-suite runs with 1).
        self._guard_spec, self._guard = guard_spec, None
        if guard_spec is not None and os.environ.get("TG_BROADPHASE_GUARD", "") != "0":
            every = guard_spec.get("every_step")
            if os.environ.get("TG_BROADPHASE_GUARD", "") == "1":
                every = True
            self.set_broadphase_guard(bool(every))
        if seed is not None:
            self.seed(seed)

    # ------------------------------------------------------------------ VecEnv API
    def seed(self, seed=None):
        """env i gets seed + i (SB3 make_vec_env convention; reference base_tactile_env.py:61-64)."""
        base = 0 if seed is None else int(seed)
        seeds = (np.arange(self.num_envs, dtype=np.uint64) + np.uint64(base)).astype(np.uint64)
        capi.check(self._L.tg_seed(self._ctx, seeds.ctypes.data_as(C.POINTER(C.c_uint64)), self.num_envs))
        return [base + i for i in range(self.num_envs)]

    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert m.shape == (self.num_envs,)
        self._bind_torch_stream()
        capi.check(self._L.tg_reset(self._ctx, m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None))
        capi.check(self._L.tg_sync(self._ctx))
        if self._obs_guard:
            self._guard_after_step()
        return self._observation()

    def _bind_torch_stream(self):
        """obs_mode="torch": the zero-copy observation / reward / done tensors and CUDA action tensors are produced and consumed on
        torch's CURRENT stream, so the library is put on that stream before work is enqueued (tg_set_stream is a pointer swap; with
        TG_STEP_GRAPH=1 the step graph is captured on a stream of the library's own and replays on whichever stream is bound).  Ordering between the policy's kernels and the env's is then the
        stream's own: no event, no host wait, and correct under non-default or per-thread torch streams as well."""
        if self.obs_mode != "torch" or self._pinned_stream:
            return
        import torch
        ptr = torch.cuda.current_stream(torch.device("cuda", self._cfg.device)).cuda_stream
        if getattr(self, "_bound_stream", None) != ptr:
            if hasattr(self, "_bound_stream"):
                capi.check(self._L.tg_sync(self._ctx))     # drain the stream being left before work goes to another one
                self._rebinds += 1
                if self._rebinds == 8:                     # every switch costs a host sync of the stream being left
                    warnings.warn("TactileVecEnv (obs_mode='torch') has been moved between torch streams 8 times: drive an env from ONE stream "
                                  "(or pin it with set_stream); each switch drains the stream being left", RuntimeWarning, stacklevel=3)
            capi.check(self._L.tg_set_stream(self._ctx, C.c_void_p(ptr)))
            self._bound_stream = ptr

    def step_async(self, actions):
        """actions: numpy float32 [N, act_dim], or a torch CUDA tensor (read in place on the current torch stream)."""
        self._bind_torch_stream()
        if self._obs_guard:
            self._guard_before_step()
        if hasattr(actions, "data_ptr") and getattr(actions, "is_cuda", False):
            assert actions.dtype.is_floating_point and actions.element_size() == 4 and actions.is_contiguous()
            assert tuple(actions.shape) == (self.num_envs, self.act_dim)
            self._held = actions  # keep alive until step_wait
            capi.check(self._L.tg_step(self._ctx, C.c_void_p(actions.data_ptr()), 1))
        else:
            np.copyto(self._actions, np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.act_dim))
            capi.check(self._L.tg_step(self._ctx, C.c_void_p(self._actions.ctypes.data), 0))
        if self._obs_guard:
            self._guard_after_step()

    def set_lazy_info(self, on):
        """False: a fresh info dict per env and step (stable_baselines3's DummyVecEnv behaviour) instead of one shared empty dict for the envs that
        did not finish."""
        self._lazy_info = bool(on)

    def set_obs_guard(self, on=True):
        """Debug switch for the read-only contract of obs_mode="torch" (the block raster rewrites only the image blocks that change):
        the tactile buffer is checksummed after every step and checked before the next one; a caller that wrote into the tensor it was handed
        (in-place augmentation) gets a RuntimeError instead of silently corrupted later frames.  Costs a device reduction and a host
        synchronisation per step."""
        self._obs_guard, self._guard_sum = bool(on), {}     # per render target (rank 0's direct mode alternates two of them)

    def _guard_checksum(self):
        t = self.tactile_torch()
        return int(t.reshape(-1).view(__import__("torch").int64).sum().item())

    def _guard_before_step(self):
        sel = getattr(self, "_obs_sel", 0)          # the target this step draws into: its buffer must still hold what the library last drew there
        if self._obs_guard and self.obs_mode == "torch" and sel in self._guard_sum and self._guard_checksum() != self._guard_sum[sel]:
            raise RuntimeError("the tactile observation tensor handed out by the last step / reset was modified in place: obs_mode='torch' tensors "
                               "alias the library's buffer, of which only the changed 16 x 16 blocks are rewritten per step (DESIGN.md 4.2).  Copy "
                               "before augmenting in place, or run with TG_RASTER_REWRITE_ALL=1")

    def _guard_after_step(self):
        if self._obs_guard and self.obs_mode == "torch":
            self._guard_sum[getattr(self, "_obs_sel", 0)] = self._guard_checksum()

    def set_monitor(self, monitor_dir, env_id=None):
        """What `make_vec_env(..., monitor_dir=d)` asks for (sb3_helpers/rl_utils.py:22, 59; read back by stable_baselines3's load_results, which
        sb3_helpers/rl_plot_utils.py and custom_callbacks.py call): a Monitor csv in `monitor_dir` - one file for the whole batch,
        `tactile_gym_hip.monitor.csv`, header `#{"t_start": ..., "env_id": ...}`, columns r,l,t, one row per finished episode (the device-side
        episode statistics of info["episode"]), flushed per row like SB3's Monitor."""
        if self._monitor is not None:
            self._monitor.close()
        self._monitor = MonitorCsv(monitor_dir, self._t_start, env_id) if monitor_dir else None

    def bank_stats(self):
        """Reset bank: {"mode": "off" | "on" | "sync", "swapped": auto-resets that took a precomputed entry, "late": resets done on the spot}."""
        sw, late, mode = C.c_int64(), C.c_int64(), C.c_int32()
        capi.check(self._L.tg_get_bank_stats(self._ctx, C.byref(sw), C.byref(late), C.byref(mode)))
        return {"mode": ("off", "on", "sync", "template")[mode.value], "swapped": int(sw.value), "late": int(late.value)}

    def step_mode(self):
        """"fused": the env step is one launch (csrc/tg_fused.hip: the wavefront that steps an env resets and draws it; `fused_step`); "separate":
        step, reset and render are launches of their own."""
        mode, epw = C.c_int32(), C.c_int32()
        capi.check(self._L.tg_get_step_mode(self._ctx, C.byref(mode), C.byref(epw)))
        return "fused" if mode.value else "separate"

    def sample_actions(self, out, seed, counter):
        """action_space.sample() for the whole batch on the device (tg_sample_actions): fills the torch CUDA float32 tensor `out`
        [N, act_dim] with U[min_action, max_action) draws that depend on (seed, counter, element) only; enqueued on the env's stream."""
        assert out.is_cuda and out.is_contiguous() and out.element_size() == 4 and tuple(out.shape) == (self.num_envs, self.act_dim)
        capi.check(self._L.tg_sample_actions(self._ctx, C.c_uint64(seed), C.c_uint64(counter), C.c_void_p(out.data_ptr())))
        return out

    def step_random_async(self, seed, first_draw=0, restart=False):
        """One step of a random-action rollout, `step(action_space.sample())` for the whole batch, with the policy inside the step: the uniform draw (draw k =
        sample_actions(seed, k)) is made by the step kernel itself (or by the step's first launch) and the draw counter lives on the device (tg_step_random).  restart: the next draw
        is first_draw + 1.  The actions used are `actions_torch()`."""
        self._bind_torch_stream()
        if self._obs_guard:
            self._guard_before_step()
        capi.check(self._L.tg_step_random(self._ctx, C.c_uint64(seed), C.c_uint64(first_draw), 1 if restart else 0))
        if self._obs_guard:
            self._guard_after_step()

    def actions_torch(self):
        """The context's own action buffer as a float32 [N, act_dim] device tensor (what step_random_async drew)."""
        if "act" not in self._views:
            import torch
            p = C.c_void_p()
            capi.check(self._L.tg_get_actions(self._ctx, C.byref(p)))
            self._views["act"] = torch.as_tensor(_DevArray(p.value, (self.num_envs, self.act_dim), "<f4"), device=f"cuda:{self._cfg.device}")
        return self._views["act"]

    def step_wait(self):
        td = getattr(self, "_tile_download", None)
        if td is not None and "tactile" in self.observation_mode and self.obs_mode != "torch":
            # tile download: the observation fetch brings reward / done along under its one synchronisation
            td.rd_fresh = False
            obs = self._observation()
            if td.rd_fresh:
                np.copyto(self._reward, td.rew_host); np.copyto(self._done, td.done_host)
            else:
                capi.check(self._L.tg_get_reward_done(self._ctx, self._reward.ctypes.data_as(C.POINTER(C.c_float)),
                                                      self._done.ctypes.data_as(C.POINTER(C.c_uint8))))
            self._held = None
        else:
            capi.check(self._L.tg_get_reward_done(self._ctx, self._reward.ctypes.data_as(C.POINTER(C.c_float)),
                                                  self._done.ctypes.data_as(C.POINTER(C.c_uint8))))
            self._held = None
            obs = self._observation()
        dones = self._done.astype(bool)
        # SB3 reads infos[i].get(...) / "key" in infos[i]; only the envs that finished carry anything.  lazy_info (default): the others share ONE
        # empty dict (1024 dict constructions per step are a tenth of the numpy step's host time); an env's own dict is made when it has something to
        # say.  lazy_info=False restores a fresh dict per env for callers that write into the infos of running envs.
        infos = [_EMPTY_INFO] * self.num_envs if self._lazy_info else [{} for _ in range(self.num_envs)]
        # tile download (round 6): the finished envs' ids, episode statistics and terminal images came along with the observation fetch
        # (host_tiles.TileDownload.done_rows; tg_pack_done_rows) - no further round trip to the device in this step
        dr = td.done_rows if (td is not None and td.rd_fresh and "tactile" in self.observation_mode and self.obs_mode != "torch") else None
        if dr is not None and not np.array_equal(dr[0], np.nonzero(dones)[0]):
            dr = None                                 # (cannot happen: the same done flags; the copies below are always right)
        if dones.any():
            # what the reference's callers read from the Monitor wrapper around every env (sb3_helpers/rl_utils.py:17-30, 59; SB3's logger and
            # EvalCallback consume info["episode"]): return and length of the episode that just ended, added up on the device
            t = round(time.time() - self._t_start, 6)
            if dr is not None:
                stats = zip(dr[0].tolist(), dr[1].tolist(), dr[2].tolist())
            else:
                capi.check(self._L.tg_copy_episode_stats(self._ctx, self._ep_ret.ctypes.data_as(C.POINTER(C.c_float)),
                                                         self._ep_len.ctypes.data_as(C.POINTER(C.c_int32))))
                stats = ((int(i), float(self._ep_ret[i]), int(self._ep_len[i])) for i in np.nonzero(dones)[0])
            for i, r, l in stats:
                infos[i] = {"episode": {"r": round(float(np.float32(r)), 6), "l": int(l), "t": t}}
                if self._monitor is not None:
                    self._monitor.write(infos[i]["episode"])
        if self._cfg.auto_reset and dones.any():
            idx = np.nonzero(dones)[0]
            if self.obs_mode == "torch":
                term = self._terminal_observation()
                for i in idx:   # owned copies: the library's terminal buffers are rewritten by the next auto-reset
                    infos[i]["terminal_observation"] = {k: v[i].clone() for k, v in term.items()}
                    infos[i]["TimeLimit.truncated"] = False
            else:
                # only the finished envs' images cross PCIe (tg_copy_obs_rows): with the episodes out of phase some env finishes in nearly every
                # step, and the whole terminal batch per such step (16.8 MB) was two thirds of the step's time (tools/pcie_rate.py --staggered)
                term = self._terminal_rows(idx, tactile_rows=dr[3] if dr is not None else None)
                for j, i in enumerate(idx):
                    infos[i]["terminal_observation"] = {k: v[j] for k, v in term.items()}      # rows of arrays made for this step: owned
                    infos[i]["TimeLimit.truncated"] = False
        return obs, self._reward.copy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if not self._closed:
            self._closed = True
            if self._monitor is not None:
                self._monitor.close()
            if getattr(self, "_tile_download", None) is not None:
                self._tile_download.close()
                self._tile_download = None
            self._L.tg_destroy(self._ctx)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        idx = range(self.num_envs) if indices is None else ([indices] if np.isscalar(indices) else indices)
        return [getattr(self, attr_name) for _ in idx]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        idx = range(self.num_envs) if indices is None else ([indices] if np.isscalar(indices) else indices)
        return [getattr(self, method_name)(*args, **kwargs) for _ in idx]

    def _set_scene(self, every_step):
        from .robot_model import SceneDesc
        sp = self._scene_spec
        body = None if self._mesh is None else (self._mesh.verts, self._mesh.tris)
        scene = SceneDesc(sp["arm_type"], self._sensor.t_s_type, self._sensor.t_s_name, self._robot.ndof, (self.H, self.W), sp["camera"],
                          body, sp.get("body_rgb", (0, 0, 255)), every_step, body_heightfield=self._cfg.env_kind == capi.ENV_SURFACE_FOLLOW_AUTO)
        capi.check(self._L.tg_set_scene(self._ctx, C.byref(scene.struct)))
        self._scene = scene                         # only a scene the library accepted counts as set

    def get_images(self):
        """One render() frame per env (BaseTactileEnv.render, base_tactile_env.py:284-303): [H, 2W, 3] uint8, the scene camera's rgb image
        beside the tactile image as grey rgb.  Envs without a built scene: the tactile image alone, [H, W, 3]."""
        tac = np.repeat(self.tactile_numpy()[..., :1], 3, axis=3)
        if self._scene_spec is None:
            return list(tac)
        if self._scene is None:
            self._set_scene(every_step=False)
        if not self._visual:
            capi.check(self._L.tg_render_scene(self._ctx))
        return list(np.concatenate([self.visual_numpy(), tac], axis=2))

    def render(self, mode="rgb_array"):
        """The per-env render() frames tiled into one image (SB3 VecEnv.render)."""
        if mode != "rgb_array":
            return np.array([])
        imgs = self.get_images()
        h, w = imgs[0].shape[:2]
        cols = int(np.ceil(np.sqrt(self.num_envs)))
        rows = int(np.ceil(self.num_envs / cols))
        canvas = np.zeros((rows * h, cols * w, 3), dtype=np.uint8)
        for i in range(self.num_envs):
            r, c = divmod(i, cols)
            canvas[r * h:(r + 1) * h, c * w:(c + 1) * w] = imgs[i]
        return canvas

    # ------------------------------------------------------------------ device / host views
    def sync(self):
        capi.check(self._L.tg_sync(self._ctx))

    def set_stream(self, hip_stream_ptr):
        """Pin all of this env's work to one HIP stream (TorchShard(pipelined=True)); None returns to following torch's current stream."""
        self._pinned_stream = hip_stream_ptr is not None
        capi.check(self._L.tg_set_stream(self._ctx, C.c_void_p(hip_stream_ptr)))
        self._bound_stream = hip_stream_ptr

    def tactile_device_ptr(self, terminal=False):
        p = C.c_void_p()
        fn = self._L.tg_get_terminal_obs if terminal else self._L.tg_get_obs_tactile
        capi.check(fn(self._ctx, C.byref(p)))
        return p.value

    def set_obs_targets(self, dev_ptrs):
        """Up to two caller-owned device buffers uint8 [N, H, W] the tactile observations can be drawn into instead of the context's own
        (tg_set_obs_targets: rank 0's blocks of the gathered batches, parallel.ShardedVecEnv); [] forgets them.  select_obs_target(k) picks
        where the next steps / resets draw: 0 own buffer, 1 / 2 the caller's."""
        arr = (C.c_void_p * max(len(dev_ptrs), 1))(*[C.c_void_p(int(p)) for p in dev_ptrs])
        capi.check(self._L.tg_set_obs_targets(self._ctx, len(dev_ptrs), arr))
        self._obs_sel = 0
        for k in [k for k in self._views if isinstance(k, tuple) and k[0] == "obs" and k[1] != 0]:
            del self._views[k]

    def select_obs_target(self, index):
        if index != getattr(self, "_obs_sel", 0):
            capi.check(self._L.tg_select_obs_target(self._ctx, int(index)))
            self._obs_sel = int(index)

    def tactile_torch(self, terminal=False):
        """Zero-copy torch.uint8 [N,H,W,1] view of the device observation buffer (of the selected render target)."""
        key = "term" if terminal else ("obs", getattr(self, "_obs_sel", 0))
        if key not in self._views:   # the library's buffers never move: build each aliasing tensor once
            import torch
            arr = _DevArray(self.tactile_device_ptr(terminal), (self.num_envs, self.H, self.W, 1), "|u1")
            self._views[key] = torch.as_tensor(arr, device=f"cuda:{self._cfg.device}")
        return self._views[key]

    def reward_done_torch(self):
        if "rd" not in self._views:
            import torch
            r, d = C.c_void_p(), C.c_void_p()
            capi.check(self._L.tg_get_reward_done_dev(self._ctx, C.byref(r), C.byref(d)))
            dev = f"cuda:{self._cfg.device}"
            self._views["rd"] = (torch.as_tensor(_DevArray(r.value, (self.num_envs,), "<f4"), device=dev),
                                 torch.as_tensor(_DevArray(d.value, (self.num_envs,), "|u1"), device=dev))
        return self._views["rd"]

    def packed_torch(self):
        """Zero-copy torch.uint8 view of the whole per-step output block [obs | pad | reward f32 | done u8 | pad | feature f32[N][12]],
        the reward offset and the feature offset (-1: this env has no extended_feature)."""
        if "packed" not in self._views:
            import torch
            p, ob, tot, fo, fd = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
            capi.check(self._L.tg_get_packed_outputs(self._ctx, C.byref(p), C.byref(ob), C.byref(tot)))
            capi.check(self._L.tg_get_packed_feature(self._ctx, C.byref(fo), C.byref(fd)))
            self._views["packed"] = (torch.as_tensor(_DevArray(p.value, (tot.value,), "|u1"), device=f"cuda:{self._cfg.device}"), ob.value, fo.value)
        return self._views["packed"]

    def tactile_numpy(self, terminal=False):
        """Host copy of the observation batch.  copy_obs=True: a new array per call.  copy_obs=False: the device -> host copy lands in
        one of four rotating host buffers (an extra 16.8 MB `.copy()` per step costs more than the PCIe transfer itself): the array stays
        untouched for the next three steps."""
        if not terminal and getattr(self, "_tile_download", None) is not None:
            return self._tile_download.fetch()
        if terminal or self.copy_obs:
            buf = np.empty_like(self._obs_host)
        else:
            self._obs_ring_i = (getattr(self, "_obs_ring_i", -1) + 1) % 4
            if not hasattr(self, "_obs_ring"):
                self._obs_ring = [self._obs_host] + [np.zeros_like(self._obs_host) for _ in range(3)]
            buf = self._obs_ring[self._obs_ring_i]
        capi.check(self._L.tg_copy_obs_tactile(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint8)), int(terminal)))
        return buf

    def set_obs_transfer(self, how):
        """How `obs_mode="numpy"` observations cross PCIe.  "full" (default): the whole batch every step.  "tiles": only the 16 x 16 tiles
        that differ from the untouched sensor's image, rebuilt on the host (host_tiles.py; lossless; needs torch and lib/libtg_host.so);
        the batch handed out is then one of five ring buffers, untouched for the next three steps (as with copy_obs=False).  Terminal
        observations always take the full copy."""
        old = getattr(self, "_tile_download", None)
        if old is not None:
            old.close()
        if how == "full":
            self._tile_download = None
        elif how == "tiles":
            from .host_tiles import TileDownload
            self._tile_download = TileDownload(self)
        else:
            raise ValueError(f"obs_transfer {how!r}: 'full' or 'tiles'")
        return self

    def visual_torch(self, terminal=False):
        """Zero-copy torch.uint8 [N, H, W, 3] view of the device-resident scene-camera images."""
        key = ("vis", bool(terminal))
        if key not in self._views:
            import torch
            p = C.c_void_p()
            capi.check(self._L.tg_get_obs_visual(self._ctx, C.byref(p), int(terminal)))
            self._views[key] = torch.as_tensor(_DevArray(p.value, (self.num_envs, self.H, self.W, 3), "|u1"), device=f"cuda:{self._cfg.device}")
        return self._views[key]

    def visual_numpy(self, terminal=False):
        buf = np.empty((self.num_envs, self.H, self.W, 3), dtype=np.uint8)
        capi.check(self._L.tg_copy_obs_visual(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint8)), int(terminal)))
        return buf

    def _observation(self):
        obs = {}
        if "oracle" in self.observation_mode:
            obs["oracle"] = self.oracle_obs()
        if "tactile" in self.observation_mode:
            obs["tactile"] = self.tactile_torch() if self.obs_mode == "torch" else self.tactile_numpy()
        if self._visual:
            obs["visual"] = self.visual_torch() if self.obs_mode == "torch" else self.visual_numpy()
        if "feature" in self.observation_mode:
            obs["extended_feature"] = self.feature_torch() if self.obs_mode == "torch" else self.feature_numpy()
        return obs

    def oracle_terminal(self):
        """The oracle vectors of the last step before the auto-reset (rows valid where done); observation_mode "oracle" only."""
        if self.obs_mode == "torch":
            if "oracle_term" not in self._views:
                import torch
                p = C.c_void_p()
                capi.check(self._L.tg_get_obs_oracle_terminal(self._ctx, C.byref(p)))
                self._views["oracle_term"] = torch.as_tensor(_DevArray(p.value, (self.num_envs, self._oracle_dim), "<f4"), device=f"cuda:{self._cfg.device}")
            return self._views["oracle_term"]
        buf = np.empty((self.num_envs, self._oracle_dim), dtype=np.float32)
        capi.check(self._L.tg_copy_obs_oracle_terminal(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_float))))
        return buf

    def _image_rows(self, idx, visual, terminal=True):
        """[len(idx), H, W, C] uint8: the tactile (or scene-camera) images of the envs `idx` from the terminal (or current) observation buffer."""
        ids = np.ascontiguousarray(idx, dtype=np.int32)
        shape = (self.H, self.W, 3) if visual else (self.H, self.W, 1)
        out = np.empty((len(ids),) + shape, dtype=np.uint8)
        capi.check(self._L.tg_copy_obs_rows(self._ctx, 1 if visual else 0, 1 if terminal else 0, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids),
                                            out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def _terminal_rows(self, idx, tactile_rows=None):
        """The terminal observation of the envs `idx` only (numpy): images by tg_copy_obs_rows (or `tactile_rows`, what the tile download brought along),
        the small per-env vectors from their whole-batch copies."""
        obs = {}
        if "oracle" in self.observation_mode:
            obs["oracle"] = np.array(self.oracle_terminal()[idx])
        if "tactile" in self.observation_mode:
            obs["tactile"] = tactile_rows if tactile_rows is not None else self._image_rows(idx, False)
        if self._visual:
            obs["visual"] = self._image_rows(idx, True)
        if "feature" in self.observation_mode:
            obs["extended_feature"] = np.array(self.feature_numpy(True)[idx])
        return obs

    def _terminal_observation(self):
        obs = {}
        if "oracle" in self.observation_mode:
            obs["oracle"] = self.oracle_terminal()
        if "tactile" in self.observation_mode:
            obs["tactile"] = self.tactile_torch(True) if self.obs_mode == "torch" else self.tactile_numpy(True)
        if self._visual:
            obs["visual"] = self.visual_torch(True) if self.obs_mode == "torch" else self.visual_numpy(True)
        if "feature" in self.observation_mode:
            obs["extended_feature"] = self.feature_torch(True) if self.obs_mode == "torch" else self.feature_numpy(True)
        return obs

    def _feature_ptr(self, terminal):
        p, dim = C.c_void_p(), C.c_int32()
        capi.check(self._L.tg_get_obs_feature(self._ctx, C.byref(p), C.byref(dim), int(terminal)))
        return p.value, dim.value

    def feature_torch(self, terminal=False):
        """Zero-copy torch.float32 [N, feature_dim] view of the device-resident extended_feature observation."""
        key = ("feat", bool(terminal))
        if key not in self._views:
            import torch
            ptr, dim = self._feature_ptr(terminal)
            self._views[key] = torch.as_tensor(_DevArray(ptr, (self.num_envs, dim), "<f4"), device=torch.device("cuda", self._cfg.device))
        return self._views[key]

    def feature_numpy(self, terminal=False):
        if self.feature_dim == 0:
            return np.zeros((self.num_envs, 0), dtype=np.float32)
        buf = np.zeros((self.num_envs, self.feature_dim), dtype=np.float32)
        capi.check(self._L.tg_copy_obs_feature(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_float)), int(terminal)))
        return buf

    def oracle_obs(self):
        """get_oracle_obs for the whole batch, computed on the device from the current state (tg_get_obs_oracle): float32 [N, dim];
        a zero-copy torch view in obs_mode "torch" (valid until the next call), an owned numpy array otherwise.  The env classes'
        oracle_obs_host() is the same vector from a state read-back (kept as a cross-check)."""
        if self.obs_mode == "torch":
            if "oracle" not in self._views:
                import torch
                p, d = C.c_void_p(), C.c_int32()
                capi.check(self._L.tg_get_obs_oracle(self._ctx, C.byref(p), C.byref(d)))
                self._views["oracle"] = torch.as_tensor(_DevArray(p.value, (self.num_envs, d.value), "<f4"), device=f"cuda:{self._cfg.device}")
            else:
                capi.check(self._L.tg_get_obs_oracle(self._ctx, C.byref(C.c_void_p()), None))
            return self._views["oracle"]
        buf = np.empty((self.num_envs, self._oracle_dim), dtype=np.float32)
        capi.check(self._L.tg_copy_obs_oracle(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_float))))
        return buf

    def oracle_obs_host(self):
        raise NotImplementedError

    def _workframe(self):
        from .pb_math import WorkFrame
        if not hasattr(self, "_wf"):
            self._wf = WorkFrame([self._cfg.workframe_pos[k] for k in range(3)], [self._cfg.workframe_rpy[k] for k in range(3)])
        return self._wf

    def _tcp_workframe_state(self, st):
        """get_current_TCP_pos_vel_workframe (base_robot_arm.py:153-172) for every env: pos, rpy, orn (quaternion), linear and angular
        velocity of the TCP frame in the work frame, from the device state read-back (host side; the tactile path does not need it)."""
        from . import hip_ops, pb_math as pm
        wf = self._workframe()
        J, pos, rot = hip_ops.jacobian_tcp(self._robot, st["q"], dtype="f64")
        rpy_world = pm.euler_from_quat(pm.quat_from_mat(rot))
        p, rpy = wf.pose(pos, rpy_world)
        lin = wf.vec(np.einsum("nij,nj->ni", J[:, :3, :], st["qd"]))
        ang = wf.vec(np.einsum("nij,nj->ni", J[:, 3:, :], st["qd"]))
        return p, rpy, pm.quat_from_euler(rpy), lin, ang

    def _obj_workframe_state(self, st):
        """get_obj_pos_workframe / get_obj_vel_workframe (base_object_env.py:118-139): object base pose and velocity in the work frame."""
        from . import pb_math as pm
        wf = self._workframe()
        rpy_world = pm.euler_from_quat(pm.quat_from_mat(st["body_rot"]))
        p, rpy = wf.pose(st["body_pos"], rpy_world)
        return p, rpy, pm.quat_from_euler(rpy), wf.vec(st["body_linvel"]), wf.vec(st["body_angvel"])

    # ------------------------------------------------------------------ parity / inspection
    def set_broadphase_guard(self, every_step=False):
        """(Re)install the broadphase guard: the check that no pair of collision objects other than the ones the solver has rows for can touch
        (what PyBullet's broadphase would find, robots/arms/robot.py:141; include/tactile_gym_hip.h: tg_set_broadphase).  every_step: the check is a
        node of every step's graph; otherwise check_broadphase() runs it on demand.  Results: get_state()["broadphase_pairs" / "_hits" / "_mask"]."""
        from .broadphase import Guard
        sp = self._guard_spec
        if sp is None:
            raise NotImplementedError("this env has no broadphase guard scene")
        self._guard = Guard(sp["arm_type"], self._sensor.t_s_name, self._sensor.t_s_type, sp["t_s_core"], edge=sp.get("edge"), obj=sp.get("obj"),
                            ball_radius=sp.get("ball_radius"), every_step=every_step)
        capi.check(self._L.tg_set_broadphase(self._ctx, C.byref(self._guard.struct)))

    def check_broadphase(self):
        """One check of the current state; returns (pairs, hits, mask) int32 [N] each (see set_broadphase_guard)."""
        if self._guard is None:
            self.set_broadphase_guard(False)
        capi.check(self._L.tg_check_broadphase(self._ctx))
        st = self.get_state()
        return st["broadphase_pairs"], st["broadphase_hits"], st["broadphase_mask"]

    def broadphase_totals(self):
        """{"env_checks", "pairs", "hits"} since the guard was installed: env states checked, unexpected pairs Bullet's broadphase would have handed to
        its narrowphase (world AABBs overlap), and those that survive the oriented-box / hull tests (0 = no unmodelled contact was possible)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(self._L.tg_get_broadphase_totals(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return {"env_checks": a.value, "pairs": b.value, "hits": c.value}

    def get_state(self):
        """Host copy of the per-env state (tg_get_state)."""
        n, nd = self.num_envs, self.ndof
        out = dict(q=np.zeros((n, nd)), qd=np.zeros((n, nd)), qd_target=np.zeros((n, nd)), tcp_pos=np.zeros((n, 3)),
                   tcp_rpy=np.zeros((n, 3)), edge_ang=np.zeros(n), embed_dist=np.zeros(n), stim_xform=np.zeros((n, 12), np.float32),
                   step_count=np.zeros(n, np.int32), reset_ticks=np.zeros(n, np.int32), rng_state=np.zeros(n, np.uint64),
                   solver_sweeps=np.zeros(n, np.int32),   # threshold mode (solver_residual_threshold > 0): PGS sweeps of the last step's ticks
                   broadphase_pairs=np.zeros(n, np.int32), broadphase_hits=np.zeros(n, np.int32), broadphase_mask=np.zeros(n, np.int32))
        if self._cfg.env_kind == capi.ENV_OBJECT_BALANCE:
            out.update(body_pos=np.zeros((n, 3)), body_rot=np.zeros((n, 3, 3)), body_linvel=np.zeros((n, 3)), body_angvel=np.zeros((n, 3)),
                       gravity_z=np.zeros(n))
            if self._cfg.balance_object == capi.BALANCE_OBJECT["ball_on_plate"]:
                out.update(ball_pos=np.zeros((n, 3)), ball_linvel=np.zeros((n, 3)), ball_angvel=np.zeros((n, 3)), ball_impulse=np.zeros(n))
            if self._cfg.balance_object == capi.BALANCE_OBJECT["spinning_plate"]:   # the dish (tg_state_view.dish_state); body_* is the spool
                out.update(dish_state=np.zeros((n, 20)))
        if self._cfg.env_kind == capi.ENV_OBJECT_PUSH:
            out.update(body_pos=np.zeros((n, 3)), body_rot=np.zeros((n, 3, 3)), body_linvel=np.zeros((n, 3)), body_angvel=np.zeros((n, 3)),
                       traj=np.zeros((n, 3, capi.MAX_TRAJ_POINTS)), goal_id=np.zeros(n, np.int32), obj_mass=np.zeros(n))
        if self._cfg.env_kind == capi.ENV_OBJECT_ROLL:   # obj_mass: the episode's marble radius; goal_pos: the goal in the TCP frame
            out.update(body_pos=np.zeros((n, 3)), body_rot=np.zeros((n, 3, 3)), body_linvel=np.zeros((n, 3)), body_angvel=np.zeros((n, 3)),
                       goal_pos=np.zeros((n, 3)), obj_mass=np.zeros(n))
        if self._cfg.env_kind in (capi.ENV_OBJECT_PUSH, capi.ENV_OBJECT_ROLL):   # contact pairs of the last sim tick (tg_state_view)
            out.update(contact_count=np.zeros(n, np.int32), contact_ids=np.zeros((n, 8), np.int32))
        if self._cfg.env_kind == capi.ENV_SURFACE_FOLLOW_AUTO:
            out.update(goal_pos=np.zeros((n, 3)), direction=np.zeros((n, 2)), surf_zoff=np.zeros(n, np.float32),
                       heights=np.zeros((n, self._cfg.surf_rows, self._cfg.surf_cols)))
        v = capi.TgStateView()
        for k, a in out.items():
            ct = {np.dtype(np.float64): C.c_double, np.dtype(np.float32): C.c_float, np.dtype(np.int32): C.c_int32,
                  np.dtype(np.uint64): C.c_uint64}[a.dtype]
            setattr(v, k, a.ctypes.data_as(C.POINTER(ct)))
        capi.check(self._L.tg_get_state(self._ctx, C.byref(v)))
        return out

    def set_joint_state(self, q, qd):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(self.num_envs, self.ndof)
        qd = np.ascontiguousarray(qd, dtype=np.float64).reshape(self.num_envs, self.ndof)
        dp = C.POINTER(C.c_double)
        capi.check(self._L.tg_set_joint_state(self._ctx, q.ctypes.data_as(dp), qd.ctypes.data_as(dp)))

    def profile(self, enable=True):
        """True / 1: HIP events around every launch class (no graph) + the kernels' own clock; 2 / "clock": the own clock only, the step's launches stay
        as the rollout itself runs them; False: off."""
        capi.check(self._L.tg_profile_enable(self._ctx, 2 if enable == "clock" else int(enable)))

    def profile_get(self):
        """{class: (total ms, scopes)} of profiling mode.  "step", "render" (the one launch of a fused step), "reset" (the reset sequence),
        "render_masked", "scene", "empty_event_pair": HIP events on the launch stream - every figure carries what the empty pair measures.
        "<class>_clock": the same scopes by the kernels' own clock (csrc/tg_kt.hpp: first wavefront start -> last wavefront end)."""
        out = {}

        def get(which):
            ms, cnt = C.c_double(), C.c_int64()
            capi.check(self._L.tg_profile_get(self._ctx, which, C.byref(ms), C.byref(cnt)))
            return (ms.value, cnt.value)
        for which, name in enumerate(("step", "render", "reset", "render_masked", "scene", "empty_event_pair")):
            out[name] = get(which)
        for k, name in enumerate(("step", "render", "reset", "render_masked")):
            out[name + "_clock"] = get(8 + k)
        return out


class SingleTactileEnv(_GymEnvBase):
    """One environment with the reference's gym.Env surface (old 4-tuple API, base_tactile_env.py:166-185): reset / step / render / seed /
    close, action_space, observation_space, min_action / max_action - a 1-env TactileVecEnv underneath (auto-reset off: a gym.Env is reset
    by its caller).  Subclasses name their VecEnv class and the reference constructor they mirror.  Every array handed out is owned by
    the caller."""

    metadata = {"render.modes": ["rgb_array"]}
    vec_cls = None            # the TactileVecEnv subclass

    def __init__(self, max_steps=1000, image_size=(64, 64), env_modes=None, show_gui=False, show_tactile=False, physics_dtype="f64",
                 device=0, **kwargs):
        if show_gui or show_tactile:
            raise NotImplementedError("GUI / cv2 windows are not part of the headless device path")
        if env_modes is None:
            env_modes = self.default_env_modes
        self._vec = self.vec_cls(1, max_steps, image_size, env_modes, physics_dtype, auto_reset=False, device=device, **kwargs)
        # what HipVecEnv needs to build the N-env context this env stands for (make_vec_env hands it constructors, not arguments)
        self._ctor = dict(max_steps=max_steps, image_size=list(image_size), env_modes=dict(env_modes), physics_dtype=physics_dtype, device=device, **kwargs)
        self.action_space, self.observation_space = self._vec.action_space, self._vec.observation_space
        self.min_action, self.max_action = self._vec.min_action, self._vec.max_action
        self._max_steps, self._image_size, self._seed = max_steps, list(image_size), None

    @classmethod
    def make_vec(cls, num_envs, **kwargs):
        kwargs.pop("show_gui", None)
        kwargs.pop("show_tactile", None)
        return cls.vec_cls(num_envs, **kwargs)

    @staticmethod
    def _first(obs):
        return {k: (v[0].clone() if hasattr(v, "clone") else np.array(v[0])) for k, v in obs.items()}

    def seed(self, seed=None):
        self._seed = seed
        return self._vec.seed(seed)[:1]

    def reset(self):
        return self._first(self._vec.reset())

    def step(self, action):
        obs, rew, done, _ = self._vec.step(np.asarray(action, dtype=np.float32)[None])
        return self._first(obs), float(rew[0]), bool(done[0]), {}      # base_tactile_env.py:185 returns an empty info; a Monitor around this env adds its own

    def render(self, mode="rgb_array"):
        """BaseTactileEnv.render (base_tactile_env.py:284-303): the scene camera's rgb image beside the tactile image, [H, 2W, 3] uint8
        (no cv2 window: headless)."""
        if mode != "rgb_array":
            return np.array([])
        return self._vec.get_images()[0]

    def close(self):
        self._vec.close()

    def get_state(self):
        return {k: v[0] for k, v in self._vec.get_state().items()}


class HipVecEnv:
    """`vec_env_cls` for stable_baselines3's make_vec_env: the reference's `make_training_envs` / `make_eval_env`
    (sb3_helpers/rl_utils.py:15-37, 49-68) then change by one token,

        make_vec_env(env_id, env_kwargs=env_args, n_envs=n, seed=seed, vec_env_cls=tg.HipVecEnv, monitor_dir=save_dir)

    instead of `vec_env_cls=SubprocVecEnv`.  SB3 hands a vec_env_cls a list of CONSTRUCTORS - each would make one env (gym.make(env_id,
    **env_kwargs), seeded seed + rank, wrapped in Monitor) in its own process.  Here the first constructor is called once as a probe: the env
    it makes names its class and constructor arguments, and ONE N-env device context is built from them with env i seeded seed + i (the
    seeds the N constructors would have used).  The per-env Monitor's bookkeeping is done on the device: every step's info carries
    info["episode"] = {"r", "l", "t"} for the envs that finished (what SB3's logger and EvalCallback read), and with `monitor_dir` the same
    rows go to `<monitor_dir>/tactile_gym_hip.monitor.csv` in SB3's Monitor format (load_results(monitor_dir), which the reference's
    sb3_helpers/rl_plot_utils.py and custom_callbacks.py call, reads every *monitor.csv of the directory).  Extra keyword arguments
    (vec_env_kwargs: obs_mode, copy_obs, physics_dtype, device, obs_transfer ...) go to the vectorised constructor; start_method is accepted and
    ignored.  The result is a TactileVecEnv (an SB3 VecEnv subclass wherever SB3 is importable), not an instance of this class."""

    def __new__(cls, env_fns, start_method=None, **kwargs):
        env_fns = list(env_fns)
        if not env_fns:
            raise ValueError("HipVecEnv needs at least one env constructor")
        probe = env_fns[0]()
        single = probe
        def ours(e):                                 # one of this package's single-env classes (by what it carries, not by class identity)
            return hasattr(e, "_ctor") and hasattr(e, "_seed") and callable(getattr(type(e), "make_vec", None))

        for _ in range(16):                          # through Monitor / TimeLimit / any gym.Wrapper down to the env itself
            if ours(single):
                break
            nxt = getattr(single, "env", None)
            if nxt is None:
                nxt = getattr(single, "unwrapped", None)
                if nxt is single:
                    nxt = None
            if nxt is None:
                break
            single = nxt
        if not ours(single):
            raise TypeError(f"HipVecEnv: the constructors make {type(single).__name__}, not one of this package's envs "
                            f"(register ids with `import tactile_gym_amd` and pass one of tactile_gym_amd.registered_ids())")
        ctor = dict(single._ctor)
        obs_transfer = kwargs.pop("obs_transfer", None)   # vec_env_kwargs=dict(obs_transfer="tiles"): the tile-sparse observation download
        monitor_dir = kwargs.pop("monitor_dir", None)     # explicit; otherwise taken from the probe's Monitor wrapper (make_vec_env(monitor_dir=...))
        probe_csv = None
        rw = getattr(probe, "results_writer", None)       # stable_baselines3.common.monitor.Monitor
        fh = getattr(rw, "file_handler", None) or getattr(probe, "file_handler", None)
        if fh is not None and getattr(fh, "name", None):
            import os
            probe_csv = os.path.abspath(fh.name)
            if monitor_dir is None:
                monitor_dir = os.path.dirname(probe_csv)
        ctor.update(kwargs)
        seed = single._seed                          # make_vec_env's make_env(rank) called env.seed(seed + rank): rank 0 -> the base seed
        env_cls = type(single)
        probe.close()
        if probe_csv is not None:                    # the probe's own Monitor file holds a header and no episode: it would only dilute load_results
            import os
            try:
                with open(probe_csv) as f:
                    rows = sum(1 for _ in f)
                if rows <= 2:
                    os.remove(probe_csv)
            except OSError:
                pass
        venv = env_cls.make_vec(num_envs=len(env_fns), seed=seed, **ctor)
        if obs_transfer is not None:
            venv.set_obs_transfer(obs_transfer)
        if monitor_dir:
            venv.set_monitor(monitor_dir, env_id=getattr(getattr(single, "spec", None), "id", None) or env_cls.__name__)
        return venv
