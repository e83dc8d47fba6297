#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of edge_follow-v0 (UR5 + TacTip, 128x128 tactile obs), BASELINE.json configs[1].

    python bench.py --gpus 1 --steps 2000 --warmup 100      (the defaults: SURVEY 8d asks for 100 warm-up and >= 2000 timed steps)
    python bench.py --gpus N ...                            (no WORLD_SIZE in the environment: bench.py starts its own N ranks under
                                                             torch.distributed.run on 127.0.0.1 and rank 0 prints the line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one VecEnv.step() of the whole batch: controller + 24 sim ticks + tactile render for every env, random
actions ~ U[-0.25, 0.25) drawn on the device by tg_sample_actions (synthetic), auto-reset on (episodes of 200 steps, so resets fall
inside the timed region whenever K + W crosses a multiple of 200; reported separately via `resets_in_timed_region`).
Observations stay resident in HBM (device tensors); the PCIe-inclusive rate is quoted in DESIGN.md, never here.
The rollout is device resident: steps are enqueued on a torch stream (TorchShard(pipelined=True)) and their outputs consumed on it,
the host does not wait per step (--sync-steps restores the blocking VecEnv.step_wait path).
Extra fields: roofline (dominant kernel, HIP-event durations, PMC traffic), cpu_baseline (the CPU oracle on all host cores),
literal_solver (the same workload with exactly 150 PGS sweeps in every tick), without_full_batch_reset, other_configs (one short
companion run per remaining BASELINE config at 1024 envs on this GPU), roofline_16384 (the dominant kernel at 16 384 envs),
staggered_episodes (the headline workload with the envs' episodes out of phase, as in an RL run: some env finishes in nearly every step,
whereas the timed region starts all envs together and sees one full-batch reset per 200 steps; tools/desync_rate.py).
Weak scaling: every rank owns --num-envs envs; rank 0 receives all observations / rewards / dones once per step (--payload: what the
message carries, --transport: how it travels; parallel.py), started asynchronously so that it overlaps the next step's simulation
(SURVEY 8e); the last exchange is waited for inside the timed region.  `no_gather` is the same K steps without the exchange (per-rank
learners): compute scaling apart from the link bound.  With the default --transport auto --payload auto the transport is chosen by a
short probe of both (exchange.probe); an ipc exchange that reports an error falls back to the RCCL gather (exchange.fallback).
Prints ONE JSON line on rank 0, and nothing else on stdout (native libraries' writes to fd 1 are sent to stderr).
"""
import argparse
import gc
import contextlib
import hashlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SURF_MODES = dict(movement_mode="xyzRxRy", control_mode="TCP_velocity_control", noise_mode="simplex", observation_mode="tactile",
                  reward_mode="dense", arm_type="ur5", tactile_sensor_name="digit")   # BASELINE configs[2], params/surface_follow_auto_params.py
BAL_MODES = dict(movement_mode="xy", control_mode="TCP_velocity_control", object_mode="pole", rand_gravity=True, rand_embed_dist=True,
                 observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")   # configs[4], params/object_balance_params.py
PUSH_MODES = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
                  observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")   # configs[3], params/object_push_params.py
ROLL_MODES = dict(movement_mode="xy", control_mode="TCP_velocity_control", rand_init_obj_pos=True, rand_obj_size=True, rand_embed_dist=True,
                  observation_mode="tactile_and_feature", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")   # params/object_roll_params.py
VERT_MODES = dict(movement_mode="xRz", control_mode="TCP_velocity_control", noise_mode="vertical_simplex", observation_mode="tactile",
                  reward_mode="dense", arm_type="mg400", tactile_sensor_name="tactip")   # params/surface_follow_vert_params.py
MODES = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
             reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
ENV_MODES = {"edge_follow-v0": MODES, "surface_follow-v0": SURF_MODES, "object_balance-v0": BAL_MODES, "object_push-v0": PUSH_MODES,
             "object_roll-v0": ROLL_MODES, "surface_follow-v2": VERT_MODES}
MAX_STEPS = {"object_balance-v0": 250, "object_push-v0": 1000, "object_roll-v0": 250}     # params/*_params.py max_ep_len (default 200)
ALGO_BYTES_PER_ENV_STEP = 16600.0   # BASELINE.md section 3 / SURVEY 8(d): 16 384 B image + ~0.2 KB state/action/reward
ALGO_BYTES_SURFACE = 33000.0        # config 3: + the per-env 64x64 f32 heightfield read
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
TRAFFIC_FILE = os.path.join("profiles", "r6_traffic.json")


def algo_bytes(env_id, image_size):
    """compulsory HBM bytes per env step (image written once + state in/out; config 3 adds the heightfield read)."""
    return {"edge_follow-v0": ALGO_BYTES_PER_ENV_STEP if image_size == 128 else image_size * image_size + 216.0,
            "surface_follow-v0": image_size * image_size + 16616.0, "surface_follow-v2": image_size * image_size + 16616.0,
            "object_balance-v0": image_size * image_size + 300.0,             # config 5: 65.8 KB at 256x256
            "object_push-v0": image_size * image_size + 400.0, "object_roll-v0": image_size * image_size + 400.0}[env_id]


def source_hash():
    """sha256 (first 16 hex digits) over the sources the HIP library is built from: what a PMC measurement in profiles/ is valid for."""
    import glob
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "tactile_gym_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def _cpu_worker(idx, cpu, seconds, repeats, barrier, out_q):
    """One oracle env (edge_follow-v0, 128x128, random actions, resets included) pinned to one host CPU; `repeats` windows of `seconds`, every
    window started together with all other workers (barrier), so that a window's aggregate is what the box sustains with every core busy."""
    try:
        if cpu is not None:
            os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    try:                                                  # one CPU per process: numpy's BLAS / OpenMP pools must not start a thread per core of the box each
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:  # noqa: BLE001
        pass
    import numpy as np
    from oracle.ref_env import OracleEdgeFollowEnv
    env = OracleEdgeFollowEnv(seed=1 + idx, max_steps=200, image_size=(128, 128), env_modes=MODES)
    env.reset()
    rng = np.random.default_rng(1 + idx)
    for _ in range(5):                                    # page in code and assets before the first window
        env.step(rng.uniform(-0.25, 0.25, 2))
    out = []
    for _ in range(repeats):
        if barrier is not None:
            barrier.wait()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            _, _, done, _ = env.step(rng.uniform(-0.25, 0.25, 2))
            n += 1
            if done:
                env.reset()
        out.append((n, time.perf_counter() - t0))
    if out_q is not None:
        out_q.put((idx, out))
    return out


def pybullet_reference(seconds):
    """PyBullet itself on this box's host, when `import pybullet` works and TG_PYBULLET_ASSETS points at a checkout of the
    reference's assets directory (tools/pybullet_probe.py: raw pybullet, no tactile_gym source).  One process, one env: an env step =
    velocity-control set-up + 24 x (gravity compensation + stepSimulation) + one 128 x 128 getCameraImage of the in-sensor camera - the engine
    calls of BaseTactileEnv.step without the reference's Python around them (an upper bound on the reference's own rate)."""
    try:
        import pybullet  # noqa: F401
    except Exception:  # noqa: BLE001
        return "unavailable (import pybullet fails on this box)"
    assets = os.environ.get("TG_PYBULLET_ASSETS")
    if not assets or not os.path.isdir(assets):
        return "pybullet is importable but TG_PYBULLET_ASSETS does not point at the reference's assets directory: not timed"
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pybullet_probe as probe
        b = probe.PyBulletBackend(assets)
        b.reset_joints(probe.UR5_REST)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            b.motors_velocity(probe.JOINT_VEL if n % 2 == 0 else [-v for v in probe.JOINT_VEL])
            for _ in range(24):
                b.tick()
            b.depth_of_edge(128)
            n += 1
        dt = time.perf_counter() - t0
        return {"value": round(n / dt, 1), "unit": "env-steps/s", "cores": 1, "kind": "reference engine (raw pybullet, tools/pybullet_probe.py)",
                "sample": f"{n} env steps in {dt:.1f} s: UR5 + TacTip, 24 ticks per step with gravity compensation, one 128x128 getCameraImage per step"}
    except Exception as e:  # noqa: BLE001
        return f"pybullet present but the probe failed: {e!r}"


def _cgroup_cpu_quota():
    """CPUs' worth of time this container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(seconds=10.0, repeats=3):
    """The CPU oracle (oracle/: a port, not PyBullet — PyBullet is not installable here) on this box's host cores: one process per CPU of the
    affinity mask, each pinned to its CPU (os.sched_setaffinity) with its BLAS / OpenMP pools limited to one thread, `repeats` windows of
    `seconds` started on a common barrier.  value = the MEDIAN over the windows of (steps of all processes in the window / the window's longest
    process time) - one disturbed window (another tenant of the box) does not move it; the windows, their spread and the per-process rate are
    reported beside it (two runs of this must agree within 5 %)."""
    import multiprocessing as mp
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    quota = _cgroup_cpu_quota()
    note = ""
    if quota is not None and quota < len(cpus):
        # the container's CPU-time quota, not the CPU count, is what this process tree can use: more processes than that only take turns
        # (measured on the GPU box: 256 logical CPUs visible, 256 pinned processes ran at 3 % of the single-process rate each, 7.6x it in total)
        k = max(1, int(quota))
        note = f"; cgroup cpu quota {quota:.1f} of {len(cpus)} visible CPUs: {k} processes"
        cpus = cpus[::max(1, len(cpus) // k)][:k]
    cores = len(cpus)
    ctx = mp.get_context("fork")
    solo = _cpu_worker(0, cpus[0], min(4.0, seconds / 2), 1, None, None)[0]
    barrier, q = ctx.Barrier(cores), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(i, cpus[i], seconds, repeats, barrier, q), daemon=True) for i in range(cores)]
    for pr in procs:
        pr.start()
    res = [q.get() for _ in range(cores)]
    for pr in procs:
        pr.join()
    windows = []
    for r in range(repeats):
        windows.append(sum(out[r][0] for _, out in res) / max(out[r][1] for _, out in res))
    per_proc = [out[r][0] / out[r][1] for _, out in res for r in range(repeats)]
    mean = sorted(windows)[len(windows) // 2]
    pyb = pybullet_reference(min(seconds, 5.0))   # the real engine, if this box happens to have it
    return {"value": round(mean, 1), "unit": "env-steps/s", "cores": cores, "kind": "port", "pybullet_reference": pyb,
            "windows": [round(w, 1) for w in windows], "spread": round((max(windows) - min(windows)) / mean, 4),
            "per_process": {"mean": round(sum(per_proc) / len(per_proc), 2), "min": round(min(per_proc), 2), "max": round(max(per_proc), 2)},
            "sample": f"{cores} processes x 1 oracle env (edge_follow-v0, 128x128, random actions, resets included), one per CPU of the affinity mask and pinned to it, "
                      f"{repeats} windows of {seconds:.0f} s started on a common barrier; single pinned process alone: {solo[0] / solo[1]:.1f} env-steps/s{note}"}


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: become `python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>`
    (one rank per GPU, rendezvous on 127.0.0.1 at a free port).  Does not return."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.execv(sys.executable, argv)


class Workload:
    """One env configuration on this rank's GPU with its action sampler and stream."""

    def __init__(self, env_id, n, image_size, physics, rank, local_rank, full_sweeps=False, observation_mode=None, pipelined=True,
                 residual_threshold=0.0, **extra):
        import torch
        import tactile_gym_amd as tg
        from tactile_gym_amd.parallel import TorchShard
        self.torch, self.env_id, self.n, self.image_size, self.rank = torch, env_id, n, image_size, rank
        self.modes = dict(ENV_MODES[env_id])
        if observation_mode:
            self.modes["observation_mode"] = observation_mode
        self.act_dim = 3 if env_id == "surface_follow-v0" else 2
        self.max_steps = MAX_STEPS.get(env_id, 200)
        if env_id not in ("object_push-v0", "object_roll-v0"):
            extra = {}
        if residual_threshold:
            extra = dict(extra, solver_residual_threshold=residual_threshold)
        self.residual_threshold = residual_threshold
        self.ticks_per_step = 12 if env_id == "object_balance-v0" else 24
        self.venv = tg.make_vec(env_id, num_envs=n, max_steps=self.max_steps, image_size=[image_size, image_size], env_modes=self.modes,
                                seed=1 + rank * n, physics_dtype=physics, auto_reset=True, device=local_rank, obs_mode="torch",
                                pgs_full_sweeps=full_sweeps, **extra)
        # device-resident rollout: the step is enqueued on a torch stream and its outputs are consumed on that stream
        self.shard = TorchShard(self.venv, pipelined=pipelined)
        self.act_buf = torch.empty(n, self.act_dim, device=f"cuda:{local_rank}", dtype=torch.float32)
        self.draw = 0

    def actions(self):
        """action_space.sample() for the whole batch: U[-0.25, 0.25), one device kernel on the env's stream (tg_sample_actions, counter
        based: draw k of rank r depends on (1234 + r, k) only), inside the timed region like every step's policy would be."""
        self.draw += 1
        self._fused_synced = False
        return self.venv.sample_actions(self.act_buf, 1234 + self.rank, self.draw)

    def on_stream(self):
        return self.torch.cuda.stream(self.shard.stream) if self.shard.pipelined else contextlib.nullcontext()

    fused_policy = True     # the uniform random policy inside the step (tg_step_random: drawn by the step kernel itself, or the step's first launch); False: its own launch before every step
    _fused_synced = False   # the context's device-side draw counter equals self.draw

    def step(self, env):
        """One rollout step: action_space.sample() for the whole batch + VecEnv.step.  The same draws either way (draw k of seed 1234 + rank)."""
        if self.fused_policy and hasattr(env, "step_random"):
            out = env.step_random(1234 + self.rank, self.draw, restart=not self._fused_synced)
            self.draw += 1
            self._fused_synced = True
            return out
        return env.step(self.actions())

    def timed(self, env, steps, barrier, flush=None):
        # The collector is held off the timed region, as timeit does (a 20-step window is 0.86 ms on the headline; a young-generation pass of a
        # process with torch imported can take 0.1 - 0.2 ms).  No gc.collect() HERE: a full collection is ~0.1 s of host time during which the
        # device idles, and the window then starts on a part that has clocked down - measured, round 6: driver-like runs 45.4 instead of 43.1 us
        # per step, one_step_window_ms 0.090 instead of 0.058.  The collection runs once, before the spin-up (run_window).
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._timed(env, steps, barrier, flush)
        finally:
            if gc_was_on:
                gc.enable()

    def _timed(self, env, steps, barrier, flush=None):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(env)
        self.flushed, self.error = None, None
        if flush is not None:
            try:
                self.flushed = flush()                             # rank 0: the last step's gathered batch
            except RuntimeError as e:                              # the ipc transport reports flag timeouts here; the barrier below must still be met
                self.error = str(e)
        barrier()
        return time.perf_counter() - t0

    def profile(self, env, steps, barrier, clock=True):
        """Per-kernel durations, outside the timed region.  First the rollout as it is timed - the same launches, the same in-step policy - with the
        kernels stamping their own clock (tg_profile_enable(2)); then a few steps launch by launch with HIP event pairs (tg_profile_enable(1)): the
        figures earlier rounds quoted, kept beside the clock's for comparison together with what an empty event pair measures."""
        prof = {}
        if clock:       # (not under a process group: with TG_STEP_GRAPH=1 switching this mode on and off re-captures the step graphs, see TorchShard.prime)
            self.venv.profile("clock")
            self._fused_synced = False
            self.step(env)                               # (TG_STEP_GRAPH=1: the re-capture of the graphs happens here, outside the window below)
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(env)
            barrier()
            window = time.perf_counter() - t0
            prof = self.venv.profile_get()
            prof["clock_window_ms_per_step"] = 1e3 * window / max(steps, 1)
            steps += 1
        self.venv.profile(True)
        for _ in range(min(steps, 10)):
            env.step(self.actions())
        barrier()
        ev = self.venv.profile_get()
        self.venv.profile(False)
        self._fused_synced = False
        for k in ("step", "render", "reset", "render_masked", "scene", "empty_event_pair"):
            prof[k] = ev[k]
        # scopes per env step and class: from the clocked rollout where the class carries a clock, from the event run otherwise
        n_ev = max(min(steps, 10), 1)
        for k in ("step", "render", "reset", "render_masked"):
            c = prof.get(k + "_clock", (0.0, 0))[1] if clock else 0
            prof[k + "_per_step"] = c / max(steps, 1) if c > 0 else ev[k][1] / n_ev
        return prof

    @staticmethod
    def per_launch(prof, key):
        ms, k = prof.get(key, (0.0, 0))
        return ms / max(k, 1)

    def kernel_times(self, prof):
        """ms per launch of the step's kernel classes: by the kernels' OWN clock where the kernel carries one (csrc/tg_kt.hpp: first wavefront
        start -> last wavefront end, wall_clock64 - what rocprofv3 --kernel-trace reports per dispatch), HIP events otherwise (they carry the
        empty event pair's 3 - 5 us on top of the kernel: VERDICT r4 found event figures that did not fit inside the step they add up to)."""
        out, src = {}, {}
        for name, key in (("k_step", "step"), ("k_render_tactile", "render"), ("k_reset_per_launch", "reset"), ("k_render_tactile_masked", "render_masked")):
            if prof.get(key + "_clock", (0.0, 0))[1] > 0:
                out[name], src[name] = self.per_launch(prof, key + "_clock"), "kernel clock"
            else:
                # an event pair around a launch carries what an EMPTY pair measures (3 - 5 us) on top of the kernel: taken off, so that the
                # kernels of a step add up to no more than the step (config 4's event figures summed to 1.9603 ms of a 1.9586 ms step)
                empty = self.per_launch(prof, "empty_event_pair")
                raw = self.per_launch(prof, key)
                out[name], src[name] = (max(raw - empty, 0.0), "hip events minus the empty event pair") if raw > 0 else (0.0, "hip events")
        return out, src

    def dominant(self, prof):
        """(kernel name, ms per launch, key) of the kernel with the largest summed duration; the render kernel named as launched
        (csrc/tg_raster.hip launch_render: the edge and the cube take the block kernel, the pole's plate the two-pass small-mesh kernel)."""
        km, _ = self.kernel_times(prof)
        big = self.image_size % 128 == 0
        render_name = ("k_render_blocks<16>" if self.env_id in ("edge_follow-v0", "object_push-v0") and big else
                       "k_render_small<128,64,2>" if self.env_id == "object_balance-v0" and big else
                       "k_render_scatter" if self.env_id == "object_roll-v0" else "k_render_tactile")
        if prof["step"][1] == 0 and prof["render"][1] > 0:            # fused_step: the one launch (csrc/tg_fused.hip)
            return "k_step_render", km["k_render_tactile"], "k_render_tactile"
        # The render is ALWAYS the kernel `frac` is quoted for: it is the kernel that moves the algorithmic bytes (the image), whatever its share
        # of the step - on the headline k_step and the render are a coin flip from run to run (16.6 against 16.1 us; until round 5 the line changed
        # its kernel with the box), on config 4 k_step is 96 % of the step and moves 1 % of the bytes.  Both are reported with their own durations
        # and traffic under roofline.kernels; roofline.step_frac is the whole step.
        return render_name, km["k_render_tactile"], "k_render_tactile"

    def roofline(self, prof, traffic_ok=True, ms_per_step=None):
        name, dom_ms, key = self.dominant(prof)
        km, src = self.kernel_times(prof)
        ab = algo_bytes(self.env_id, self.image_size)
        achieved = ab * self.n / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = read_traffic(self.env_id, self.n, self.image_size, key, ab) if traffic_ok else None
        lps = {q: prof.get(q + "_per_step", 1.0 if prof[q][1] else 0.0) for q in ("step", "render", "reset", "render_masked")}
        launches = {"k_step": prof["step"][1], "k_render_tactile": prof["render"][1], "k_reset": prof["reset"][1]}
        kernels_per_step = sum(km[k] * lps[q] for k, q in (("k_step", "step"), ("k_render_tactile", "render"), ("k_reset_per_launch", "reset"),
                                                           ("k_render_tactile_masked", "render_masked")))
        out = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "algorithmic_bytes_per_env_step": ab,
               "kernel_ms": {k: round(v, 4) for k, v in km.items()}, "kernel_ms_source": src,
               "kernel_ms_sum_per_step": round(kernels_per_step, 4),
               "profiled_window_ms_per_step": round(prof["clock_window_ms_per_step"], 4) if "clock_window_ms_per_step" in prof else None,
               "kernel_ms_hip_events": {"k_step": round(self.per_launch(prof, "step"), 4), "k_render_tactile": round(self.per_launch(prof, "render"), 4),
                                        "k_reset_per_launch": round(self.per_launch(prof, "reset"), 4),
                                        "empty_event_pair": round(self.per_launch(prof, "empty_event_pair"), 4)},
               "launches": launches}
        share = {k: km[k] * lps[q] for k, q in (("k_step", "step"), ("k_render_tactile", "render"))}
        out["kernels"] = {
            "k_step": {"ms": round(km["k_step"], 4), "traffic": read_traffic(self.env_id, self.n, self.image_size, "k_step", ab) if traffic_ok else None,
                       "share_of_kernel_time": round(share["k_step"] / kernels_per_step, 3) if kernels_per_step > 0 else None},
            name: {"ms": round(km["k_render_tactile"], 4), "traffic": traffic, "frac": out["frac"],
                   "share_of_kernel_time": round(share["k_render_tactile"] / kernels_per_step, 3) if kernels_per_step > 0 else None}}
        if ms_per_step:
            # the WHOLE step against the roofline: compulsory bytes of every env step of the batch / the step's wall time (not one kernel's)
            step_gbs = ab * self.n / (ms_per_step * 1e-3) / 1e9
            out["step_achieved"] = round(step_gbs, 3)
            out["step_frac"] = round(step_gbs / HBM_PEAK_GBS, 6)
        return out

    def close(self):
        self.venv.close()


def read_traffic(env_id, n, image_size, key, ab):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes in profiles/ (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE; counters cannot be read from inside this process).  Quoted only when the file was measured on a library built from the
    sources in this tree (`source_sha16`) and on this configuration; otherwise null with the reason."""
    try:
        tr = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
    except (OSError, ValueError):
        return None
    have = source_hash()
    if tr.get("source_sha16") != have:
        return {"bytes_per_launch": None, "stale": f"{TRAFFIC_FILE} was measured on sources {tr.get('source_sha16')}, this tree is {have}"}
    for wl in tr.get("workloads", []):
        if (wl["env"], wl["num_envs"], wl["image_size"], wl["physics"]) == (env_id, n, image_size, "f64") and key in wl:
            k = wl[key]
            b = (k["fetch_corrected_kb"] + k["write_kb"]) * 1024.0
            return {"bytes_per_launch": round(b), "kernel": k["kernel"], "source_sha16": have,
                    "source": f"{TRAFFIC_FILE} (rocprofv3 PMC: FETCH_SIZE x2 + WRITE_SIZE)", "vs_algorithmic": round(b / (ab * n), 3)}
    return None


def guard_report(w, env, steps, barrier):
    """The broadphase guard over `steps` further steps of the rollout with the check as a node of every step (tg_set_broadphase; survey row n2):
    how many env states were checked, how many unexpected pairs Bullet's broadphase would have handed to its narrowphase (world AABBs overlap) and
    how many of them survive the oriented-box / hull tests (hits: 0 = no contact other than the ones the solver has rows for was possible), and
    what the check costs the step."""
    import numpy as np
    v = w.venv
    default_on = v._guard is not None and bool(v._guard.struct.every_step)
    v.set_broadphase_guard(True)                       # (also zeroes the totals)
    w._fused_synced = False
    for _ in range(5):
        w.step(env)
    dt = w.timed(env, steps, barrier)
    tot = v.broadphase_totals()
    st = v.get_state()
    names = v._guard.describe(int(np.bitwise_or.reduce(st["broadphase_mask"]))) if tot["hits"] else []
    if not default_on:
        v.set_broadphase_guard(False)
        w._fused_synced = False
    return {"env_checks": tot["env_checks"], "stage1_pairs": tot["pairs"], "hits": tot["hits"], "slots_in_hits_of_the_last_step": names,
            "ms_per_step_with_the_check": round(1e3 * dt / steps, 4), "default": "a node of every step" if default_on else "on demand (check_broadphase)"}


def companion(env_id, image_size, n, physics, steps, barrier, what, **kw):
    """A short run of another configuration on this GPU (after the headline's timed region): value, ms per step and its dominant
    kernel's roofline fraction."""
    w = Workload(env_id, n, image_size, physics, 0, 0, **kw)
    # untimed steps before the window: enough for the reset bank's first refill of ALL envs (it runs beside the steps that follow a reset of the whole
    # batch, on the bank's stream, and takes CUs from them: a 20-step window opened 10 steps after the reset - the driver's command - measured
    # surface_follow-v0 at 113 us per step against 99 in steady state)
    warm = 60
    with w.on_stream():
        w.shard.reset()
        for _ in range(warm):
            w.step(w.shard)
        dt = w.timed(w.shard, steps, barrier)
        w.shard.reset()                                    # the profile window covers the same phase of the episodes as the timed window
        for _ in range(warm):
            w.step(w.shard)
        prof = w.profile(w.shard, steps, barrier)
    roof = w.roofline(prof, ms_per_step=1e3 * dt / steps)
    out = {"workload": f"{env_id}, {w.modes['arm_type'].upper()} + {w.modes['tactile_sensor_name']}, {n} vec-envs, {image_size}x{image_size}" + what,
           "value": round(n * steps / dt, 1), "unit": "env-steps/s", "steps": steps, "warmup": warm, "ms_per_step": round(1e3 * dt / steps, 4),
           "roofline": {k: roof[k] for k in ("kernel", "achieved", "frac", "step_frac", "traffic", "algorithmic_bytes_per_env_step", "kernel_ms",
                                             "kernel_ms_source", "kernel_ms_sum_per_step", "profiled_window_ms_per_step", "kernels")}}
    if not w.residual_threshold and n <= 1024:
        with w.on_stream():
            out["broadphase_guard"] = guard_report(w, w.shard, steps, barrier)
    if w.residual_threshold:
        sw = w.venv.get_state()["solver_sweeps"]
        out["solver_residual_threshold"] = w.residual_threshold
        out["mean_sweeps_per_tick"] = round(float(sw.mean()) / w.ticks_per_step, 3)
        out["max_sweeps_per_tick"] = round(float(sw.max()) / w.ticks_per_step, 3)
    w.close()
    return out


def threshold_companions(physics, steps, barrier, extra):
    """PARITY_ASSUMPTIONS A7b: the reference never sets solverResidualThreshold (base_tactile_env.py:127-130), so PyBullet's own default
    applies - believed to be 1e-7, under which the Gauss-Seidel loop of every tick leaves after a few sweeps instead of at convergence.  The
    headline keeps threshold 0 (the library default of Bullet itself, and the reading every earlier round measured) until someone runs the
    PyBullet kit; this is the other answer on record: every BASELINE config at 1024 envs with tg_config.solver_residual_threshold = 1e-7 -
    value, ms per step and the mean number of sweeps a tick ran (tg_state_view.solver_sweeps of the last timed step / ticks per step)."""
    out = {}
    for name, env_id, size, k, kw in (("config2_edge_follow", "edge_follow-v0", 128, steps, {}), ("config3_surface_follow", "surface_follow-v0", 128, steps, {}),
                                      ("config4_object_push", "object_push-v0", 128, max(20, steps // 2), extra),
                                      ("config5_object_balance", "object_balance-v0", 256, steps, {})):
        out[name] = companion(env_id, size, 1024, physics, k, barrier, ", solver_residual_threshold 1e-7", residual_threshold=1e-7, **kw)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed steps (SURVEY 8d: >= 2000)")
    ap.add_argument("--warmup", type=int, default=100, help="untimed steps (SURVEY 8d: 100)")
    ap.add_argument("--num-envs", type=int, default=1024, help="envs per GPU (BASELINE configs[1]: 1024)")
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--physics", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-literal", action="store_true", help="skip the literal-solver companion run (keeps a rocprofv3 trace of this command to one solver mode)")
    ap.add_argument("--no-companions", action="store_true", help="skip other_configs / roofline_16384 (same purpose)")
    ap.add_argument("--separate-policy", action="store_true", help="launch the uniform random policy (tg_sample_actions) as its own kernel before every step instead of "
                    "inside the step (tg_step_random)")
    ap.add_argument("--sync-steps", action="store_true", help="block the host on every step (VecEnv.step_wait semantics) instead of pipelining")
    ap.add_argument("--full-sweeps", action="store_true",
                    help="always run all 150 PGS sweeps per tick instead of leaving the loop at convergence to the last bit (DESIGN.md 4.1)")
    ap.add_argument("--env", default="edge_follow-v0", choices=sorted(ENV_MODES),
                    help="headline = edge_follow-v0 (BASELINE configs[1]); surface_follow-v0 = configs[2]; object_push-v0 = configs[3]; "
                         "object_balance-v0 = configs[4] (use --image-size 256)")
    ap.add_argument("--contact-mapping", default="auto", choices=["auto", "lane", "wave"],
                    help="object_push / object_roll: one wavefront per env or one lane per env for the contact solve (tg_config.contact_mapping)")
    ap.add_argument("--observation-mode", default=None, help="override the config's observation_mode (e.g. visuotactile: adds the RGB scene camera, SURVEY 8 row f4); "
                    "measurement aid, the bench line stays the tactile configuration")
    ap.add_argument("--narrowphase", default="closed_form", choices=["closed_form", "gjk_manifold", "gjk_single"],
                    help="object_push: the tip - cube narrowphase (tg_config.narrowphase); measurement aid, the bench line stays the default closed form")
    ap.add_argument("--solver-iters", type=int, default=None, help="object_push / object_roll: override numSolverIterations (150); measurement aid, not a bench configuration")
    ap.add_argument("--payload", default=os.environ.get("TG_BENCH_PAYLOAD", "auto"), choices=["auto", "full", "interior", "tiles"],
                    help="N > 1: what the tactile part of the per-step message to rank 0 carries (parallel.py)")
    ap.add_argument("--transport", default=os.environ.get("TG_BENCH_TRANSPORT", "auto"), choices=["auto", "collective", "ipc"],
                    help="N > 1: peers store straight into rank 0's IPC-mapped receive slots (ipc), or torch.distributed collectives over RCCL; "
                         "auto (default) = ipc when its set-up handshake succeeds on every rank, else collective")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: time only the exchange-free rollout (per-rank learners)")
    ap.add_argument("--pre-warm-ms", type=float, default=float(os.environ.get("TG_BENCH_PRE_WARM_MS", "400")),
                    help="untimed device spin-up BEFORE the reset + W warm-up steps + K timed steps: the same rollout stepped for this many milliseconds, so "
                         "that a short window (the driver's W = 5, K = 20 is 1 ms of device time) does not sit on the clock ramp of a part that has just "
                         "been idle; reported as pre_warm_ms / pre_warm_steps; 0 switches it off")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("TG_BENCH_SPAWN") == "1"):   # TG_BENCH_SPAWN: the same path with one rank (1-GPU test)
        spawn_ranks(args.gpus)                      # re-executes this script under torch.distributed.run; never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or without a launcher: bench.py starts its own ranks)")

    # stdout carries the ONE JSON line and nothing else: whatever native libraries write to file descriptor 1 (RCCL prints a five-line
    # version banner there when its communicator comes up, flushed at exit, i.e. after the JSON) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from tactile_gym_amd.parallel import ShardedVecEnv
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the env step has no CPU fallback (the CPU oracle is only the reported baseline)")
    torch.cuda.set_device(local_rank)
    if args.separate_policy:
        Workload.fused_policy = False
    n = args.num_envs
    extra = dict(contact_mapping=args.contact_mapping, solver_iterations=args.solver_iters)
    if args.env == "object_push-v0" and args.narrowphase != "closed_form":
        extra["narrowphase"] = args.narrowphase
    w = Workload(args.env, n, args.image_size, args.physics, rank, local_rank, full_sweeps=args.full_sweeps,
                 observation_mode=args.observation_mode, pipelined=not args.sync_steps, **extra)
    venv, shard, modes, max_steps = w.venv, w.shard, w.modes, w.max_steps
    if world > 1 or os.environ.get("TG_BENCH_FORCE_COLLECTIVE") == "1":
        # TG_STEP_GRAPH=1: every graph the rollout will launch is captured HERE, before the process group (and its watchdog thread) exists: no step below captures
        # (default since round 6: the steps are enqueued launch by launch and nothing is ever captured)
        # anything (parallel.TorchShard.prime; rounds 3-4 slept three watchdog periods before the first capture instead)
        with w.on_stream():
            shard.prime(w.act_buf)
    dist, rccl_ranks = None, 1
    force = world == 1 and os.environ.get("TG_BENCH_FORCE_COLLECTIVE") == "1"   # 1-GPU check of the RCCL gather path (one rank)
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        ones = torch.ones(1, device=f"cuda:{local_rank}", dtype=torch.int32)
        dist.all_reduce(ones)                       # every rank is on the communicator: the sum of ones is the number of RCCL ranks
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)

    gathered = dist is not None and not args.no_gather

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if dist is None:
            return x
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(transport, payload, warmup, steps=None):
        """reset + warm-up + the K timed steps (or `steps`: a probe) through one exchange configuration; every rank learns whether any rank saw an error, and
        whether what rank 0 was handed is what the ranks rendered (byte sums of every rank's last tactile batch, outside the timed region)."""
        e = ShardedVecEnv(shard, dist, overlap=True, force_collective=force, payload=payload, transport=transport) if gathered else shard
        with w.on_stream():
            gc.collect()                                   # (before the spin-up, never between it and the timed region: Workload.timed)
            if args.pre_warm_ms > 0 and steps is None:
                # device spin-up, untimed and before the window's own reset: the timed region below is reset -> W warm-up steps -> K steps as always
                e.reset()
                t_pw, k_pw = time.perf_counter(), 0
                while time.perf_counter() - t_pw < 1e-3 * args.pre_warm_ms:
                    for _ in range(50):
                        w.step(e)
                    k_pw += 50
                    torch.cuda.synchronize()
                pre_warm["steps"], pre_warm["ms"] = k_pw, round(1e3 * (time.perf_counter() - t_pw), 1)
            e.reset()
            for _ in range(warmup):
                w.step(e)
            t = allmax(w.timed(e, steps or args.steps, barrier, flush=e.flush if gathered else None))
            ok, why = True, None
            if gathered:
                if os.environ.get("TG_BENCH_INJECT_EXCHANGE_FAULT") and getattr(e, "transport", None) == "ipc":
                    w.error = w.error or "injected fault (TG_BENCH_INJECT_EXCHANGE_FAULT: exercises the fallback below, tests only)"
                bad = allmax(1.0 if w.error else 0.0)
                if bad:
                    ok, why = False, w.error or "another rank's exchange timed out"
                else:
                    cs = venv.tactile_torch().sum(dtype=torch.int64).reshape(1)
                    sums = [torch.zeros_like(cs) for _ in range(world)]
                    dist.all_gather(sums, cs)
                    same = 1.0
                    if rank == 0 and w.flushed is not None:
                        got = w.flushed[0]["tactile"].reshape(world, -1).sum(dim=1, dtype=torch.int64)
                        same = 1.0 if bool((got == torch.cat(sums)).all().item()) else 0.0
                    if allmax(1.0 - same):
                        ok, why = False, "the batch rank 0 was handed differs from what the ranks rendered"
        return e, t, ok, why

    pre_warm = {"steps": 0, "ms": 0.0}
    env, dt, verified, fallback, probe = None, None, None, None, None
    transport, payload = args.transport, args.payload
    if gathered and transport == "auto" and payload == "auto":
        # Nothing here has run across GPUs before the driver's run, so the choice between the two transports is made by measurement: a
        # short probe of each (same envs, same steps), the timed K steps then go through the faster one that delivered a verified batch.
        probe_steps = max(20, min(args.steps, 100))
        probe, best = {"steps": probe_steps}, None
        for tr in ("auto", "collective"):
            if tr == "collective" and probe.get("ipc_unavailable"):
                break
            e, t, ok, why = measure(tr, "auto", min(args.warmup, 10), probe_steps)
            name = f"{e.transport} + {e.payload}"
            if tr == "auto" and e.transport != "ipc":
                probe["ipc_unavailable"] = True                 # the set-up handshake failed on some rank: there is only the gather
            probe[name] = {"ms_per_step": round(1e3 * t / probe_steps, 4), "verified": ok, **({"why": why} if why else {})}
            if ok and (best is None or t < best[0]):
                best = (t, e.transport, e.payload)
            if not ok and e.transport == "ipc":
                fallback = {"from": name, "why": why}           # the ipc transport did not deliver: the number comes from the RCCL gather
            try:
                e.close()
            except Exception:  # noqa: BLE001
                pass
        if best is not None:
            transport, payload = best[1], best[2]
            probe["chosen"] = f"{transport} + {payload}"
    env, dt, ok, why = measure(transport, payload, args.warmup)
    if gathered:
        verified = ok
        if not ok and env.transport == "ipc":
            # the ipc transport did not deliver (flag timeouts or a wrong batch): the contract value is measured again through the RCCL gather
            fallback = {"from": f"ipc + {env.payload}", "why": why}
            try:
                env.close()
            except Exception:  # noqa: BLE001
                pass
            env, dt, verified, why = measure("collective", "auto" if args.payload == "tiles" else args.payload, min(args.warmup, 10))
    with w.on_stream():
        no_gather = None
        if gathered and world > 1:               # the same K steps without the exchange: what per-rank learners would see
            dt_ng = allmax(w.timed(shard, args.steps, barrier))
            no_gather = {"value": round(n * world * args.steps / dt_ng, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * dt_ng / args.steps, 4),
                         "what": "the same K steps on every rank without the per-step exchange to rank 0 (observations consumed where they are produced)"}
        # What a short timed window carries besides its steps: the launch latency of its first step and the completion latency of the final
        # synchronisation (pipeline fill + drain) are paid once per window - at the driver's K = 20 they are ~2.5 us of every step, at K = 2000
        # nothing.  Measured here, outside the timed region: single-step windows, and (for short runs) a 1000-step window as the steady state.
        window = None
        if dist is None:
            ones = sorted(w.timed(env, 1, barrier) for _ in range(7))
            one_ms = 1e3 * ones[len(ones) // 2]
            long_k = 1000
            long_ms = 1e3 * w.timed(env, long_k, barrier) / long_k if args.steps < long_k else 1e3 * dt / args.steps
            window = {"one_step_window_ms": round(one_ms, 4), "steady_state_ms_per_step": round(long_ms, 4),
                      "steady_state_value": round(n * 1e3 / long_ms, 1), "steady_state_steps": long_k if args.steps < long_k else args.steps,
                      "fill_and_drain_ms_per_window": round(one_ms - long_ms, 4),
                      "explained_ms_per_step": round(long_ms + (one_ms - long_ms) / args.steps, 4),
                      "what": "ms_per_step of a K-step window = steady state + (fill + drain) / K: a window starts on an idle queue (launch latency of its first "
                              "step) and ends in a host synchronisation (completion latency); one_step_window_ms is a window of K = 1 (median of 7)"}
        guard = guard_report(w, env, max(20, min(args.steps, 300)), barrier) if dist is None else None
        prof = w.profile(env, min(args.steps, 50), barrier, clock=dist is None)
        # SURVEY 8d asks for the rate with and without episode resets: a window that starts right after a reset of every env and
        # ends before any env can reach max_steps (an env that meets its goal early is still reset, as in any rollout)
        no_reset = None
        if dist is None:
            env.reset()
            for _ in range(10):
                w.step(env)
            k_nr = max(10, min(args.steps, max_steps - 20))
            dt_nr = w.timed(env, k_nr, barrier)
            no_reset = {"value": round(n * k_nr / dt_nr, 1), "unit": "env-steps/s", "steps": k_nr, "ms_per_step": round(1e3 * dt_nr / k_nr, 4),
                        "what": "window between full-batch resets (no env reaches max_steps inside it)"}
        separate = None
        if dist is None and Workload.fused_policy and hasattr(env, "step_random"):
            # the same K steps with the policy's draw as its own launch in front of every step (what a torch policy would be: separate kernels)
            Workload.fused_policy = False
            for _ in range(10):
                w.step(env)
            dt_sp = w.timed(env, args.steps, barrier)
            Workload.fused_policy = True
            separate = {"value": round(n * args.steps / dt_sp, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * dt_sp / args.steps, 4),
                        "what": "tg_sample_actions as its own kernel launch before every tg_step instead of a draw inside the step"}
    exchange = env.exchange_info() if gathered and hasattr(env, "exchange_info") else None
    if exchange is not None:
        exchange["verified"] = verified
        if fallback is not None:
            exchange["fallback"] = fallback
        if probe is not None:
            exchange["probe"] = probe

    solo = world == 1 and not force
    literal = None
    if solo and not args.full_sweeps and not args.no_literal:
        # the same workload with the literal solver (every tick: dynamics + all 150 PGS sweeps), for comparison; short run
        lw = Workload(args.env, n, args.image_size, args.physics, rank, local_rank, full_sweeps=True, observation_mode=args.observation_mode,
                      pipelined=False, **extra)
        lw.shard.reset()
        for _ in range(5):
            lw.step(lw.shard)
        lsteps = max(10, min(args.steps, 40))
        ldt = lw.timed(lw.shard, lsteps, barrier)
        lprof = lw.profile(lw.shard, 10, barrier)
        literal = {"value": round(n * lsteps / ldt, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * ldt / lsteps, 4), "steps": lsteps,
                   "k_step_ms": round(lw.kernel_times(lprof)[0]["k_step"], 4),
                   "k_render_ms": round(lw.kernel_times(lprof)[0]["k_render_tactile"], 4),
                   "what": "pgs_full_sweeps=1: dynamics + exactly 150 Gauss-Seidel sweeps in every one of the 24 ticks (no convergence exit, no analytic fixed point)"}
        lw.close()
    others, big, staggered, thr_runs = None, None, None, None
    headline = args.env == "edge_follow-v0" and n == 1024 and args.image_size == 128 and not args.observation_mode and not args.full_sweeps
    if solo and headline and not args.no_companions:
        # one driver-visible line per remaining BASELINE config, 1024 envs on this GPU (the configs' own sharding puts 1024 on each GPU)
        k = max(20, min(args.steps, 200))
        others = [companion("surface_follow-v0", 128, 1024, args.physics, k, barrier, " (BASELINE configs[2])"),
                  companion("object_push-v0", 128, 1024, args.physics, max(20, min(args.steps, 100)), barrier,
                            " (BASELINE configs[3]: 4096 envs over 4 GPUs = 1024 per GPU)", **extra),
                  companion("object_balance-v0", 256, 1024, args.physics, k, barrier, " (BASELINE configs[4]: 8192 envs over 8 GPUs = 1024 per GPU)")]
        thr_runs = threshold_companions(args.physics, k, barrier, extra)
        b = companion("edge_follow-v0", 128, 16384, args.physics, k, barrier, " (the headline workload at 16 384 envs: the chip filled)")
        big = {"num_envs": 16384, "value": b["value"], "ms_per_step": b["ms_per_step"], **b["roofline"]}
        # the headline workload with its episodes out of phase (an RL run's condition: some env finishes in nearly every step); the timed
        # region above starts all envs together, so they all finish in the same step, once per 200
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from desync_rate import run as desync_run
        sdt, sinfo = desync_run(True, 1024, max(200, min(args.steps, 1000)), MAX_STEPS.get("edge_follow-v0", 200))
        # (the same for configs 3 and 4, inside their other_configs entries; config 5's episodes end by the pole falling - out of phase in its line already)
        for entry, env_id, k_st in ((others[0], "surface_follow-v0", max(200, min(args.steps, 1000))), (others[1], "object_push-v0", max(100, min(args.steps, 300)))):
            odt, oinfo = desync_run(True, 1024, k_st, MAX_STEPS.get(env_id, 200), env_id)
            entry["staggered_episodes"] = {"value": round(1024 / odt, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * odt, 4), "steps": k_st, **oinfo}
        staggered = {"value": round(1024 / sdt, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * sdt, 4), **sinfo,
                     "what": "the headline workload with every env's episode phase drawn uniformly (masked resets during the first 200 steps): ~5 of 1024 envs "
                             "finish in every step; finished envs take their precomputed reset from the reset bank (tg_config.reset_bank, on by default) "
                             "and keep their solver licence (tools/desync_rate.py, profiles/r6_final_desync.txt)"}

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / dt
        roof = w.roofline(prof, traffic_ok=not args.full_sweeps and args.physics == "f64", ms_per_step=1e3 * dt / args.steps)
        roof["note"] = ("HBM roofline in form only: the step is bound by per-workgroup latency chains (render) and by the serial solver "
                        "(k_step), not by bytes; frac = the dominant kernel alone, step_frac = the whole step (its dependent launches and the gaps "
                        "between them); kernel_ms by the kernels' own clock, their sum per step fits inside ms_per_step; see DESIGN.md 4.3")
        if big is not None:
            roof["at_16384_envs"] = big
        par = f"env-shard x{world}"
        if exchange is not None:
            par += " + " + exchange["what"]
        out = {
            "metric": "env-steps/sec (128x128 tactile obs) at N envs", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "pre_warm_ms": pre_warm["ms"], "pre_warm_steps": pre_warm["steps"], "window": window,
            "dtype": "f64" if args.physics == "f64" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.env}, {modes['arm_type'].upper()} + {modes['tactile_sensor_name']}, {n} vec-envs per MI355X, {args.image_size}x{args.image_size} tactile obs, "
                                   f"random actions, TCP_velocity_control, {12 if args.env == 'object_balance-v0' else 24} sim ticks per step (PGS budget 150 sweeps per tick), auto-reset on",
                       "envs_per_gpu": n, "total_envs": total_envs, "parallelism": par},
            "rccl_ranks": rccl_ranks if dist is not None else None,
            "exchange": exchange,
            "no_gather": no_gather,
            "roofline": roof,
            "solver": "literal: dynamics + 150 PGS sweeps every tick (pgs_full_sweeps)" if args.full_sweeps else
                      "default: PGS leaves at last-bit convergence; ticks whose motor solve is provably unclamped and demonstrably converged take "
                      "the solver's analytic fixed point (qd = target); joints within 1e-11 rad of the literal solver over 1024 envs x 260 steps incl. auto-resets "
                      "(tests/test_gpu_parity.py::test_default_solver_equals_literal_solver_at_config_scale; DESIGN.md 4.1)",
            "policy": ("uniform random actions (action_space.sample() for the whole batch), drawn on the device inside the step (tg_step_random: by the step kernel "
                       "itself for edge_follow / surface_follow under velocity control, as the step's first launch elsewhere; draw k identical to tg_sample_actions(seed, k))") if Workload.fused_policy and hasattr(env, "step_random") else
                      "uniform random actions drawn on the device by tg_sample_actions, one launch before every step",
            "separate_policy_launch": separate,
            "literal_solver": literal,
            "without_full_batch_reset": no_reset,
            "resets_in_timed_region": bool((args.warmup % max_steps) + args.steps >= max_steps),
            "broadphase_guard": guard,
            "other_configs": others,
            "residual_threshold_1e-7": thr_runs,
            "staggered_episodes": staggered,
        }
        if args.no_gather and world > 1:
            out["config"]["parallelism"] = f"env-shard x{world}, no exchange (--no-gather)"
        if prof["scene"][1]:    # --observation-mode visual / visuotactile: the scene camera's two kernels (eye<-frame transforms + raster), per draw
            out["scene_camera"] = {"observation_mode": modes["observation_mode"], "ms_per_draw": round(prof["scene"][0] / prof["scene"][1], 4),
                                   "draws": prof["scene"][1]}
        if solo and not args.no_cpu_baseline and args.env == "edge_follow-v0":
            out["cpu_baseline"] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if gathered and hasattr(env, "close"):
        env.close()
    w.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
