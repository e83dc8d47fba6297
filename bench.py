#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of edge_follow-v0 (UR5 + TacTip, 128x128 tactile obs), BASELINE.json configs[1].

    python bench.py --gpus 1 --steps 2000 --warmup 100      (the defaults: SURVEY 8d asks for 100 warm-up and >= 2000 timed steps)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one VecEnv.step() of the whole batch: controller + 24 sim ticks + tactile render for every env, random
actions ~ U[-0.25, 0.25) drawn on the device by tg_sample_actions (synthetic), auto-reset on (episodes of 200 steps, so resets fall
inside the timed region whenever K + W crosses a multiple of 200; reported separately via `resets_in_timed_region`).
Observations stay resident in HBM (device tensors); the PCIe-inclusive rate is quoted in DESIGN.md, never here.
The rollout is device resident: steps are enqueued on a torch stream (TorchShard(pipelined=True)) and their outputs consumed on it,
the host does not wait per step (--sync-steps restores the blocking VecEnv.step_wait path; both rates are in profiles/).
Extra fields: roofline (dominant kernel, HIP-event durations, PMC traffic), cpu_baseline (the CPU oracle on all host cores),
literal_solver (the same workload with exactly 150 PGS sweeps in every tick), without_full_batch_reset.
Weak scaling: every rank owns --num-envs envs; rank 0 receives all observations / rewards / dones by one packed RCCL gather
per step, started asynchronously so that it overlaps the next step's simulation; the last gather is waited for inside
the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
ALGO_BYTES_PER_ENV_STEP = 16600.0   # BASELINE.md section 3 / SURVEY 8(d): 16 384 B image + ~0.2 KB state/action/reward
ALGO_BYTES_SURFACE = 33000.0        # config 3: + the per-env 64x64 f32 heightfield read
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def _cpu_worker(args):
    seed, seconds = args
    import numpy as np
    from oracle.ref_env import OracleEdgeFollowEnv
    env = OracleEdgeFollowEnv(seed=seed, max_steps=200, image_size=(128, 128), env_modes=MODES)
    env.reset()
    rng = np.random.default_rng(seed)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, _, done, _ = env.step(rng.uniform(-0.25, 0.25, 2))
        n += 1
        if done:
            env.reset()
    return n, time.perf_counter() - t0


def cpu_baseline(seconds=12.0):
    """The CPU oracle (oracle/: a port, not PyBullet — PyBullet is not installable here) on this box's host cores."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    n1, t1 = _cpu_worker((1, min(4.0, seconds / 3)))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(1 + i, seconds) for i in range(cores)])
    total = sum(r[0] for r in res) / max(r[1] for r in res)
    try:   # time the real reference if the box happens to have it (it does not travel with this repo)
        import pybullet  # noqa: F401
        pyb = "importable on this box but not timed: the reference's env classes do not travel with this repo"
    except Exception:  # noqa: BLE001
        pyb = "unavailable (import pybullet fails on this box)"
    return {"value": round(total, 1), "unit": "env-steps/s", "cores": cores, "kind": "port", "pybullet_reference": pyb,
            "sample": f"{cores} processes x 1 oracle env (edge_follow-v0, 128x128, random actions, resets included) for {seconds:.0f} s; "
                      f"single process: {n1 / t1:.1f} env-steps/s"}


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
    ap.add_argument("--sync-steps", action="store_true", help="block the host on every step (VecEnv.step_wait semantics) instead of pipelining")
    ap.add_argument("--full-sweeps", action="store_true",
                    help="always run all 150 PGS sweeps per tick instead of leaving the loop at convergence to the last bit (DESIGN.md 4.1)")
    ap.add_argument("--env", default="edge_follow-v0", choices=["edge_follow-v0", "surface_follow-v0", "object_balance-v0", "object_push-v0", "object_roll-v0", "surface_follow-v2"],
                    help="headline = edge_follow-v0 (BASELINE configs[1]); surface_follow-v0 = configs[2]; object_balance-v0 = configs[4] "
                         "(use --image-size 256)")
    ap.add_argument("--contact-mapping", default="auto", choices=["auto", "lane", "wave"],
                    help="object_push / object_roll: one wavefront per env or one lane per env for the contact solve (tg_config.contact_mapping)")
    ap.add_argument("--observation-mode", default=None, help="override the config's observation_mode (e.g. visuotactile: adds the RGB scene camera, SURVEY 8 row f4); "
                    "measurement aid, the bench line stays the tactile configuration")
    ap.add_argument("--solver-iters", type=int, default=None, help="object_push / object_roll: override numSolverIterations (150); measurement aid, not a bench configuration")
    args = ap.parse_args()

    import torch
    import tactile_gym_amd as tg
    from tactile_gym_amd.parallel import ShardedVecEnv, TorchShard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the env step has no CPU fallback (the CPU oracle is only the reported baseline)")
    torch.cuda.set_device(local_rank)
    dist = None
    force = world == 1 and os.environ.get("TG_BENCH_FORCE_COLLECTIVE") == "1"   # 1-GPU check of the RCCL gather path (one rank)
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))

    n = args.num_envs
    modes = {"edge_follow-v0": MODES, "surface_follow-v0": SURF_MODES, "object_balance-v0": BAL_MODES, "object_push-v0": PUSH_MODES,
             "object_roll-v0": ROLL_MODES, "surface_follow-v2": VERT_MODES}[args.env]
    if args.observation_mode:
        modes = dict(modes, observation_mode=args.observation_mode)
    act_dim = 3 if args.env == "surface_follow-v0" else 2
    max_steps = {"object_balance-v0": 250, "object_push-v0": 1000, "object_roll-v0": 250}.get(args.env, 200)   # params/*_params.py max_ep_len
    extra = dict(contact_mapping=args.contact_mapping, solver_iterations=args.solver_iters) if args.env in ("object_push-v0", "object_roll-v0") else {}
    venv = tg.make_vec(args.env, num_envs=n, max_steps=max_steps, image_size=[args.image_size, args.image_size], env_modes=modes,
                       seed=1 + rank * n, physics_dtype=args.physics, auto_reset=True, device=local_rank, obs_mode="torch",
                       pgs_full_sweeps=args.full_sweeps, **extra)
    # device-resident rollout: the step is enqueued on a torch stream and its outputs are consumed on that stream (no host wait per
    # step); --sync-steps restores the blocking VecEnv.step_wait behaviour
    shard = TorchShard(venv, pipelined=not args.sync_steps)
    env = ShardedVecEnv(shard, dist, overlap=True, force_collective=force, payload=os.environ.get("TG_BENCH_PAYLOAD", "auto")) if dist is not None else shard   # gather of step t overlaps the simulation of step t+1
    act_buf = torch.empty(n, act_dim, device="cuda", dtype=torch.float32)
    draw = [0]

    def actions():   # action_space.sample() for the whole batch: U[-0.25, 0.25), one device kernel on the env's stream (tg_sample_actions,
        draw[0] += 1  # counter based: draw k of rank r depends on (1234 + r, k) only), inside the timed region like every step's policy would be
        return venv.sample_actions(act_buf, 1234 + rank, draw[0])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    import contextlib
    on_stream = torch.cuda.stream(shard.stream) if shard.pipelined else contextlib.nullcontext()
    with on_stream:
        env.reset()
        for _ in range(args.warmup):
            env.step(actions())
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            env.step(actions())
        if dist is not None:
            env.flush()        # the last step's gather completes inside the timed region: K steps simulated AND delivered to rank 0
        barrier()
        dt = time.perf_counter() - t0
        # per-kernel durations for the roofline leg: HIP events on the launch stream, outside the timed region
        # (event records add ~2 x 4 launches of host work per step)
        venv.profile(True)
        for _ in range(min(args.steps, 50)):
            env.step(actions())
            venv.sync()
        barrier()
        prof = venv.profile_get()
        venv.profile(False)
        # SURVEY 8d asks for the rate with and without episode resets: a window that starts right after a reset of every env and
        # ends before any env can reach max_steps (an env that meets its goal early is still reset, as in any rollout)
        no_reset = None
        if dist is None:
            env.reset()
            for _ in range(10):
                env.step(actions())
            barrier()
            k_nr = max(10, min(args.steps, max_steps - 20))
            t_nr = time.perf_counter()
            for _ in range(k_nr):
                env.step(actions())
            barrier()
            dt_nr = time.perf_counter() - t_nr
            no_reset = {"value": round(n * k_nr / dt_nr, 1), "unit": "env-steps/s", "steps": k_nr, "ms_per_step": round(1e3 * dt_nr / k_nr, 4),
                        "what": "window between full-batch resets (no env reaches max_steps inside it)"}
    literal = None
    if world == 1 and not args.full_sweeps and not args.no_literal:
        # the same workload with the literal solver (every tick: dynamics + all 150 PGS sweeps), for comparison; short run
        lit = tg.make_vec(args.env, num_envs=n, max_steps=max_steps, image_size=[args.image_size, args.image_size], env_modes=modes,
                          seed=1 + rank * n, physics_dtype=args.physics, auto_reset=True, device=local_rank, obs_mode="torch",
                          pgs_full_sweeps=True, **extra)
        lshard = TorchShard(lit)
        lshard.reset()
        for _ in range(5):
            lshard.step(actions())
        torch.cuda.synchronize()
        lsteps = max(10, min(args.steps, 40))
        tl = time.perf_counter()
        for _ in range(lsteps):
            lshard.step(actions())
        torch.cuda.synchronize()
        ldt = time.perf_counter() - tl
        lit.profile(True)
        for _ in range(10):
            lshard.step(actions())
        torch.cuda.synchronize()
        lprof = lit.profile_get()
        literal = {"value": round(n * lsteps / ldt, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * ldt / lsteps, 4), "steps": lsteps,
                   "k_step_ms": round(lprof["step"][0] / max(lprof["step"][1], 1), 4),
                   "k_render_ms": round(lprof["render"][0] / max(lprof["render"][1], 1), 4),
                   "what": "pgs_full_sweeps=1: dynamics + exactly 150 Gauss-Seidel sweeps in every one of the 24 ticks (no convergence exit, no analytic fixed point)"}
        lit.close()
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / dt
        step_ms, step_n = prof["step"]
        rend_ms, rend_n = prof["render"]
        rst_ms, rst_n = prof["reset"]
        k_step = step_ms / max(step_n, 1)                      # ms per launch
        k_render_main = rend_ms / max(rend_n, 1)
        # the render kernel actually launched (csrc/tg_raster.hip: launch_render): small shared meshes (edge, pole, cube) take the
        # two-pass small-mesh kernel, the heightfield and the marble k_render_tactile<128,128>
        small = args.env in ("edge_follow-v0", "object_balance-v0", "object_push-v0") and args.image_size % 128 == 0
        render_name = "k_render_small<128,64,2>" if small else ("k_render_scatter" if args.env == "object_roll-v0" else "k_render_tactile")
        dominant = "k_step" if step_ms >= rend_ms else render_name
        dom_ms = k_step if dominant == "k_step" else k_render_main
        algo_bytes = {"edge_follow-v0": ALGO_BYTES_PER_ENV_STEP, "surface_follow-v0": ALGO_BYTES_SURFACE,
                      "object_balance-v0": args.image_size * args.image_size + 300.0,             # config 5: 65.8 KB at 256x256
                      "object_push-v0": args.image_size * args.image_size + 400.0,
                      "object_roll-v0": args.image_size * args.image_size + 400.0, "surface_follow-v2": ALGO_BYTES_SURFACE}[args.env]
        achieved = algo_bytes * n / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure measured with
        # rocprofv3 on this workload (profiles/r2_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) is reported when the
        # configuration matches, else null.
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r2_traffic.json")))
            for wl in tr["workloads"]:
                if (wl["env"], wl["num_envs"], wl["image_size"], wl["physics"]) == (args.env, n, args.image_size, args.physics) and not args.full_sweeps:
                    k = wl["k_step" if dominant == "k_step" else "k_render_tactile"]
                    traffic = {"bytes_per_launch": round((k["fetch_corrected_kb"] + k["write_kb"]) * 1024.0), "kernel": k["kernel"],
                               "source": "profiles/r2_traffic.json (rocprofv3 PMC: FETCH_SIZE x2 + WRITE_SIZE)",
                               "vs_algorithmic": round((k["fetch_corrected_kb"] + k["write_kb"]) / (algo_bytes * n / 1024.0), 3)}
        except (OSError, KeyError, ValueError):
            traffic = None
        out = {
            "metric": "env-steps/sec (128x128 tactile obs) at N envs", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.physics == "f64" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.env}, {modes['arm_type'].upper()} + {modes['tactile_sensor_name']}, {n} vec-envs per MI355X, {args.image_size}x{args.image_size} tactile obs, "
                                   f"random actions, TCP_velocity_control, {12 if args.env == 'object_balance-v0' else 24} sim ticks per step (PGS budget 150 sweeps per tick), auto-reset on",
                       "envs_per_gpu": n, "total_envs": total_envs, "parallelism": f"env-shard x{world}" + (" + one packed RCCL gather (obs u8" + (" interior pixels only, border ring restored on rank 0" if getattr(env, "_interior", None) is not None else "") + ", reward f32, done u8) to rank 0 per step, "
                                                                          "overlapped with the next step's simulation" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_env_step": algo_bytes,
                         "kernel_ms": {"k_step": round(k_step, 4), "k_render_tactile": round(k_render_main, 4),
                                       "k_reset_per_launch": round(rst_ms / max(rst_n, 1), 4),
                                       "k_render_tactile_masked": round(prof["render_masked"][0] / max(prof["render_masked"][1], 1), 4)},
                         "launches": {"k_step": step_n, "k_render_tactile": rend_n, "k_reset": rst_n},
                         "note": "HBM roofline in form only: the step is bound by per-workgroup latency chains (render) and by the serial solver "
                                 "(k_step), not by bytes; see DESIGN.md 4.3"},
            "solver": "literal: dynamics + 150 PGS sweeps every tick (pgs_full_sweeps)" if args.full_sweeps else
                      "default: PGS leaves at last-bit convergence; ticks whose motor solve is provably unclamped and demonstrably converged take "
                      "the solver's analytic fixed point (qd = target); joints within 1e-11 rad of the literal solver over 1024 envs x 260 steps incl. auto-resets "
                      "(tests/test_gpu_parity.py::test_default_solver_equals_literal_solver_at_config_scale; DESIGN.md 4.1)",
            "literal_solver": literal,
            "without_full_batch_reset": no_reset,
            "resets_in_timed_region": bool((args.warmup % 200) + args.steps >= 200),
        }
        if prof["scene"][1]:    # --observation-mode visual / visuotactile: the scene camera's two kernels (eye<-frame transforms + raster), per draw
            out["scene_camera"] = {"observation_mode": modes["observation_mode"], "ms_per_draw": round(prof["scene"][0] / prof["scene"][1], 4),
                                   "draws": prof["scene"][1]}
        if world == 1 and not args.no_cpu_baseline and args.env == "edge_follow-v0":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    venv.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
