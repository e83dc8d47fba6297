"""Test infrastructure: rollouts of many CPU-oracle envs on the host cores (one `spawn`ed process per chunk of envs, so the children
never inherit the parent's HIP runtime state).  Used by tests/test_gpu_config_scale.py (HIP vs oracle at the BASELINE configs' own batch
sizes and over whole episodes).  Not collected by pytest."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rollout_chunk(args):
    """One chunk of oracle envs: (cls, kwargs, seed0, idx, actions[steps, len(idx), act_dim], auto_reset, follow) -> per-env record.
    auto_reset: an env that reports done is reset before the next step and the frame kept for that step is the fresh episode's first
    observation (SB3 VecEnv convention, what the device does); its terminal image is kept separately.
    follow: optional int array [steps, len(idx)] of the device's goal index after each step (object_push only): where the oracle's
    goal advance differs from it on the documented knife edge (PARITY_ASSUMPTIONS A29) the oracle is made to follow."""
    cls, kwargs, seed0, idx, actions, auto_reset, follow = args[:7]
    digest = len(args) > 7 and args[7]       # keep crc32 of every frame instead of the frame (256 x 256 x 250 steps x 64 envs is 1 GB through a pipe)
    import zlib
    keep = (lambda im: np.uint32(zlib.crc32(np.ascontiguousarray(im).tobytes()))) if digest else (lambda im: im.copy())
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    warnings.simplefilter("ignore")
    from oracle import ref_env
    out = []
    kwargs = dict(kwargs)
    res_thr = kwargs.pop("solver_residual_threshold", 0.0)      # btContactSolverInfo::m_leastSquaresResidualThreshold (PARITY A7b)
    for k, i in enumerate(idx):
        o = getattr(ref_env, cls)(seed=seed0 + i, **kwargs)
        o.solver_residual_threshold = res_thr
        ob = o.reset()
        rec = dict(img=[keep(ob["tactile"][..., 0])], q=[o.arm.q.copy()], rew=[], done=[], reset_ticks=[o.reset_ticks], knife=0,
                   term={}, feat=[], cc=[], cid=[], body=[], xf=[np.asarray(o.stimulus_transform(), dtype=np.float32).ravel().copy()])
        push = cls == "OracleObjectPushEnv"
        has_body = hasattr(o, "cube_pose") or hasattr(o, "body_pose")
        for s in range(actions.shape[0]):
            goal_before = o.goal_pos_world.copy() if push else None
            sweeps_before = o.sweeps_total
            ob, r, d, _ = o.step(actions[s, k])
            rec.setdefault("sweeps", []).append(o.sweeps_total - sweeps_before)   # PGS sweeps of this step's ticks (before any auto-reset)
            if push and follow is not None and follow[s, k] == o.targ_traj_list_id + 1:
                pos = o.cube_pose()[0]
                if abs(np.linalg.norm(pos - goal_before) - o.termination_pos_dist) < 1e-12:
                    assert o._update_goal()
                    ob = o._observation()
                    rec["knife"] += 1
            rec["rew"].append(r), rec["done"].append(d)
            q_step = o.arm.q.copy()
            if push:
                rec["cc"].append(int(o.scene.n_contacts)), rec["cid"].append(np.array(o.scene.contact_ids, dtype=np.int32).copy())
                rec["goal_id"] = rec.get("goal_id", []) + [int(o.targ_traj_list_id)]
            if "extended_feature" in ob:
                rec["feat"].append(np.asarray(ob["extended_feature"], dtype=np.float32).copy())
            img = keep(ob["tactile"][..., 0])
            xf = np.asarray(o.stimulus_transform(), dtype=np.float32).ravel().copy()
            if d and auto_reset:
                rec["term"][s] = img
                ob = o.reset()
                rec["reset_ticks"].append(o.reset_ticks)
                img = keep(ob["tactile"][..., 0])
                q_step = o.arm.q.copy()
                xf = np.asarray(o.stimulus_transform(), dtype=np.float32).ravel().copy()
            if has_body:                           # after an auto-reset: the fresh episode's object pose, like q (what the device state holds)
                p, R = o.cube_pose() if hasattr(o, "cube_pose") else o.body_pose()
                rec["body"].append(np.concatenate([np.asarray(p).ravel(), np.asarray(R).ravel()]))
            rec["img"].append(img), rec["q"].append(q_step), rec["xf"].append(xf)
        for key in ("img", "q", "rew", "done", "feat", "cc", "cid", "body", "xf", "sweeps"):
            rec[key] = np.asarray(rec[key])
        out.append(rec)
    return list(idx), out


def oracle_rollouts(cls, kwargs, seed0, actions, auto_reset=False, follow=None, procs=None, digest=False):
    """actions: float32 [steps, n, act_dim].  Returns a list of n per-env records (see _rollout_chunk)."""
    import multiprocessing as mp
    n = actions.shape[1]
    procs = procs or min(128, os.cpu_count() or 8, n)
    chunks = [c for c in np.array_split(np.arange(n), min(n, 4 * procs)) if len(c)]
    jobs = [(cls, kwargs, seed0, list(map(int, c)), np.ascontiguousarray(actions[:, c]), auto_reset, None if follow is None else follow[:, c], digest)
            for c in chunks]
    recs = [None] * n
    with mp.get_context("spawn").Pool(procs) as pool:
        for idx, out in pool.imap_unordered(_rollout_chunk, jobs):
            for i, r in zip(idx, out):
                recs[i] = r
    return recs
