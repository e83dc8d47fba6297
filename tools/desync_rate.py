"""How much the headline depends on every env finishing its episode in the same step.

bench.py's rollout starts all envs together and edge_follow's random policy almost never reaches the goal, so all 1024 envs hit max_steps in
the same step and the licences of sim_tick's analytic fixed point renew in lockstep: seven of eight launches run the light
path in every wavefront.  An RL run does not stay aligned.  This script staggers the episodes (masked resets of random subsets during the
first max_steps steps) and times the same device-resident random rollout, then the aligned one, on the same box.

Round 5, MI355X, 1024 envs (profiles/r5_final_desync.txt): aligned 24.1 M env-steps/s either way; staggered 6.2 M before round 5's two changes (the
licence dropped at every reset: k_step 49 us in every launch; every reset on the spot: 90 - 110 us per launch), 18.3 M with them (the reset
renews the licence; reset bank on by default, its refill paced without a cross-stream wait).  TG_RESET_BANK=0 shows the on-the-spot resets.

Round 6 (three surfaces per env, one render launch, a swap-in that loads before it stores - DESIGN.md 4.1h): edge_follow 21.1 M, surface_follow-v0
7.0 -> 9.2 M with the episodes out of phase.

    python tools/desync_rate.py [--env edge_follow-v0] [--envs 1024] [--steps 2000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(stagger, n, steps, max_steps, env_id="edge_follow-v0"):
    import torch
    import tactile_gym_amd as tg
    from bench import ENV_MODES
    v = tg.make_vec(env_id, num_envs=n, max_steps=max_steps, image_size=[128, 128], env_modes=ENV_MODES[env_id], seed=1, auto_reset=True, obs_mode="torch")
    v.reset()
    rng = np.random.default_rng(0)
    draw = 0
    if stagger:
        phase = rng.integers(0, max_steps, size=n)           # env e's episodes start at steps = phase[e] (mod max_steps)
        for k in range(max_steps):
            v.step_random_async(5, draw, restart=(draw == 0)); draw += 1
            m = (phase == k).astype(np.uint8)
            if m.any():
                v.reset(m)
    for k in range(100):
        v.step_random_async(5, draw, restart=(draw == 0)); draw += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        v.step_random_async(5, draw); draw += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    bank = v.bank_stats()
    v.profile("clock")                                        # the kernels' own clock, the step still one graph
    for k in range(400):
        v.step_random_async(5, draw); draw += 1
    torch.cuda.synchronize()
    prof = {k: round(ms / max(cnt, 1) * 1e3, 2) for k, (ms, cnt) in v.profile_get().items() if k.endswith("_clock") and cnt}
    v.profile(False)
    v.close()
    return dt, {"bank": bank, "kernel_us": prof}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--env", default="edge_follow-v0")
    a = ap.parse_args()
    out = {}
    for name, st in (("aligned", False), ("staggered", True)):
        dt, counts = run(st, a.envs, a.steps, a.max_steps, a.env)
        out[name] = {"ms_per_step": round(dt * 1e3, 4), "env_steps_per_s": round(a.envs / dt, 1), **counts}
    print(json.dumps(out))
