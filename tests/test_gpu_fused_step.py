"""k_step_render (csrc/tg_fused.hip: the env step, its auto-reset and its tactile image(s) in ONE launch, the wavefront that steps an env
draws it) against the three-launch sequence it replaces (k_step -> k_reset -> k_render_blocks).

The step and reset code is the same device functions (tg_kernels.hpp: step_env, reset_or_swap); the raster is k_render_blocks' arithmetic on a
different lane mapping (tg_raster_dev.hpp: render_blocks_wave).  The same rollouts run in child processes with TG_FUSED_STEP=1 and =0 (the
switch is read once per context) and every observation, terminal observation, reward, done flag, reset tick count and step count must be
identical byte for byte - joint angles to 1e-12 rad (two translation units, FMA contraction) - over episodes that end and restart - at one env per wavefront (n <= 1024), several (n > 1024: the wavefront steps
E = ceil(n / 1024) envs in its first lanes and draws them one after the other), ragged last groups, 256 x 256 images (four block regions), the
MG400 (reset bank on: the swap-in inside the fused launch), a rewritten image buffer and the random-action step (tg_step_random).  The one-launch
step is opt-in (it measured slower); this file pins it to the default path, which the oracle comparisons run through."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
import numpy as np
import tactile_gym_amd as tg
env_id, n, size, steps, max_steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
modes = json.loads(sys.argv[6]); random_step = int(sys.argv[7])
venv = tg.make_vec(env_id, num_envs=n, max_steps=max_steps, image_size=[size, size], env_modes=modes, seed=11, auto_reset=True)
KEYS = ("reset_ticks", "step_count")
h, hs = hashlib.sha256(), {k: hashlib.sha256() for k in KEYS}
traj = []
rng = np.random.default_rng(3)
obs = venv.reset()
h.update(np.ascontiguousarray(obs["tactile"] if isinstance(obs, dict) else obs).tobytes())
sums, dones, terms = [], 0, 0
for k in range(steps):
    if random_step:
        venv.step_random_async(77, k, restart=(k == 0))
        obs, rew, done, infos = venv.step_wait()
    else:
        a = rng.uniform(-0.25, 0.25, size=(n, venv.act_dim)).astype(np.float32)
        obs, rew, done, infos = venv.step(a)
    img = np.ascontiguousarray(obs["tactile"] if isinstance(obs, dict) else obs)
    h.update(img.tobytes()); h.update(np.asarray(rew, dtype=np.float32).tobytes()); h.update(np.asarray(done, dtype=np.uint8).tobytes())
    sums.append(int(img.astype(np.int64).sum()))
    dones += int(np.sum(done))
    for i in range(n):
        if done[i]:
            t = infos[i]["terminal_observation"]
            h.update(np.ascontiguousarray(t["tactile"] if isinstance(t, dict) else t).tobytes()); terms += 1
    if k % 5 == 4 or k == steps - 1:
        st = venv.get_state()
        for key in KEYS:
            hs[key].update(np.ascontiguousarray(st[key]).tobytes())
        traj.append([st[key][:256].tolist() for key in ("q", "qd", "tcp_pos")])
print(json.dumps({"sha": h.hexdigest(), "state": {k: v.hexdigest()[:16] for k, v in hs.items()}, "traj": traj, "sums": sums, "dones": dones, "terms": terms, "mode": venv.step_mode()}))
"""

EDGE = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
EDGE_DIGIT = dict(EDGE, tactile_sensor_name="digit", movement_mode="xyRz")
EDGE_MG400 = dict(EDGE, arm_type="mg400", tactile_sensor_name="digitac", movement_mode="xy")


def _run(env_id, n, size, steps, max_steps, modes, random_step=0, **switches):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **switches)
    out = subprocess.run([sys.executable, "-c", CHILD, env_id, str(n), str(size), str(steps), str(max_steps), json.dumps(modes), str(random_step)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def _same_trajectories(a, b):
    """Joint angles, joint velocities and the TCP position of (up to) the first 256 envs at every fifth step.  The two launches run the same
    device functions, but in two translation units whose f64 expressions the compiler contracts into FMAs independently: the last bits may
    differ (measured: < 1e-15 rad); anything above 1e-12 would be a different computation."""
    import numpy as np
    for ta, tb in zip(a["traj"], b["traj"]):
        for xa, xb in zip(ta, tb):
            assert np.max(np.abs(np.asarray(xa) - np.asarray(xb))) < 1e-12


@pytest.mark.parametrize("n,size,steps,max_steps,modes,random_step", [
    (300, 128, 40, 12, EDGE, 0),             # one env per wavefront
    (1024, 128, 30, 9, EDGE, 1),             # the headline's shape, random-action step (the draw counter inside the fused launch)
    (2500, 128, 24, 10, EDGE, 0),            # three envs per wavefront, ragged last group
    (33, 256, 20, 7, EDGE_DIGIT, 0),         # four block regions per image, rotation actions
    (96, 128, 14, 6, EDGE_MG400, 0),         # MG400: reset bank on (swap-in or on-the-spot reset inside the launch)
])
def test_one_launch_step_equals_the_three_launch_step(n, size, steps, max_steps, modes, random_step):
    a = _run("edge_follow-v0", n, size, steps, max_steps, modes, random_step, TG_FUSED_STEP="1")
    b = _run("edge_follow-v0", n, size, steps, max_steps, modes, random_step, TG_FUSED_STEP="0")
    assert a["mode"] == "fused" and b["mode"] == "separate"
    assert a["dones"] >= n and a["terms"] == a["dones"]                      # episodes ended and restarted inside the rollout
    assert len(set(a["sums"])) > steps // 2
    assert a["sums"] == b["sums"]
    assert a["sha"] == b["sha"], "observations / rewards / dones / terminal observations differ between the one-launch and the three-launch step"
    assert a["state"] == b["state"], "reset tick counts / step counts differ"
    _same_trajectories(a, b)


def test_one_launch_step_with_every_block_rewritten():
    a = _run("edge_follow-v0", 200, 128, 30, 11, EDGE, 0, TG_FUSED_STEP="1")
    b = _run("edge_follow-v0", 200, 128, 30, 11, EDGE, 0, TG_FUSED_STEP="1", TG_RASTER_REWRITE_ALL="1")
    c = _run("edge_follow-v0", 200, 128, 30, 11, EDGE, 0, TG_FUSED_STEP="0", TG_NO_BLOCK_RASTER="1")
    assert a["mode"] == b["mode"] == "fused" and c["mode"] == "separate"
    assert a["sha"] == b["sha"] == c["sha"] and a["state"] == c["state"]
    _same_trajectories(a, c)


SURF = dict(movement_mode="xyzRxRy", control_mode="TCP_velocity_control", noise_mode="simplex", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="digit")


@pytest.mark.parametrize("env_id,n,steps,max_steps,modes,random_step,bank", [
    ("edge_follow-v0", 1024, 40, 9, EDGE, 1, "1"),        # the headline's shape: bank swap-in (or the reset on the spot) inside k_step<.., true>
    ("edge_follow-v0", 300, 30, 7, EDGE, 0, "0"),         # no bank: the reset stays the k_reset launch (a reset computed in another kernel instantiation may differ in the last bits)
    ("surface_follow-v0", 200, 30, 8, SURF, 0, "1"),      # phase 1 (task draws / swap-in) inside the step, k_gen_surface + phase 2 behind it
    ("surface_follow-v0", 130, 24, 8, SURF, 1, "0"),
])
def test_reset_inside_the_step_launch_equals_the_reset_launch(env_id, n, steps, max_steps, modes, random_step, bank):
    """Round 6: with the reset bank on, k_step<T, 0, true> runs reset_or_swap - the body of k_reset - on the finished env's lane straight after its
    step (one dependent launch fewer per step).  TG_NO_INLINE_RESET=1 keeps the k_reset launch: same images, terminal images, rewards, dones, tick
    counts.  (Bank off: both runs take the k_reset launch.)"""
    a = _run(env_id, n, 128, steps, max_steps, modes, random_step, TG_FUSED_STEP="0", TG_RESET_BANK=bank)
    b = _run(env_id, n, 128, steps, max_steps, modes, random_step, TG_FUSED_STEP="0", TG_RESET_BANK=bank, TG_NO_INLINE_RESET="1")
    assert a["dones"] >= n and a["terms"] == a["dones"]
    assert a["sums"] == b["sums"]
    assert a["sha"] == b["sha"], "observations / rewards / dones / terminal observations differ between the in-step reset and the k_reset launch"
    assert a["state"] == b["state"], "reset tick counts / step counts differ"
    _same_trajectories(a, b)
