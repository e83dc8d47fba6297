"""k_render_blocks (csrc/tg_raster.hip) against the other ways this library can draw the same frames.

The block kernel skips work in three places - blocks no triangle can change (a depth-plane bound against a host-made table), records that
reach none of a round's blocks, and blocks that already hold the untouched-sensor image from the launch before (`RasterParams::drawn`) - and
none of them may change a pixel.  The oracle comparisons (tests/test_gpu_parity.py, tests/test_gpu_config_scale.py) already run through it;
here the same rollouts are drawn three times in child processes (the switches are read once per process):
    default                       k_render_blocks, dirty-block skipping on
    TG_RASTER_REWRITE_ALL=1       k_render_blocks, every launch rewrites every block
    TG_NO_BLOCK_RASTER=1          k_render_small (two workgroups per image, no block tests at all)
and every observation, terminal observation, reward and done flag must be identical byte for byte, over episodes that end and restart
(short max_steps: the terminal image goes to its own buffer, the masked re-draw saves the old image first)."""
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
modes = json.loads(sys.argv[6])
venv = tg.make_vec(env_id, num_envs=n, max_steps=max_steps, image_size=[size, size], env_modes=modes, seed=11, auto_reset=True)
h = hashlib.sha256()
rng = np.random.default_rng(3)
obs = venv.reset()
h.update(np.ascontiguousarray(obs["tactile"] if isinstance(obs, dict) else obs).tobytes())
sums, dones, terms = [], 0, 0
for k in range(steps):
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
print(json.dumps({"sha": h.hexdigest(), "sums": sums, "dones": dones, "terms": terms}))
"""

EDGE = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
            reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
PUSH = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
            observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")
EDGE_DIGIT = dict(EDGE, tactile_sensor_name="digit", movement_mode="xyRz")


def _run(env_id, n, size, steps, max_steps, modes, **switches):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **switches)
    out = subprocess.run([sys.executable, "-c", CHILD, env_id, str(n), str(size), str(steps), str(max_steps), json.dumps(modes)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("env_id,n,size,steps,max_steps,modes", [("edge_follow-v0", 300, 128, 40, 12, EDGE), ("edge_follow-v0", 33, 256, 20, 7, EDGE_DIGIT),
                                                                  ("object_push-v0", 64, 128, 24, 9, PUSH)])
def test_block_raster_draws_what_the_two_pass_raster_draws(env_id, n, size, steps, max_steps, modes):
    a = _run(env_id, n, size, steps, max_steps, modes)
    b = _run(env_id, n, size, steps, max_steps, modes, TG_RASTER_REWRITE_ALL="1")
    c = _run(env_id, n, size, steps, max_steps, modes, TG_NO_BLOCK_RASTER="1")
    assert a["dones"] >= n and a["terms"] == a["dones"]                      # episodes ended and restarted inside the rollout
    assert len(set(a["sums"])) > steps // 2                                    # the frames are not all alike (contacts come and go)
    assert a["sums"] == b["sums"] == c["sums"]
    assert a["sha"] == b["sha"], "skipping blocks that hold the untouched-sensor image changed a frame"
    assert a["sha"] == c["sha"], "k_render_blocks and k_render_small drew different frames"
