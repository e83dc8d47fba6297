"""Measurement aid: env-steps/s of object_balance-v0 with object_mode "ball_on_plate" (lane mapping, full 12-row solve on every tick) beside the
pole's, same modes otherwise (BASELINE configs[4]: 256 x 256, 1024 envs per GPU).  Usage: python tools/ball_rate.py [num_envs ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import tactile_gym_amd as tg  # noqa: E402

BAL = dict(movement_mode="xy", control_mode="TCP_velocity_control", object_mode="pole", rand_gravity=True, rand_embed_dist=True,
           observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")


def rate(mode, n, steps=200, size=256):
    v = tg.make_vec("object_balance-v0", num_envs=n, max_steps=250, image_size=[size, size], env_modes=dict(BAL, object_mode=mode), seed=3,
                    auto_reset=True, obs_mode="torch")
    v.reset()
    import torch
    a = torch.from_numpy(np.random.default_rng(0).uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)).cuda()
    for _ in range(20):
        v.step_async(a)
    v.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        v.step_async(a)
    v.sync()
    dt = time.perf_counter() - t0
    v.close()
    return n * steps / dt, dt / steps * 1e3


if __name__ == "__main__":
    for n in [int(x) for x in sys.argv[1:]] or [1024, 8192]:
        for mode in ("pole", "ball_on_plate"):
            r, ms = rate(mode, n)
            print(f"object_balance {mode:14s} {n:6d} envs 256x256: {r / 1e6:.3f} M env-steps/s ({ms:.3f} ms per step)", flush=True)
