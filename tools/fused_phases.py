"""Per-kernel durations of one configuration, launch by launch: HIP event pairs and the kernels' own clock side by side (tg_profile_enable(1);
TG_FUSED_STEP=1 in the environment times the one-launch step, csrc/tg_fused.hip).  The per-phase figures of profiles/r5_exp_fused_step.txt came
from a development build of this kernel with phase stamps (-DTG_FUSED_STAMPS, removed since).
usage: python tools/fused_phases.py [num_envs] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tactile_gym_amd as tg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
MODES = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
             reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=MODES, seed=1, auto_reset=True, obs_mode="torch")
venv.reset()
rng = np.random.default_rng(0)
for _ in range(20):
    venv.step(rng.uniform(-0.25, 0.25, size=(n, venv.act_dim)).astype(np.float32))
venv.profile(True)
for _ in range(steps):
    venv.step(rng.uniform(-0.25, 0.25, size=(n, venv.act_dim)).astype(np.float32))
prof = venv.profile_get()
venv.profile(False)
print(venv.step_mode(), {k: (round(1e3 * v[0] / max(v[1], 1), 2), v[1]) for k, v in prof.items()}, "(us per launch, launches)")
