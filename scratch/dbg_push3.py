import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import tactile_gym_amd as tg
from tactile_gym_amd.rl_envs import object_push as op
orig = op.build_config
for rep, iters, dt in ((1, 0, 'f64'), (1, 150, 'f64'), (24, 150, 'f64'), (24, 150, 'f32')):
    def patched(*a, **k):
        out = orig(*a, **k)
        out[0].action_repeat, out[0].solver_iterations = rep, iters
        return out
    op.build_config = patched
    MODES = dict(movement_mode="xyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
                 observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")
    venv = op.ObjectPushVecEnv(2, max_steps=5, image_size=[128, 128], env_modes=MODES, seed=31, auto_reset=False, physics_dtype=dt)
    venv.reset()
    st0 = venv.get_state()
    venv.step(np.zeros((2, 3), np.float32))
    st = venv.get_state()
    print(rep, iters, dt, "reset_ticks", st0["reset_ticks"], "q0", st0["q"][0][:3], "q", st["q"][0][:3], "qd", st["qd"][0][:3], "body", st["body_pos"][0], st["body_linvel"][0], st["body_angvel"][0])
    venv.close()
