import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import tactile_gym_amd as tg
for mm, ad in (("xyRz", 3), ("TyRz", 2)):
    MODES = dict(movement_mode=mm, control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
                 observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")
    venv = tg.make_vec("object_push-v0", num_envs=2, max_steps=5, image_size=[128, 128], env_modes=MODES, seed=31, auto_reset=False)
    venv.reset()
    a = np.zeros((2, ad), np.float32)
    venv.step(a)
    st = venv.get_state()
    print(mm, "qd_target", st["qd_target"][0], "\nq", st["q"][0], "\nqd", st["qd"][0], "\nbody", st["body_pos"][0], st["body_linvel"][0], st["body_angvel"][0], "\ntcp", st["tcp_pos"][0])
    venv.close()
