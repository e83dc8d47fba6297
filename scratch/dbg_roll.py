import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, tactile_gym_amd as tg
modes = dict(movement_mode="xy", control_mode="TCP_velocity_control", rand_init_obj_pos=False, rand_obj_size=True, rand_embed_dist=True,
             observation_mode="tactile_and_feature", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
n = 1024
venv = tg.make_vec("object_roll-v0", num_envs=n, max_steps=200, image_size=[128, 128], env_modes=modes, seed=5, auto_reset=False)
venv.reset(); st0 = venv.get_state()
a = np.tile(np.array([[0.25, 0.0]], dtype=np.float32), (n, 1))
for k in range(5):
    venv.step(a)
st = venv.get_state()
tip = np.linalg.norm(st["tcp_pos"][:, :2] - st0["tcp_pos"][:, :2], axis=1)
ball = np.linalg.norm(st["body_pos"][:, :2] - st0["body_pos"][:, :2], axis=1)
bad = np.nonzero((st["embed_dist"] > 0.0019) & (np.abs(ball / tip - 0.5) > 0.05))[0]
print("bad", bad[:10], len(bad))
for i in bad[:5]:
    print(i, "embed", st["embed_dist"][i], "r", st["obj_mass"][i], "ball", ball[i], "tip", tip[i], "ticks", st0["reset_ticks"][i], "tcp z", st0["tcp_pos"][i, 2], "target z", 2 * st["obj_mass"][i] - st["embed_dist"][i], "body z", st["body_pos"][i, 2])
venv.close()
