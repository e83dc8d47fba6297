import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import tactile_gym_amd as tg
from oracle.ref_env import OracleObjectPushEnv
MODES = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=True, rand_obj_mass=True, traj_type="simplex",
             observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name=sys.argv[1] if len(sys.argv) > 1 else "digitac")
n, size, steps = 4, 128, 10
venv = tg.make_vec("object_push-v0", num_envs=n, max_steps=steps, image_size=[size, size], env_modes=MODES, seed=31, auto_reset=False)
oracles = [OracleObjectPushEnv(seed=31 + i, max_steps=steps, image_size=(size, size), env_modes=MODES) for i in range(n)]
rng = np.random.default_rng(5)
for ep in range(2):
    t0 = time.time(); obs = venv.reset(); print("reset s", time.time() - t0)
    ref = [o.reset() for o in oracles]
    st = venv.get_state()
    for i, o in enumerate(oracles):
        print("reset", ep, i, "ticks", st["reset_ticks"][i], o.reset_ticks, "dq %.2e" % np.abs(st["q"][i] - o.arm.q).max(),
              "dpos %.2e" % np.abs(st["body_pos"][i] - o.cube_pose()[0]).max(), "drot %.2e" % np.abs(st["body_rot"][i] - o.cube_pose()[1]).max(),
              "mass", st["obj_mass"][i] == o.cube.mass, "traj %.2e" % np.abs(st["traj"][i][:2, :10].T - o.traj_pos_work[:, :2]).max(),
              "yaw %.2e" % np.abs(st["traj"][i][2, :10] - o.traj_rpy_work[:, 2]).max(),
              "px", int((obs["tactile"][i] != ref[i]["tactile"]).sum()), "feat %.2e" % np.abs(obs["extended_feature"][i] - ref[i]["extended_feature"]).max())
    for step in range(steps):
        a = rng.uniform(-0.25, 0.25, size=(n, 2)).astype(np.float32)
        t0 = time.time(); obs, rew, done, _ = venv.step(a); dt = time.time() - t0
        st = venv.get_state()
        for i, o in enumerate(oracles):
            ro, rr, rd, _ = o.step(a[i])
            pos, R = o.cube_pose()
            print(ep, step, i, "dq %.2e" % np.abs(st["q"][i] - o.arm.q).max(), "dpos %.2e" % np.abs(st["body_pos"][i] - pos).max(),
                  "drot %.2e" % np.abs(st["body_rot"][i] - R).max(), "rew %.3e" % abs(rew[i] - rr), bool(done[i]) == rd, "gid", st["goal_id"][i], o.targ_traj_list_id,
                  "px", int((obs["tactile"][i] != ro["tactile"]).sum()), "nz", int((ro["tactile"] > 0).sum()),
                  "feat %.2e" % np.abs(obs["extended_feature"][i] - ro["extended_feature"]).max(), "t %.3f" % dt)
venv.close()
