"""How far from bit-exact are the tactile images of the envs whose camera transform goes through FMA-contracted f64 (free-body and MG400 envs;
VERDICT r1 item 5)?  HIP vs the CPU oracle, same seeds and actions: histogram of the number of differing pixels per image and of the size of the
differences.  Runs on the GPU box (the oracle envs run on the host cores, one process per chunk).  A checker, hence under tests/ (not collected by pytest).  python tests/pixel_mismatch_histogram.py"""
import os, sys, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import multiprocessing as mp

CASES = [   # env id, oracle class, modes, envs, steps, act_dim, size
    ("edge_follow-v0", "OracleEdgeFollowEnv", dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile",
                                                  reward_mode="dense", arm_type="mg400", tactile_sensor_name="tactip"), 1024, 8, 2, 128),
    ("object_balance-v0", "OracleObjectBalanceEnv", dict(movement_mode="xy", control_mode="TCP_velocity_control", object_mode="pole", rand_gravity=True, rand_embed_dist=True,
                                                        observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip"), 1024, 8, 2, 256),
    ("object_push-v0", "OracleObjectPushEnv", dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=True, rand_obj_mass=True, traj_type="simplex",
                                                  observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac"), 1024, 8, 2, 128),
]


def _oracle_chunk(args):
    cls, modes, size, seed0, idx, actions = args
    warnings.simplefilter("ignore")
    from oracle import ref_env
    out = []
    for i in idx:
        o = getattr(ref_env, cls)(seed=seed0 + i, max_steps=1000, image_size=(size, size), env_modes=modes)
        imgs = [o.reset()["tactile"][..., 0].copy()]
        for a in actions[:, i]:
            imgs.append(o.step(a)[0]["tactile"][..., 0].copy())
        out.append(np.stack(imgs))
    return idx, out


def main():
    import tactile_gym_amd as tg
    warnings.simplefilter("ignore")
    for env_id, cls, modes, n, steps, act_dim, size in CASES:
        rng = np.random.default_rng(7)
        actions = rng.uniform(-0.25, 0.25, size=(steps, n, act_dim)).astype(np.float32)
        v = tg.make_vec(env_id, num_envs=n, max_steps=1000, image_size=[size, size], env_modes=modes, seed=900, auto_reset=False)
        hip = [v.reset()["tactile"][..., 0].copy()]
        for s in range(steps):
            hip.append(v.step(actions[s])[0]["tactile"][..., 0].copy())
        v.close()
        hip = np.stack(hip, axis=1)                                   # [n, steps + 1, H, W]
        chunks = np.array_split(np.arange(n), min(n, 128))
        with mp.Pool(min(128, os.cpu_count() or 8)) as pool:
            res = pool.map(_oracle_chunk, [(cls, modes, size, 900, list(c), actions) for c in chunks])
        ref = np.zeros_like(hip)
        for idx, out in res:
            for i, im in zip(idx, out):
                ref[i] = im
        diff = hip.astype(np.int16) - ref.astype(np.int16)
        per_image = (diff != 0).reshape(n * (steps + 1), -1).sum(1)
        hist = np.bincount(per_image, minlength=5)
        mags = np.bincount(np.abs(diff[diff != 0]), minlength=3)
        print(f"{env_id} ({modes['arm_type']} + {modes['tactile_sensor_name']}, {size}x{size}): {n} envs x (reset + {steps} steps) = {per_image.size} images; "
              f"differing pixels per image: " + ", ".join(f"{k}: {c}" for k, c in enumerate(hist) if c) +
              f"; max {per_image.max()}; |difference| of those pixels: " + (", ".join(f"{k}: {c}" for k, c in enumerate(mags) if c) or "none"), flush=True)


if __name__ == "__main__":
    main()
