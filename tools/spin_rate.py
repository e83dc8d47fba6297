"""Rate of object_balance's spinning_plate mode (device-resident random rollout, auto-reset on).  usage: spin_rate.py [num_envs ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tactile_gym_amd as tg

MODES = dict(movement_mode="xyRxRy", control_mode="TCP_velocity_control", object_mode="spinning_plate", rand_gravity=True, rand_embed_dist=False,
             observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
if __name__ == "__main__":
    if os.environ.get("SPIN_ITERS"):          # experiment: the solver's share of the tick (results are no longer the reference's)
        from tactile_gym_amd.rl_envs import object_balance as ob
        _bc = ob.build_config
        def patched(*a, **k):
            out = _bc(*a, **k)
            out[0].solver_iterations = int(os.environ["SPIN_ITERS"])
            return out
        ob.build_config = patched
    for n in [int(x) for x in sys.argv[1:]] or [1024]:
        v = tg.make_vec("object_balance-v0", num_envs=n, max_steps=250, image_size=[128, 128], env_modes=MODES, seed=1, auto_reset=True, obs_mode="torch")
        v.reset()
        for k in range(20):
            v.step_random_async(5, k, restart=(k == 0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 100
        for k in range(K):
            v.step_random_async(5, 20 + k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        v.profile(True)
        for k in range(20):
            v.step_random_async(5, 200 + k)
        torch.cuda.synchronize()
        prof = {k: round(ms / max(cnt, 1), 4) for k, (ms, cnt) in v.profile_get().items() if cnt}
        st = v.get_state()
        print(f"spinning_plate {n} envs: {n / dt:.0f} env-steps/s, {1e3 * dt:.3f} ms/step; contacts now: {np.bincount(st['dish_state'][:, 19].astype(int), minlength=5).tolist()}; kernel ms {prof}")
        v.close()
