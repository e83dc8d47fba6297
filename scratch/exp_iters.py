import sys, time, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import tactile_gym_amd as tg
from tactile_gym_amd.rl_envs import edge_follow as ef
modes = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
def run(n, dtype, iters, repeat=24, steps=30):
    cfg, robot, sensor, mesh, m = ef.build_config(n, 200, [128,128], modes, dtype, True, 0)
    cfg.solver_iterations = iters; cfg.action_repeat = repeat
    from tactile_gym_amd.vec_env import TactileVecEnv
    v = TactileVecEnv(cfg, robot, sensor, mesh, observation_mode="tactile", obs_mode="torch", seed=1)
    v.reset()
    a = (torch.rand(n,2,device='cuda')-0.5)*0.5
    for _ in range(3): v.step_async(a); v.sync()
    v.profile(True)
    t0=time.perf_counter()
    for _ in range(steps): v.step_async(a); v.sync()
    dt=(time.perf_counter()-t0)/steps
    p=v.profile_get(); v.close()
    return dt*1e3, p['step'][0]/p['step'][1], p['render'][0]/p['render'][1], p['reset'][0]/p['reset'][1]
for n in (1024, 16384):
  for dtype in ('f64','f32'):
    for iters in (0,1,150):
        print(n,dtype,iters,'ms/step %.3f k_step %.3f render %.4f reset %.4f'%run(n,dtype,iters), flush=True)
