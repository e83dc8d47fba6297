import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import tactile_gym_amd as tg
from tactile_gym_amd.rl_envs import edge_follow as ef
from tactile_gym_amd.vec_env import TactileVecEnv
modes = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
def run(n, iters, full, steps=40):
    cfg, robot, sensor, mesh, m = ef.build_config(n, 200, [128,128], modes, "f64", True, 0)
    cfg.solver_iterations = iters; cfg.pgs_full_sweeps = full
    v = TactileVecEnv(cfg, robot, sensor, mesh, observation_mode="tactile", obs_mode="torch", seed=1)
    v.reset()
    a = (torch.rand(n,2,device='cuda')-0.5)*0.5
    for _ in range(5): v.step_async(a); v.sync()
    v.profile(True)
    for _ in range(steps): v.step_async((torch.rand(n,2,device='cuda')-0.5)*0.5); v.sync()
    p=v.profile_get(); v.close()
    return p['step'][0]/p['step'][1]
for iters, full in ((0,1),(32,1),(48,1),(64,1),(80,1),(100,1),(150,1),(150,0)):
    print(iters, full, 'k_step ms %.4f' % run(1024, iters, full), flush=True)
