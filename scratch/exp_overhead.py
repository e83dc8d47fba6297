import sys, time
sys.path.insert(0,'/root/repo')
import torch, numpy as np
import tactile_gym_amd as tg
from tactile_gym_amd.parallel import TorchShard
modes = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
n=1024
venv = tg.make_vec("edge_follow-v0", num_envs=n, max_steps=200, image_size=[128,128], env_modes=modes, seed=1, obs_mode="torch")
sh=TorchShard(venv); sh.reset()
a=torch.empty(n,2,device='cuda')
def bench(fn, k=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/k*1e3
print('full step (uniform_+step)     %.3f ms'%bench(lambda: sh.step(a.uniform_(-0.25,0.25))))
print('step only (fixed actions)     %.3f ms'%bench(lambda: sh.step(a)))
def raw():
    venv.step_async(a); venv.sync()
print('step_async+sync               %.3f ms'%bench(raw))
import ctypes as C
from tactile_gym_amd import _capi as capi
L=capi.lib(); ctx=venv._ctx; ptr=C.c_void_p(a.data_ptr())
def raw2():
    L.tg_step(ctx, ptr, 1); L.tg_sync(ctx)
print('tg_step+tg_sync               %.3f ms'%bench(raw2))
def raw3():
    for _ in range(10): L.tg_step(ctx, ptr, 1)
    L.tg_sync(ctx)
print('10x tg_step then sync (per)   %.3f ms'%(bench(raw3,20)/10))
