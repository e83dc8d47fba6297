import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, tactile_gym_amd as tg
from tactile_gym_amd import _capi as capi
from bench import MODES
v = tg.make_vec("edge_follow-v0", num_envs=1024, max_steps=200, image_size=[128,128], env_modes=MODES, seed=1, obs_mode="torch")
v.reset()
a = torch.empty(1024, 2, device="cuda")
L = capi.lib(); L.tg_debug_kstep.restype = C.c_int
acc = []
for it in range(30):
    v.step_async(a.uniform_(-0.25, 0.25)); v.sync()
    buf = np.zeros(16 * 8, dtype=np.uint64)
    L.tg_debug_kstep(buf.ctypes.data_as(C.POINTER(C.c_uint64)))
    d = buf.reshape(16, 8).astype(np.int64)
    acc.append((d[:, 1:6] - d[:, 0:5]) / 100.0)
acc = np.array(acc)   # [it, wg, phase]
names = ["load state", "encode+trig_init", "tcp_velocity_control", "ticks", "store+finish_env"]
for i, nm in enumerate(names):
    print(f"{nm:22s} mean {acc[5:, :, i].mean():6.2f} us   per-step means {np.round(acc[5:21, :, i].mean(axis=1), 1)}")
print("total", acc[5:].sum(axis=2).mean())
v.close()
