import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, tactile_gym_amd as tg
from tactile_gym_amd import _capi as capi
from bench import SURF_MODES
v = tg.make_vec("surface_follow-v0", num_envs=1024, max_steps=200, image_size=[128,128], env_modes=SURF_MODES, seed=1, obs_mode="torch", auto_reset=False)
v.reset()
a = torch.empty(1024, 3, device="cuda")
for _ in range(20):
    v.step_async(a.uniform_(-0.25, 0.25)); v.sync()
L = capi.lib()
buf = np.zeros(64 * 8, dtype=np.uint64)
L.tg_debug_raster.restype = C.c_int
L.tg_debug_raster(buf.ctypes.data_as(C.POINTER(C.c_uint64)))
d = buf.reshape(64, 8).astype(np.int64)
print("survivors mean/max", d[:, 4].mean(), d[:, 4].max(), "rounds mean/max", d[:, 5].mean(), d[:, 5].max(), "records total mean/max", d[:, 6].mean(), d[:, 6].max())
print("us: stage", ((d[:, 7] - d[:, 0]) / 100).mean(), "cull", ((d[:, 1] - d[:, 7]) / 100).mean(), "survivors->records", ((d[:, 2] - d[:, 1]) / 100).mean(), "pixel loop", ((d[:, 3] - d[:, 2]) / 100).mean())
print("us: stage+cull", ((d[:, 1] - d[:, 0]) / 100).mean(), "rounds(setup+pixels)", ((d[:, 3] - d[:, 1]) / 100).mean(), "max", ((d[:, 3] - d[:, 1]) / 100).max(), "total to post", ((d[:, 3] - d[:, 0]) / 100).mean())
v.close()
