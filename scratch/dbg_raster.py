import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np, tactile_gym_amd as tg
from tactile_gym_amd import _capi
from bench import MODES, SURF_MODES
L = C.CDLL(_capi.LIB_PATH)
for env_id, modes, ad in (("edge_follow-v0", MODES, 2), ("surface_follow-v0", SURF_MODES, 3)):
    v = tg.make_vec(env_id, num_envs=1024, max_steps=200, image_size=[128,128], env_modes=modes, seed=1, obs_mode="torch")
    v.reset()
    a = torch.empty(1024, ad, device="cuda")
    buf = (C.c_ulonglong * 8)()
    L.tg_debug_raster(buf); b0 = np.array(buf[:], dtype=np.float64)
    for _ in range(20):
        v.step_async(a.uniform_(-0.25, 0.25)); v.sync()
    L.tg_debug_raster(buf); b1 = np.array(buf[:], dtype=np.float64)
    d = b1 - b0; nwg = d[6]
    print(env_id, "WGs", nwg, "final phase cycles: issue loads %.0f, wait loads %.0f, compute+issue stores %.0f, drain stores %.0f" % (d[0]/nwg, d[1]/nwg, d[2]/nwg, d[3]/nwg))
    v.close()
