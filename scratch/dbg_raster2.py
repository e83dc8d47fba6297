import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, tactile_gym_amd as tg
from tactile_gym_amd import _capi as capi
from bench import MODES
v = tg.make_vec("edge_follow-v0", num_envs=1024, max_steps=200, image_size=[128,128], env_modes=MODES, seed=1, obs_mode="torch")
v.reset()
a = torch.empty(1024, 2, device="cuda")
for _ in range(30):
    v.step_async(a.uniform_(-0.25, 0.25)); v.sync()
L = capi.lib()
buf = np.zeros(64 * 2 * 10, dtype=np.uint64)
L.tg_debug_raster.restype = C.c_int
L.tg_debug_raster(buf.ctypes.data_as(C.POINTER(C.c_uint64)))
d = buf[:1024].reshape(64, 2, 8).astype(np.int64); w = buf[1024:].reshape(64, 2, 2).astype(np.int64)
wd = (w[:, :, 1] - w[:, :, 0]).reshape(-1); cd = (d[:, :, 6] - d[:, :, 0]).reshape(-1)
ws = np.sort((w[:, :, 0] - w[:, :, 0].min()).reshape(-1)) / 100.0; we = np.sort((w[:, :, 1] - w[:, :, 0].min()).reshape(-1)) / 100.0
print('wall start us:', ' '.join(str(x) for x in np.round(ws, 1)[::4])); print('wall end us:', ' '.join(str(x) for x in np.round(we, 1)[::4])); print('wall dur us: min', wd.min() / 100, 'max', wd.max() / 100)
print('wall ticks (100 MHz) per WG mean', wd.mean(), 'clock64 per WG mean', cd.mean(), 'ratio clock64/wall', (cd / np.maximum(wd, 1)).mean(), 'wall span all WGs (10 ns units)', w[:, :, 1].max() - w[:, :, 0].min())
t = d[:, :, :7] - d[:, :, :1]
names = ["start", "M", "setup+bar", "p0 loop", "p0 post", "p1 loop", "p1 post"]
st = d[:, :, 0] - d[:, :, 0].min(); en = d[:, :, 6] - d[:, :, 0].min()
print("start times (cycles since first sampled start): ", np.sort(st.reshape(-1))[::8])
print("end times: ", np.sort(en.reshape(-1))[::8])
S = d[:, :, 0].reshape(-1); E = d[:, :, 6].reshape(-1)
order = np.argsort(S); S, E = S[order], E[order]
cl = np.concatenate([[0], np.cumsum(np.diff(S) > 10_000_000)])
for c in np.unique(cl):
    m = cl == c
    print("cluster", c, "n", m.sum(), "span (max end - min start)", E[m].max() - S[m].min(), "starts rel", (S[m] - S[m].min())[:12], "dur", (E[m]-S[m])[:8])
print("clock64 ticks (100 MHz = 10 ns each), mean over 128 workgroups; n records mean", d[:, :, 7].mean())
for i, nm in enumerate(names):
    print(f"{nm:10s} cum {t[:, :, i].mean():9.1f}  delta {(t[:, :, i] - t[:, :, max(i-1,0)]).mean():9.1f}  (min {(t[:, :, i] - t[:, :, max(i-1,0)]).min()}, max {(t[:, :, i] - t[:, :, max(i-1,0)]).max()})")
v.close()
