"""CPU benchmark of the host half of the tile download (libtg_host.so): 1024 envs x 128 x 128, each env a contact patch of ~5 tiles that
drifts from frame to frame, four ring buffers as in host_tiles.TileDownload.  No GPU needed (messages are made with the torch restatement)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactile_gym_amd.host_tiles import HostTileBatch
from tactile_gym_amd.parallel import TILE_REC, torch_pack_tiles

n, H, W = 1024, 128, 128
rng = np.random.default_rng(0)
tmpl = np.zeros((H, W), np.uint8); tmpl[:20] = 7; tmpl[-20:] = 9; tmpl[:, :20] = 5; tmpl[:, -20:] = 3
cx, cy = rng.uniform(40, 88, n), rng.uniform(40, 88, n)
msgs = []
for f in range(12):
    cx += rng.uniform(-3, 3, n); cy += rng.uniform(-3, 3, n)
    fr = np.tile(tmpl[None], (n, 1, 1))
    for i in range(n):
        x, y = int(cx[i]), int(cy[i])
        fr[i, y - 14:y + 14, x - 18:x + 18] = rng.integers(1, 255, size=(28, 36))
    m = torch.zeros(16 + TILE_REC * n * 64, dtype=torch.uint8)
    c = torch_pack_tiles(torch, torch.from_numpy(fr), torch.from_numpy(tmpl.reshape(-1).copy()), m)
    msgs.append((m.numpy()[:16 + TILE_REC * c].copy(), fr))
print("records per frame:", [int(m[:4].view(np.int32)[0]) for m, _ in msgs][:4], "bytes", msgs[0][0].nbytes)
ring = [HostTileBatch(tmpl.reshape(-1), n, H, W) for _ in range(4)]
for k, (m, fr) in enumerate(msgs):           # correctness + warm-up
    hb = ring[k % 4]; hb.apply(m); assert np.array_equal(hb.batch, fr)
best = 1e9
for rep in range(5):
    t = time.perf_counter()
    for k in range(48):
        ring[k % 4].apply(msgs[k % 12][0])
    best = min(best, (time.perf_counter() - t) / 48)
print(f"host rebuild: {1e3 * best:.3f} ms per frame")
