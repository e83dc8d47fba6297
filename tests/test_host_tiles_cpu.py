"""libtg_host.so (tactile_gym_amd/host/tg_host_tiles.c): the host half of the tile-sparse observation download, against the torch
restatement of the tile payload (parallel.torch_pack_tiles / torch_unpack_tiles - the definition the device kernels are tested against)."""
import numpy as np
import pytest
import torch

from tactile_gym_amd.host_tiles import HostTileBatch, host_lib
from tactile_gym_amd.parallel import TILE_REC, torch_pack_tiles, torch_unpack_tiles


def _template(H, W, rng):
    t = np.zeros((H, W), np.uint8)
    t[:5] = rng.integers(1, 255, size=(5, W)); t[:, -7:] = rng.integers(1, 255, size=(H, 7))      # a "pasted ring" on two sides
    return t.reshape(-1)


def _frame(tmpl, n, H, W, rng, patches):
    f = np.tile(tmpl.reshape(1, H, W), (n, 1, 1))
    for _ in range(patches):
        i, y, x, h, w = rng.integers(0, n), rng.integers(0, H - 8), rng.integers(0, W - 8), rng.integers(1, 40), rng.integers(1, 40)
        f[i, y:y + h, x:x + w] = rng.integers(0, 255, size=f[i, y:y + h, x:x + w].shape)
    return f


@pytest.mark.parametrize("n,H,W", [(5, 128, 128), (3, 64, 256), (1, 16, 16)])
def test_host_batch_follows_a_sequence_of_tile_messages(n, H, W):
    """Frame after frame (contact patches appear, move, vanish; a frame with no live tile; a frame where every tile is live) the persistent
    host batch equals the frame, and equals what the torch restatement rebuilds from the same message."""
    rng = np.random.default_rng(n * 1000 + H)
    tmpl = _template(H, W, rng)
    hb = HostTileBatch(tmpl, n, H, W)
    assert np.array_equal(hb.batch, np.tile(tmpl.reshape(1, H, W), (n, 1, 1)))
    cap = 16 + TILE_REC * n * (H // 16) * (W // 16)
    frames = [_frame(tmpl, n, H, W, rng, p) for p in (3, 6, 0, 2, 9)] + [rng.integers(0, 255, size=(n, H, W)).astype(np.uint8) ^ 1,
                                                                        _frame(tmpl, n, H, W, rng, 1)]
    for f in frames:
        msg = torch.zeros(cap, dtype=torch.uint8)
        count = torch_pack_tiles(torch, torch.from_numpy(f.copy()), torch.from_numpy(tmpl.copy()), msg)
        got = hb.apply(msg.numpy()[:16 + TILE_REC * count])
        assert got == count
        assert np.array_equal(hb.batch, f)
        ref = torch.zeros(n, H * W, dtype=torch.uint8)
        torch_unpack_tiles(torch, msg, torch.from_numpy(tmpl.copy()), n, H, W, ref)
        assert np.array_equal(hb.batch.reshape(n, -1), ref.numpy())


def test_host_unpack_refuses_what_it_cannot_trust():
    rng = np.random.default_rng(1)
    n, H, W = 2, 32, 32
    tmpl = _template(H, W, rng)
    hb = HostTileBatch(tmpl, n, H, W)
    f = _frame(tmpl, n, H, W, rng, 2)
    msg = torch.zeros(16 + TILE_REC * n * 4, dtype=torch.uint8)
    count = torch_pack_tiles(torch, torch.from_numpy(f), torch.from_numpy(tmpl.copy()), msg)
    good = msg.numpy()[:16 + TILE_REC * count].copy()
    assert count > 0
    bad = good.copy(); bad[12] ^= 0xFF                                   # magic
    with pytest.raises(RuntimeError, match="header"):
        hb.apply(bad)
    with pytest.raises(RuntimeError, match="shorter"):
        hb.apply(good[:-8])
    bad = good.copy(); bad[16:20] = np.array([n * 4], np.int32).view(np.uint8)    # first record's tile id one past the end
    with pytest.raises(RuntimeError, match="out of range"):
        HostTileBatch(tmpl, n, H, W).apply(bad)
    bad = good.copy(); bad[4:8] = np.array([n + 1], np.int32).view(np.uint8)     # another batch size
    with pytest.raises(RuntimeError, match="header"):
        hb.apply(bad)
    assert HostTileBatch(tmpl, n, H, W).apply(good) == count                     # and the good one still goes through
    assert {"tg_host_unpack_tiles", "tg_host_fill_template"} <= {s for s in ("tg_host_unpack_tiles", "tg_host_fill_template") if hasattr(host_lib(), s)}
