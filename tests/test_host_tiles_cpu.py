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


@pytest.mark.parametrize("threads", [1, 2, 4, 7])
def test_threaded_rebuild_equals_the_single_thread_rebuild(threads):
    """tg_host_unpack_tiles_mt on a pool (thread t owns an image range) leaves the same bytes as one thread, frame after frame."""
    from tactile_gym_amd.host_tiles import HostPool
    rng = np.random.default_rng(7)
    n, H, W = 37, 64, 64
    tmpl = _template(H, W, rng)
    pool = HostPool(threads)
    assert pool.threads == threads
    a, b = HostTileBatch(tmpl, n, H, W), HostTileBatch(tmpl, n, H, W, pool=pool.handle)
    cap = 16 + TILE_REC * n * (H // 16) * (W // 16)
    for p in (20, 60, 0, 5, 120, 3):
        f = _frame(tmpl, n, H, W, rng, p)
        msg = torch.zeros(cap, dtype=torch.uint8)
        count = torch_pack_tiles(torch, torch.from_numpy(f.copy()), torch.from_numpy(tmpl.copy()), msg)
        m = msg.numpy()[:16 + TILE_REC * count]
        assert a.apply(m) == count and b.apply(m) == count
        assert np.array_equal(a.batch, f) and np.array_equal(b.batch, f)
        assert a.n_prev.value == b.n_prev.value == count and np.array_equal(a.prev[:count], b.prev[:count])
    pool.close()


def test_a_refused_message_leaves_the_buffer_and_its_tile_list_untouched():
    """validation comes before any write - after a bad tile id (in the middle of the message) the batch, the live-tile list and its
    count are exactly what they were, and the next good message still rebuilds the exact frame (no ghost tiles)."""
    rng = np.random.default_rng(3)
    n, H, W = 4, 64, 64
    tmpl = _template(H, W, rng)
    hb = HostTileBatch(tmpl, n, H, W)
    cap = 16 + TILE_REC * n * 16
    f1, f2 = _frame(tmpl, n, H, W, rng, 6), _frame(tmpl, n, H, W, rng, 6)
    m1 = torch.zeros(cap, dtype=torch.uint8); c1 = torch_pack_tiles(torch, torch.from_numpy(f1.copy()), torch.from_numpy(tmpl.copy()), m1)
    m2 = torch.zeros(cap, dtype=torch.uint8); c2 = torch_pack_tiles(torch, torch.from_numpy(f2.copy()), torch.from_numpy(tmpl.copy()), m2)
    assert c1 > 3 and c2 > 3
    hb.apply(m1.numpy()[:16 + TILE_REC * c1])
    before, prev_before, n_before = hb.batch.copy(), hb.prev.copy(), hb.n_prev.value
    bad = m2.numpy()[:16 + TILE_REC * c2].copy()
    k = c2 // 2
    bad[16 + TILE_REC * k:16 + TILE_REC * k + 4] = np.array([n * 16 + 5], np.int32).view(np.uint8)      # a record in the middle names a tile that does not exist
    with pytest.raises(RuntimeError, match="out of range"):
        hb.apply(bad)
    assert np.array_equal(hb.batch, before) and np.array_equal(hb.prev, prev_before) and hb.n_prev.value == n_before
    hb.apply(m2.numpy()[:16 + TILE_REC * c2])
    assert np.array_equal(hb.batch, f2)


def test_restore_ahead_of_time_then_scatter_equals_the_one_call_rebuild():
    """The ring's next buffer gets its previous frame's tiles restored on the pool's workers before its message exists (restore_begin); the
    rebuild that follows only scatters.  Same bytes as the one-call rebuild, also when the restore is skipped (no workers)."""
    from tactile_gym_amd.host_tiles import HostPool
    rng = np.random.default_rng(11)
    n, H, W = 9, 64, 64
    tmpl = _template(H, W, rng)
    cap = 16 + TILE_REC * n * 16
    for threads in (1, 3):
        pool = HostPool(threads)
        a, b = HostTileBatch(tmpl, n, H, W), HostTileBatch(tmpl, n, H, W, pool=pool.handle)
        for p in (8, 30, 0, 12):
            f = _frame(tmpl, n, H, W, rng, p)
            msg = torch.zeros(cap, dtype=torch.uint8)
            count = torch_pack_tiles(torch, torch.from_numpy(f.copy()), torch.from_numpy(tmpl.copy()), msg)
            m = msg.numpy()[:16 + TILE_REC * count]
            b.restore_begin()
            a.apply(m); b.apply(m)
            assert np.array_equal(a.batch, f) and np.array_equal(b.batch, f)
        b.restore_begin(); b.restore_end()
        assert np.array_equal(b.batch, np.tile(tmpl.reshape(1, H, W), (n, 1, 1))) and b.n_prev.value == 0
        pool.close()
