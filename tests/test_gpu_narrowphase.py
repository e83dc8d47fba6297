"""The general narrowphase of tg_config.narrowphase (row n2: "narrowphase GJK/EPA on the meshes", stepSimulation at robots/arms/robot.py:141; the
tip core - cube pair of object_push_env.py:216-225): the wave-mapped GJK / EPA (csrc/tg_narrowphase.hpp) against the oracle's C restatement
(oracle/narrowphase.c) on identical inputs - BIT-EXACT: signed distance, normal and both witness points (same operation order, no FMA
contraction on either side); the oracle against the independent numpy / scipy GJK / EPA is tests/test_oracle_known_answers.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def placements(hull, half, n_cases, seed):
    """The hull in random attitudes around the box: over faces, edges and corners, from 4 mm inside the surface to 8 mm outside."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_cases, hull.shape[0], 3))
    c0 = hull.mean(0)
    for t in range(n_cases):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, np.pi)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        p = rng.uniform(-1, 1, size=3)
        p = p / np.abs(p).max()
        out[t] = (hull - c0) @ R.T + p * (half + 0.012 + rng.uniform(-0.004, 0.008))
    return np.ascontiguousarray(out)


@pytest.mark.parametrize("robot", ["mg400_right_angle_digitac", "ur5_right_angle_tactip"])
def test_device_gjk_epa_equals_oracle_bit_for_bit(robot):
    from oracle import minibullet as mb
    from tactile_gym_amd import _capi
    hull = np.ascontiguousarray(np.load(os.path.join(ROOT, "tactile_gym_amd", "assets", "robots", robot + ".npz"))["tip_hull_verts"], dtype=np.float64)
    half = np.array([0.04, 0.04, 0.04])
    n = 1500
    cases = placements(hull, half, n, seed=3)
    dp = C.POINTER(C.c_double)
    dev = np.zeros((n, 11))
    assert 0 == (_capi.test_lib().tg_selftest_narrowphase(n, hull.shape[0], cases.ctypes.data_as(dp), half.ctypes.data_as(dp), dev.ctypes.data_as(dp)))
    L = mb.lib()
    ref = np.zeros((n, 11))
    for t in range(n):
        sd = C.c_double(); nn = (C.c_double * 3)(); pa = (C.c_double * 3)(); pb = (C.c_double * 3)()
        ok = L.mb_gjk_epa_hull_box(cases[t].ctypes.data_as(dp), hull.shape[0], half.ctypes.data_as(dp), C.byref(sd), nn, pa, pb)
        ref[t] = [ok, sd.value] + list(nn) + list(pa) + list(pb) if ok else [0] + [0.0] * 10
    assert (ref[:, 0] == 1).all() and (dev[:, 0] == 1).all()
    sep, pen = int((ref[:, 1] > 0).sum()), int((ref[:, 1] < 0).sum())
    assert sep > 200 and pen > 200, (sep, pen)            # both branches (GJK distance, EPA depth) are exercised
    same = dev.view(np.uint64) == ref.view(np.uint64)
    worst = np.abs(dev - ref).max()
    assert same.all(), (int((~same).any(axis=1).sum()), worst)
    print(f"{robot}: {n} placements ({sep} separated, {pen} overlapping): distance, normal, witness points bit-identical to the oracle")
