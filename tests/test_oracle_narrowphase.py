"""CPU pins of the oracle's general narrowphase (oracle/narrowphase.c, PARITY A35-A38): the C GJK / EPA against the independent numpy / scipy
implementation (oracle/gjk_epa.py: different simplex solve, scipy's convex hull for the polytope) on general hull / box placements, and the
persistent manifold's cache rules as known answers.  Reference call site: pb.stepSimulation(), robots/arms/robot.py:141."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import gjk_epa as g
from oracle import minibullet as mb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def _c_gjk(hull, half):
    sd = C.c_double(); n = (C.c_double * 3)(); pa = (C.c_double * 3)(); pb = (C.c_double * 3)()
    hull = np.ascontiguousarray(hull)
    ok = mb.lib().mb_gjk_epa_hull_box(hull.ctypes.data_as(dp), hull.shape[0], np.ascontiguousarray(half).ctypes.data_as(dp), C.byref(sd), n, pa, pb)
    return ok, sd.value, np.array(n[:]), np.array(pa[:]), np.array(pb[:])


@pytest.mark.parametrize("robot", ["mg400_right_angle_digitac", "ur5_right_angle_tactip"])
def test_c_gjk_epa_equals_independent_numpy_gjk_epa(robot):
    hull = np.ascontiguousarray(np.load(os.path.join(ROOT, "tactile_gym_amd", "assets", "robots", robot + ".npz"))["tip_hull_verts"], dtype=np.float64)
    half = np.array([0.04, 0.04, 0.04])
    rng = np.random.default_rng(1)
    sep = pen = 0
    for t in range(120):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, np.pi)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        p = rng.uniform(-1, 1, size=3)
        p = p / np.abs(p).max()                               # faces, edges and corners of the box
        hb = (hull - hull.mean(0)) @ R.T + p * (half + 0.012 + rng.uniform(-0.004, 0.004))
        ok, sd, n, pa, pb = _c_gjk(hb, half)
        assert ok
        A, B = g.hull_support(hb), g.box_support([0, 0, 0], np.eye(3), half)
        d, qa, qb, _ = g.gjk(A, B)
        if d > 0:
            ref_d, ref_n = d, (qa - qb) / d
            sep += 1
        else:
            dep, nn = g.epa(A, B)
            ref_d, ref_n = -dep, -nn
            pen += 1
        assert abs(sd - ref_d) < 1e-12 and np.abs(n - ref_n).max() < 1e-8, (t, sd, ref_d)
        assert abs((pa - pb) @ n - sd) < 1e-12                # the witness points realise the distance along the normal
        q = np.abs(pb) - half                                 # the box witness lies on the box, the hull witness inside the hull's bounding box
        assert q.max() < 1e-12 and (hb.min(0) - 1e-12 <= pa).all() and (pa <= hb.max(0) + 1e-12).all()
    assert sep >= 30 and pen >= 15, (sep, pen)


def _mani():
    return mb.MBManifold()


def _add(m, pa, pb, n, depth, breaking=1e-4):
    I = np.eye(3).reshape(9).copy(); o = np.zeros(3)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (o, I, o, I, pa, pb, n)]
    mb.lib().mb_manifold_add(C.byref(m), breaking, *[x.ctypes.data_as(dp) for x in a], C.c_double(depth))


def _setup():
    L = mb.lib()
    L.mb_manifold_add.argtypes = [C.POINTER(mb.MBManifold), C.c_double, dp, dp, dp, dp, dp, dp, dp, C.c_double]
    L.mb_manifold_refresh.argtypes = [C.POINTER(mb.MBManifold), C.c_double, dp, dp, dp, dp]


def test_manifold_add_replace_and_area_rule():
    """A36: a point within the breaking threshold of a cached one replaces it; a fifth point replaces the cached point whose removal keeps the
    largest area, never the deepest; a point farther than the threshold from the surface is not added."""
    _setup()
    m = _mani()
    n = [0, 0, 1.0]
    sq = [(0, 0), (0.01, 0), (0.01, 0.01), (0, 0.01)]
    for k, (x, y) in enumerate(sq):
        _add(m, [x, y, 0], [x, y, 1e-5], n, -1e-5 * (k + 1))
    assert m.n == 4 and [round(m.depth[k], 9) for k in range(4)] == [-1e-5, -2e-5, -3e-5, -4e-5]
    _add(m, [0.01 + 5e-5, 0, 0], [0.01 + 5e-5, 0, 2e-5], n, -7e-5)       # 0.05 mm from cached point 1: replaces it
    assert m.n == 4 and abs(m.depth[1] + 7e-5) < 1e-12 and abs(m.la[1][0] - 0.01005) < 1e-12
    _add(m, [0.5, 0.5, 0], [0.5, 0.5, 0], n, 2e-4)                         # beyond the breaking threshold: ignored
    assert m.n == 4
    _add(m, [0.03, 0.005, 0], [0.03, 0.005, 1e-6], n, -1e-6)               # a fifth, shallow point far to the right
    assert m.n == 4
    xs = sorted(round(m.la[k][0], 5) for k in range(4))
    assert 0.03 in xs                                                      # it went in ...
    assert any(abs(m.depth[k] + 7e-5) < 1e-12 for k in range(4))           # ... and the deepest point stayed


def test_manifold_refresh_breaks_separated_and_drifted_points():
    """A36: after the bodies move, points whose distance along their normal exceeds the threshold, or whose anchors drifted apart sideways by more
    than the threshold, are removed (the last point takes the slot); the others get their refreshed distance."""
    _setup()
    m = _mani()
    n = [0, 0, 1.0]
    for k, x in enumerate([0.0, 0.01, 0.02]):
        _add(m, [x, 0, 0], [x, 0, 0], n, 0.0)
    I = np.eye(3).reshape(9).copy()
    z = np.zeros(3)
    L = mb.lib()

    def refresh(oa):
        args = [np.ascontiguousarray(x, dtype=np.float64) for x in (oa, I, z, I)]
        L.mb_manifold_refresh(C.byref(m), 1e-4, *[x.ctypes.data_as(dp) for x in args])

    refresh([0, 0, 5e-5])                    # body A lifted by half the threshold: all stay, distance 5e-5
    assert m.n == 3 and all(abs(m.depth[k] - 5e-5) < 1e-15 for k in range(3))
    refresh([2e-4, 0, 0])                    # slid sideways by twice the threshold: every point drifted -> gone
    assert m.n == 0
    for k, x in enumerate([0.0, 0.01, 0.02]):
        _add(m, [x, 0, 0], [x, 0, 0], n, 0.0)
    refresh([0, 0, 3e-4])                    # lifted beyond the threshold: gone
    assert m.n == 0


def test_oracle_env_manifold_holds_several_tip_points_and_clears_at_reset():
    from oracle.ref_env import OracleObjectPushEnv
    modes = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="simplex",
                 observation_mode="tactile_and_feature", reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")
    env = OracleObjectPushEnv(seed=3, max_steps=1000, image_size=(64, 64), env_modes=modes, narrowphase="gjk_manifold")
    env.reset()
    assert env.scene.mani.n == 0
    rng = np.random.default_rng(0)
    counts = []
    for _ in range(25):
        env.step(rng.uniform(-0.25, 0.25, 2).astype(np.float32))
        counts.append(env.scene.n_contacts)
        ids = list(env.scene.contact_ids)[:env.scene.n_contacts]
        assert ids[-env.scene.mani.n:] == [8 + k for k in range(env.scene.mani.n)] if env.scene.mani.n else True
    assert max(counts) >= 6 and min(counts) >= 4      # four table contacts, one or two (or more) tip points
    env.reset()
    assert env.scene.mani.n == 0
