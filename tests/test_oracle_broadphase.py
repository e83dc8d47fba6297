"""CPU tests of the broadphase guard's oracle (oracle/broadphase.py) and of the host-side scene builder (tactile_gym_amd/broadphase.py): known
answers for the three stages, the reference's collision filters by name, and the scene the host hands to the device against the oracle's own
reading of the same assets.  The device kernel is compared with this oracle in tests/test_gpu_broadphase.py."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import broadphase as obp     # noqa: E402
from oracle import ref_env               # noqa: E402

EDGE = dict(movement_mode="xy", control_mode="TCP_velocity_control", noise_mode="rand_height", observation_mode="tactile", reward_mode="dense",
            arm_type="ur5", tactile_sensor_name="tactip")


def _box(c, h, R=np.eye(3)):
    return (np.asarray(c, float), np.asarray(R, float), np.asarray(h, float))


def test_obb_separating_axis_known_answers():
    a = _box([0, 0, 0], [1, 1, 1])
    assert obp.obb_overlap(a, _box([1.9, 0, 0], [1, 1, 1]))                      # faces overlap by 0.1
    assert not obp.obb_overlap(a, _box([2.1, 0, 0], [1, 1, 1]))                  # a face axis separates
    c, s = math.cos(math.pi / 4), math.sin(math.pi / 4)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    # a unit cube turned 45 degrees about z, corner towards the first: its corner reaches sqrt(2) from its centre
    assert obp.obb_overlap(a, _box([1 + math.sqrt(2) - 0.01, 0, 0], [1, 1, 1], Rz))
    assert not obp.obb_overlap(a, _box([1 + math.sqrt(2) + 0.01, 0, 0], [1, 1, 1], Rz))
    # two thin crossed sticks, 0.3 apart in z: only the edge-edge (cross product) axis... is the z axis here; AABBs overlap, boxes do not
    assert not obp.obb_overlap(_box([0, 0, 0], [1, 0.1, 0.1]), _box([0, 0, 0.3], [0.1, 1, 0.1]))
    # a genuine edge-edge case: sticks along (1,1,0) and (1,-1,0) directions, tilted, separated only along the cross product of their edges
    Rx = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    assert not obp.obb_overlap(_box([0, 0, 0], [1, 0.05, 0.05], Rz), _box([0, 0.0, 0.25], [1, 0.05, 0.05], Rz.T @ Rx))


def test_sweep_finds_exactly_the_overlapping_pairs_of_different_bodies():
    def item(i, body, static, lo, hi):
        lo, hi = np.asarray(lo, float), np.asarray(hi, float)
        return (i, body, static, lo, hi, ((lo + hi) / 2, np.eye(3), (hi - lo) / 2), None)
    boxes = [item(0, 0, False, [0, 0, 0], [1, 1, 1]), item(1, 0, False, [0.5, 0, 0], [1.5, 1, 1]),        # same body: never a pair
             item(16, 1, True, [0.9, 0.9, 0.9], [2, 2, 2]),                                                # touches both robot boxes
             item(17, 2, True, [0.9, 0.9, 0.9], [3, 3, 3]),                                                # static against static (16): never a pair
             item(19, 4, False, [5, 5, 5], [6, 6, 6]),                                                      # far away
             item(20, 4, False, [1.4, 0.5, 0.5], [1.6, 0.6, 0.6])]                                          # inside box 1 only
    r = obp.sweep(boxes, expected={(1, 20)})
    assert r["aabb_pairs"] == [(0, 16), (0, 17), (1, 16), (1, 17)] and r["pairs"] == 4 and r["hits"] == 4
    assert r["mask"] == (1 << 0) | (1 << 1) | (1 << 16) | (1 << 17)
    assert obp.sweep(boxes, expected=set())["aabb_pairs"] == [(0, 16), (0, 17), (1, 16), (1, 17), (1, 20)]


def test_filters_follow_the_reference_by_link_name():
    """sensors/tactile_sensor.py:46-57 and robots/arms/mg400/mg400.py:68-72."""
    assert obp.filtered_links("ur5", "tactip", "standard", "no_core") == {"tactip_body_link", "tactip_tip_link"}
    assert obp.filtered_links("ur5", "tactip", "right_angle", "fixed") == {"tactip_body_link", "tactip_adapter_link"}
    assert obp.filtered_links("ur5", "digit", "right_angle", "fixed") == {"digit_body_link"}                 # the adapter rule is the TacTip's
    assert obp.filtered_links("mg400", "digitac", "right_angle", "fixed") == {"digitac_body_link", "link4_1", "link4_2", "link5", "tcp_link", "ee_link"}


def test_rest_pose_of_edge_follow_is_clear_and_a_lowered_arm_is_not():
    e = ref_env.OracleEdgeFollowEnv(seed=1, max_steps=50, image_size=(64, 64), env_modes=EDGE)
    e.reset()
    r = obp.check(e)
    assert r["pairs"] == 0 and r["hits"] == 0 and r["mask"] == 0
    q = e.arm.q.copy()
    q[1] -= 0.65; q[2] += 0.73; q[3] += 0.08
    e.arm.reset_joint_states(q)
    names = obp.box_names("ur5", "standard", "tactip")
    hit = {(names[a], names[b]) for a, b in obp.check(e)["hit_pairs"]}
    assert ("forearm_link", "table") in hit and ("forearm_link", "edge stimulus") in hit and ("wrist_3_link", "table") in hit
    assert not any("tactip" in a for a, _ in hit)                                 # body and tip are filtered out (no_core)


def test_stage_three_clears_the_upper_arm_over_the_table():
    """surface_follow's working pose: the UR5's upper arm is a 6 cm cylinder about its joint 9 cm above the table; its BOX reaches the table top
    (stage 2 hit), its convex hull does not (lowest vertex ~3 cm up): no hit.  The arm's world AABB pair with the table stays a stage-1 pair."""
    modes = dict(movement_mode="xyzRxRy", control_mode="TCP_velocity_control", noise_mode="simplex", observation_mode="tactile", reward_mode="dense",
                 arm_type="ur5", tactile_sensor_name="digit")
    e = ref_env.OracleSurfaceFollowAutoEnv(seed=3, max_steps=50, image_size=(64, 64), env_modes=modes)
    e.reset()
    names = obp.box_names("ur5", "standard", "digit")
    ua = next(k for k, v in names.items() if v == "upper_arm_link")
    rng = np.random.default_rng(0)
    for _ in range(40):                                                            # a few random steps: the arm leans over the surface
        boxes, expected = obp.env_boxes(e)
        by = {b[0]: b for b in boxes}
        if obp.obb_overlap(by[ua][5], by[obp.TABLE][5]):
            break
        e.step(rng.uniform(-0.25, 0.25, size=3))
    assert obp.obb_overlap(by[ua][5], by[obp.TABLE][5])                            # the boxes do overlap ...
    assert 0.02 < float(by[ua][6][:, 2].min()) < 0.04                              # ... the hull stays 3 cm up
    r = obp.sweep(boxes, expected)
    assert (ua, obp.TABLE) in r["aabb_pairs"] and (ua, obp.TABLE) not in r["hit_pairs"]


@pytest.mark.parametrize("arm,sensor,typ,core,kw", [("ur5", "tactip", "standard", "no_core", dict(edge="long_edge")),
                                                     ("mg400", "digitac", "right_angle", "fixed", dict(obj="cube")),
                                                     ("ur5", "tactip", "flat", "fixed", dict(obj="sphere")),
                                                     ("ur5", "tactip", "standard", "no_core", dict(obj="round_plate", ball_radius=0.01875))])
def test_host_scene_matches_the_assets_and_the_filters(arm, sensor, typ, core, kw):
    """tactile_gym_amd/broadphase.py builds the tg_broadphase the device gets: every unfiltered robot link a TG_BP_LINK slot with its hull range,
    filtered links empty slots at the same index, table / plane in the world, the expected pairs symmetric."""
    from tactile_gym_amd import _capi as capi
    from tactile_gym_amd.broadphase import BALL, OBJ_A, PLANE, TABLE, Guard
    g = Guard(arm, sensor, typ, core, **kw)
    rb = obp._load(f"{arm}_{typ}_{sensor}")
    off = obp.filtered_links(arm, sensor, typ, core)
    for i, name in enumerate(rb["names"].tolist()):
        b = g.struct.box[i]
        assert b.src == (capi.BP_NONE if name in off else capi.BP_LINK) and b.link == int(rb["link"][i])
        assert b.hull_n == int(rb["hull_off"][i + 1] - rb["hull_off"][i]) and np.allclose(list(b.half), rb["half"][i])
    assert g.struct.box[TABLE].src == capi.BP_WORLD and abs(g.struct.box[TABLE].center[2] + g.struct.box[TABLE].half[2]) < 1e-12    # table top at z = 0
    assert g.struct.box[PLANE].is_static and g.struct.n_hull_verts == len(rb["hull_verts"])
    for a in range(capi.BP_SLOTS):
        for b in range(capi.BP_SLOTS):
            assert ((g.struct.box[a].expected >> b) & 1) == ((g.struct.box[b].expected >> a) & 1)
    if kw.get("obj") == "cube":
        tip = rb["names"].tolist().index(f"{sensor}_tip_link")
        assert g.struct.box[OBJ_A].expected == (1 << TABLE) | (1 << tip)
    if kw.get("ball_radius"):
        from tactile_gym_amd.broadphase import OBJ_B
        assert g.struct.box[BALL].src == capi.BP_BALL and g.struct.box[BALL].expected == (1 << OBJ_A) | (1 << OBJ_B)
        assert g.struct.box[OBJ_A].conj == OBJ_B and g.struct.box[OBJ_B].conj == OBJ_A      # the disc: square + square turned 45 degrees
