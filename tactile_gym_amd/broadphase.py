"""Host side of the broadphase guard (include/tactile_gym_hip.h: tg_set_broadphase; device: csrc/tg_broadphase.hip).

The reference never lists its contact pairs: it loads URDFs, switches some links' collisions off and lets Bullet's broadphase find whatever
overlaps (robots/arms/robot.py:141).  This library's solver has rows for a FIXED set of pairs per env; `build_guard` assembles the scene the
device checks that set against - one oriented box per URDF link with collision geometry (assets/collision/*.npz, written by
tools/extract_assets.py: collision_boxes) minus the links the reference filters out:

    sensors/tactile_sensor.py:46-57      the sensor body always; the TacTip adapter of the right_angle / mini_right_angle / forward mountings; the tip
                                         when the env's t_s_core is "no_core" (edge_follow_env.py:64, object_balance_env.py:54; "fixed" in
                                         base_surface_env.py:65, object_push_env.py:60, object_roll_env.py:56)
    robots/arms/mg400/mg400.py:68-72     link4_1, link4_2, link5, tcp_link, ee_link
    base_surface_env.py:432              the heightfield (no box at all)
    base_object_env.py:75, object_push_env.py:249   goal and trajectory indicators (no boxes)

Slots (tg_bp_box index): 0-15 the robot's boxes in URDF link order (a filtered link keeps its index, with src = TG_BP_NONE), 16 table, 17 plane
(base_tactile_env.py:131-139), 18 the edge stimulus (edge_follow_env.py:218-235), 19 / 20 the free object's boxes, 21 the ball of ball_on_plate.
Expected pairs (the solver has rows for them): object_push cube - table, cube - tip; object_roll marble - table, marble - tip; ball_on_plate
ball - plate."""
import ctypes as C
import os

import numpy as np

from . import _capi as capi

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "collision")
MARGIN = 0.0015          # per box: the distance a box travels inside one env step (<= 1 mm at the envs' action ranges) + contactBreakingThreshold 1e-4
HULL_MARGIN = 0.001      # gUrdfDefaultCollisionMargin
TABLE, PLANE, STIM, OBJ_A, OBJ_B, BALL = 16, 17, 18, 19, 20, 21
BODY_ROBOT, BODY_TABLE, BODY_PLANE, BODY_STIM, BODY_OBJ, BODY_BALL = range(6)


def _load(name):
    z = np.load(os.path.join(ASSETS, name + ".npz"))
    return {k: z[k] for k in z.files}


def filtered_links(arm_type, t_s_name, t_s_type, t_s_core):
    off = {f"{t_s_name}_body_link"}
    if t_s_name == "tactip" and t_s_type in ("right_angle", "mini_right_angle", "forward"):
        off.add("tactip_adapter_link")
    if t_s_core == "no_core":
        off.add(f"{t_s_name}_tip_link")
    if arm_type == "mg400":
        off |= {"link4_1", "link4_2", "link5", "tcp_link", "ee_link"}
    return off


class Guard:
    """tg_broadphase plus the arrays behind its pointer and the slot names (for messages)."""

    def __init__(self, arm_type, t_s_name, t_s_type, t_s_core, edge=None, obj=None, ball_radius=None, every_step=False):
        g = capi.TgBroadphase()
        self.names = {}
        rb = _load(f"{arm_type}_{t_s_type}_{t_s_name}")
        if len(rb["names"]) > 16:
            raise ValueError(f"{len(rb['names'])} collision links: the guard has 16 robot slots")
        off = filtered_links(arm_type, t_s_name, t_s_type, t_s_core)
        self.hull = np.ascontiguousarray(rb["hull_verts"], dtype=np.float64)
        tip = None

        def put(k, center, rot, half, src, body, static, link=-1, hull=(0, 0)):
            b = g.box[k]
            for i in range(3):
                b.center[i], b.half[i] = float(center[i]), float(half[i])
            for i in range(9):
                b.rot[i] = float(np.asarray(rot).reshape(9)[i])
            b.src, b.link, b.body, b.is_static, b.hull_off, b.hull_n, b.expected, b.conj = src, link, body, int(static), int(hull[0]), int(hull[1]), 0, -1

        for i, name in enumerate(rb["names"].tolist()):
            self.names[i] = name
            if name == f"{t_s_name}_tip_link":
                tip = i
            l = int(rb["link"][i])
            put(i, rb["center"][i], rb["rot"][i], rb["half"][i], capi.BP_NONE if name in off else capi.BP_LINK, BODY_ROBOT, l < 0, link=l,
                hull=(int(rb["hull_off"][i]), int(rb["hull_off"][i + 1] - rb["hull_off"][i])))
        for k, body, name in ((TABLE, BODY_TABLE, "table"), (PLANE, BODY_PLANE, "plane")):
            b = _load(name)
            put(k, b["center"][0] + b["base_pos"], b["rot"][0], b["half"][0], capi.BP_WORLD, body, True)
            self.names[k] = name
        g.sphere_half, g.ball_radius = 1.0, 0.0
        expected = []
        if edge is not None:                                                   # "long_edge" | "short_edge"
            b = _load(edge)
            put(STIM, b["center"][0], b["rot"][0], b["half"][0], capi.BP_EDGE, BODY_STIM, True)
            self.names[STIM] = "edge stimulus"
        if obj in ("cube", "pole", "round_plate"):
            b = _load(obj)
            for k in range(len(b["names"])):
                put(OBJ_A + k, b["center"][k], b["rot"][k], b["half"][k], capi.BP_BODY, BODY_OBJ, False)
                self.names[OBJ_A + k] = f"{obj}:{b['names'][k]}"
            if obj == "cube":
                expected = [(TABLE, OBJ_A), (tip, OBJ_A)]
            if obj == "round_plate":                                           # a disc: its square and the square turned 45 degrees bound it together
                g.box[OBJ_A].conj, g.box[OBJ_B].conj = OBJ_B, OBJ_A
        elif obj == "sphere":                                                  # object_roll's marble
            b = _load("sphere")
            put(OBJ_A, b["center"][0], np.eye(3), b["half"][0], capi.BP_SPHERE, BODY_OBJ, False)
            g.sphere_half = float(b["half"][0][0])
            self.names[OBJ_A] = "marble"
            expected = [(TABLE, OBJ_A), (tip, OBJ_A)]
        if ball_radius is not None:                                            # ball_on_plate
            b = _load("balance_ball")
            put(BALL, b["center"][0], np.eye(3), b["half"][0], capi.BP_BALL, BODY_BALL, False)
            g.sphere_half, g.ball_radius = float(b["half"][0][0]), float(ball_radius)
            self.names[BALL] = "ball"
            expected = [(OBJ_A, BALL), (OBJ_B, BALL)]
        for a, b in expected:
            if a is not None:
                g.box[a].expected |= 1 << b
                g.box[b].expected |= 1 << a
        g.margin, g.hull_margin = MARGIN, HULL_MARGIN
        g.n_hull_verts = int(self.hull.shape[0])
        g.hull_verts = self.hull.ctypes.data_as(C.POINTER(C.c_double))
        g.every_step = int(bool(every_step))
        self.struct = g

    def describe(self, mask):
        return [self.names.get(k, f"slot {k}") for k in range(capi.BP_SLOTS) if (int(mask) >> k) & 1]
