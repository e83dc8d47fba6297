#!/usr/bin/env python3
"""Compile the reference's *data* assets into the compact blobs this repo ships.

Run in the build container (where /root/reference exists):   python tools/extract_assets.py

Reads (never copies verbatim) from /root/reference/tactile_gym/assets:
  * robot URDFs + collision meshes  -> tactile_gym_amd/assets/robots/<arm>_<type>_<sensor>.npz   (TGModel arrays)
  * sensor reference images (.npy)  -> tactile_gym_amd/assets/sensors/<sensor>_<type>_<N>.npz     (hot-path constants a15)
  * stimulus meshes                 -> tactile_gym_amd/assets/stimuli/<name>.npz                  (float32 verts, int32 tris)
  * free objects (pole, cube)       -> tactile_gym_amd/assets/objects/<name>.npz                  (mass, com, inertia, visual triangles)
  * every opaque <visual>           -> tactile_gym_amd/assets/visual/{mesh_*.npz, <scene>.json}    (scene camera: unique meshes + instance lists)
  * skin / body visual meshes       -> tests/golden/<sensor>_<type>_view.npz                      (fixture-pinning only)

Nothing under tests/, bench.py or smoke() reads /root/reference at run time; they read these blobs.
Provenance and licences: tactile_gym_amd/assets/PROVENANCE.md.
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tactile_gym_amd.urdf_compile import collision_boxes_of_urdf, collision_cylinder_of_link, collision_hull_of_link, compile_free_body, compile_urdf, instance_mesh, load_mesh, parse_urdf, rpy_to_mat, visual_instances, visual_meshes_of_link  # noqa: E402

REF = os.environ.get("TG_REFERENCE_ASSETS", "/root/reference/tactile_gym/assets")
OUT = os.path.join(ROOT, "tactile_gym_amd", "assets")
GOLD = os.path.join(ROOT, "tests", "golden")

# The standard TacTip body mesh is a missing large blob in the reference checkout (.MISSING_LARGE_BLOBS); its
# collision is filtered out (tactile_sensor.py:51) so it only matters through the AABB-derived inertia of a
# 0.25 kg link.  Stand-in AABB: a 50 mm x 50 mm x 65 mm housing (65 mm = tip joint offset, urdf:329)
# [PARITY_ASSUMPTIONS A3b].
MISSING = {"tactip_body.obj": ([-0.025, -0.025, 0.0], [0.025, 0.025, 0.065])}

ROBOTS = [
    # (arm, sensor, type, frames of interest)
    ("ur5", "tactip", "standard"),
    ("ur5", "digit", "standard"),
    ("ur5", "digitac", "standard"),
    ("mg400", "tactip", "standard"),
    ("mg400", "digit", "standard"),              # edge_follow on the MG400 (edge_follow/rest_poses.py:114-153)
    ("mg400", "digitac", "standard"),
    ("mg400", "digitac", "right_angle"),
    ("mg400", "digit", "right_angle"),
    ("mg400", "tactip", "right_angle"),
    ("mg400", "tactip", "mini_right_angle"),     # object_push on the MG400 with a TacTip (object_push_env.py:70-75)
    ("ur5", "tactip", "right_angle"),            # object_push on the UR5 (object_push_env.py:59)
    ("ur5", "digit", "right_angle"),
    ("ur5", "digitac", "right_angle"),
    ("mg400", "tactip", "forward"),              # surface_follow-v2: vertical surface (base_surface_env.py:60-63)
    ("mg400", "digit", "forward"),
    ("mg400", "digitac", "forward"),
    ("ur5", "tactip", "forward"),
    ("ur5", "digit", "forward"),
    ("ur5", "digitac", "forward"),
    ("ur5", "tactip", "flat"),                   # object_roll (object_roll_env.py:57)
]

SENSOR_IMAGES = [
    ("tactip", "standard", (64, 128, 256)),
    ("tactip", "right_angle", (64, 128, 256)),
    ("tactip", "mini_right_angle", (64, 128, 256)),
    ("tactip", "forward", (64, 128, 256)),
    ("tactip", "flat", (64, 128, 256)),
    ("digit", "forward", (64, 128, 256)),
    ("digitac", "forward", (64, 128, 256)),
    ("digit", "standard", (64, 128, 256)),
    ("digit", "right_angle", (64, 128, 256)),
    ("digitac", "standard", (64, 128, 256)),
    ("digitac", "right_angle", (64, 128, 256)),
]


ONLY_NEW = "--only-new" in sys.argv     # leave blobs that already exist untouched (zip timestamps would churn the history)


def save(path, **arrays):
    if ONLY_NEW and os.path.isfile(path):
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **arrays)
    print(f"wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)} B)")


def robots():
    for arm, sensor, typ in ROBOTS:
        urdf = os.path.join(REF, "robot_assets", arm, sensor, f"{arm}_with_{typ}_{sensor}.urdf")
        if not os.path.isfile(urdf):
            print("skip (missing)", urdf)
            continue
        links, _ = parse_urdf(urdf)
        foi = [n for n in links if n in ("tcp_link", "ee_link", f"{sensor}_body_link", f"{sensor}_tip_link",
                                         "tactip_adapter_link", "link4_1", "link4_2", "link5")]
        for mode in ("collision_aabb", "urdf"):
            try:
                m = compile_urdf(urdf, frames_of_interest=foi, inertia_mode=mode, missing_mesh_aabb=MISSING,
                                 name=f"{arm}_{typ}_{sensor}")
            except FileNotFoundError as e:
                print("skip", arm, sensor, typ, mode, e)
                continue
            suffix = "" if mode == "collision_aabb" else "_urdfinertia"
            d = m.to_npz_dict()
            try:   # collision core of the sensor tip (convex hull, in the frame of the moving link it is welded to)
                hl, hv = collision_hull_of_link(urdf, f"{sensor}_tip_link")
                d.update(tip_hull_link=np.array(hl), tip_hull_verts=hv)
            except Exception as e:  # noqa: BLE001 - e.g. no collision mesh
                print("  no tip hull:", e)
                try:   # the flat TacTip's tip collides as a URDF cylinder
                    cl, cR, cp, cr, clen = collision_cylinder_of_link(urdf, f"{sensor}_tip_link")
                    d.update(tip_cyl_link=np.array(cl), tip_cyl_rot=cR, tip_cyl_pos=cp, tip_cyl_radius=np.array(cr), tip_cyl_length=np.array(clen))
                    print("  tip cylinder: link", cl, "radius", cr, "length", clen)
                except Exception as e2:  # noqa: BLE001
                    print("  no tip cylinder:", e2)
            save(os.path.join(OUT, "robots", f"{arm}_{typ}_{sensor}{suffix}.npz"), **d)


def sensors():
    for sensor, typ, sizes in SENSOR_IMAGES:
        for n in sizes:
            d = os.path.join(REF, "robot_assets", sensor, "reference_images", typ, f"{n}x{n}")
            if not os.path.isdir(d):
                print("skip (missing)", d)
                continue
            save(os.path.join(OUT, "sensors", f"{sensor}_{typ}_{n}.npz"),
                 nodef_dep=np.load(os.path.join(d, "nodef_dep.npy")).astype(np.float32),
                 nodef_gray=np.load(os.path.join(d, "nodef_gray.npy")).astype(np.float32),
                 border_mask=np.load(os.path.join(d, "border_mask.npy")).astype(np.uint8))


def stimuli():
    v, t = load_mesh(os.path.join(REF, "rl_env_assets/exploration/edge_follow/edge_stimuli/long_edge_flat/long_edge.obj"))
    save(os.path.join(OUT, "stimuli", "long_edge.npz"), verts=v.astype(np.float32), tris=t.astype(np.int32))
    # short_edge.urdf re-uses long_edge.obj with a mesh scale; read it from the URDF
    links, _ = parse_urdf(os.path.join(REF, "rl_env_assets/exploration/edge_follow/edge_stimuli/long_edge_flat/short_edge.urdf"))
    g = next(iter(links.values())).visuals[0]
    save(os.path.join(OUT, "stimuli", "short_edge.npz"), verts=(v * np.asarray(g.scale)).astype(np.float32), tris=t.astype(np.int32))


GOLDEN_FAMILIES = [   # one URDF per (sensor, type) family of reference_images/ (the mounting inside the sensor does not depend on the arm)
    ("ur5", "tactip", "standard"), ("ur5", "tactip", "flat"), ("ur5", "tactip", "forward"), ("ur5", "tactip", "right_angle"),
    ("mg400", "tactip", "mini_right_angle"), ("ur5", "digit", "standard"), ("ur5", "digit", "forward"), ("ur5", "digit", "right_angle"),
    ("ur5", "digitac", "standard"), ("ur5", "digitac", "forward"), ("mg400", "digitac", "right_angle"),
]


def rigid_group_meshes(urdf, body):
    """Visual triangles of every URDF link welded (fixed joints) to the moving link that carries `body`, expressed in `body`'s
    *inertial* frame - the frame getLinkState(...)[0:2] reports and the in-sensor camera is mounted in (tactile_sensor.py:153-187)."""
    links, joints = parse_urdf(urdf)
    par = {j.child: j for j in joints}
    root = body
    while root in par and par[root].jtype == "fixed":
        root = par[root].parent
    poses = {root: (np.eye(3), np.zeros(3))}      # link frame of each welded link in the root link's frame
    grew = True
    while grew:
        grew = False
        for j in joints:
            if j.jtype == "fixed" and j.parent in poses and j.child not in poses:
                R, p = poses[j.parent]
                Rj, pj = rpy_to_mat(j.rpy), np.asarray(j.xyz)
                poses[j.child] = (R @ Rj, R @ pj + p)
                grew = True
    Rb, pb = poses[body]
    Lb = links[body]
    Rbi, pbi = rpy_to_mat(Lb.com_rpy), np.asarray(Lb.com_xyz)
    vs, ts, names, base = [], [], [], 0
    for name, (R, p) in poses.items():
        v, t = visual_meshes_of_link(urdf, name)           # that link's inertial frame
        if len(t) == 0:
            continue
        L = links[name]
        v = v @ rpy_to_mat(L.com_rpy).T + np.asarray(L.com_xyz)   # -> link frame
        v = v @ R.T + p                                     # -> root link frame
        v = (v - pb) @ Rb                                   # -> body link frame
        v = (v - pbi) @ Rbi                                 # -> body inertial frame
        vs.append(v); ts.append(t + base); names.append(name); base += len(v)
    return np.concatenate(vs).astype(np.float32), np.concatenate(ts).astype(np.int32), names


def golden_views():
    """tests/golden/<sensor>_<type>_view.npz: what the in-sensor camera sees at rest (skin / gel, body, adapter, flange), in the
    sensor-body inertial frame, for every reference_images family.  Only the triangles that own at least one pixel of the 64x64,
    128x128 or 256x256 render are kept (the rest cannot change those images: the depth test is a min), which keeps the blobs small."""
    from oracle import minibullet as mb
    from oracle.ref_env import sensor_camera
    for arm, sensor, typ in GOLDEN_FAMILIES:
        urdf = os.path.join(REF, "robot_assets", arm, sensor, f"{arm}_with_{typ}_{sensor}.urdf")
        v, t, names = rigid_group_meshes(urdf, f"{sensor}_body_link")
        cam = sensor_camera(sensor, typ)
        M = mb.cam_from_obj_matrix(cam["pos"], rpy_to_mat(cam["rpy"]), np.zeros(3), np.eye(3))
        keep = np.zeros(len(t), bool)
        for n in (64, 128, 256):
            full = np.ones((n, n), np.float32)
            mb.render_depth(v, t, M, cam["fov"], cam["near"], cam["far"], n, n, full)
            one = np.ones((n, n), np.float32)
            for k in range(len(t)):
                if keep[k]:
                    continue
                one.fill(1.0)
                mb.render_depth(v, t[k:k + 1], M, cam["fov"], cam["near"], cam["far"], n, n, one)
                keep[k] = bool(np.any((one < 1.0) & (one == full)))
        tk = t[keep]
        used, inv = np.unique(tk.reshape(-1), return_inverse=True)
        save(os.path.join(GOLD, f"{sensor}_{typ}_view.npz"), verts=v[used], tris=inv.reshape(-1, 3).astype(np.int32),
             links=np.array(",".join(names)), urdf=np.array(f"{arm}_with_{typ}_{sensor}.urdf"))
        print(f"  {sensor}/{typ}: {int(keep.sum())} of {len(t)} triangles own a pixel; links {names}")


def objects():
    """Free objects (welded URDFs) flattened to one rigid body + their visual triangles."""
    for name, rel in (("pole", "rl_env_assets/nonprehensile_manipulation/object_balance/pole/pole.urdf"),
                      ("round_plate", "rl_env_assets/nonprehensile_manipulation/object_balance/round_plate/round_plate.urdf"),
                      ("cube", "rl_env_assets/nonprehensile_manipulation/object_push/cube/cube.urdf")):
        for mode in ("collision_aabb", "urdf"):
            d = compile_free_body(os.path.join(REF, rel), inertia_mode=mode)
            suffix = "" if mode == "collision_aabb" else "_urdfinertia"
            save(os.path.join(OUT, "objects", f"{name}{suffix}.npz"), **d)


def spinning_plate():
    """object_balance, object_mode "spinning_plate" (object_balance_env.py:198-239): the dish (spinning_plate.urdf, the free object) and the spool
    it stands on (plate_buffer.urdf, tied to the TCP).  Both collide as the CONVEX HULL of their mesh (a URDF <mesh> collision without the
    concave flag: btConvexHullShape, optimizeConvexHull): `hull` = the hull's vertices in the link frame, in mesh order."""
    from scipy.spatial import ConvexHull
    d0 = "rl_env_assets/nonprehensile_manipulation/object_balance/spinning_plate"
    for name in ("plate_buffer", "spinning_plate"):
        d = compile_free_body(os.path.join(REF, d0, f"{name}.urdf"), inertia_mode="collision_aabb")
        v = np.asarray(d["verts"], dtype=np.float64)
        keep = np.unique(ConvexHull(v).vertices)
        d["hull"] = v[keep]
        save(os.path.join(OUT, "objects", f"{name}.npz"), **d)
        print(f"  {name}: {len(v)} vertices, {len(keep)} on the hull")


def sphere():
    """object_roll marble: sphere.urdf (mass 0.05, collision sphere r = 0.0025; inertia of a solid sphere, what Bullet computes from the
    collision shape) + the tessellation upstream ships next to it (sphere.obj, r = 0.0025) as the visual [PARITY_ASSUMPTIONS A30]."""
    d = os.path.join(REF, "rl_env_assets/nonprehensile_manipulation/object_roll/sphere")
    links, _ = parse_urdf(os.path.join(d, "sphere.urdf"))
    L = next(iter(links.values()))
    r = float(L.collisions[0].size[0])
    v, t = load_mesh(os.path.join(d, "sphere.obj"))
    save(os.path.join(OUT, "objects", "sphere.npz"), mass=np.array(L.mass), radius=np.array(r), urdf_inertia=np.array(L.inertia),
         verts=v.astype(np.float32), tris=t.astype(np.int32))


def balance_ball():
    """object_balance's ball (object_mode "ball_on_plate"): object_balance/sphere/sphere.urdf, loaded with globalScaling 7.5
    (object_balance_env.py:245-260): mass, collision radius as written; the cylinder of round_plate.urdf it rolls on."""
    d = os.path.join(REF, "rl_env_assets/nonprehensile_manipulation/object_balance")
    links, _ = parse_urdf(os.path.join(d, "sphere", "sphere.urdf"))
    L = next(iter(links.values()))
    plinks, _ = parse_urdf(os.path.join(d, "round_plate", "round_plate.urdf"))
    P = next(iter(plinks.values()))
    cyl = P.collisions[0]
    assert cyl.kind == "cylinder" and list(cyl.origin_xyz) == list(P.com_xyz)        # the collision cylinder is centred on the inertial frame
    save(os.path.join(OUT, "objects", "balance_ball.npz"), mass=np.array(L.mass), radius=np.array(float(L.collisions[0].size[0])),
         urdf_inertia=np.array(L.inertia), plate_radius=np.array(float(cyl.size[0])), plate_length=np.array(float(cyl.size[1])))


def _weld(v, t):
    """Merge bit-identical vertices (STL stores three per triangle) and drop triangles that collapse."""
    v32 = np.ascontiguousarray(v, dtype=np.float32)
    uniq, inv = np.unique(v32.view([("", np.float32)] * 3), return_inverse=True)
    t2 = inv.reshape(-1)[t]
    keep = (t2[:, 0] != t2[:, 1]) & (t2[:, 1] != t2[:, 2]) & (t2[:, 0] != t2[:, 2])
    return uniq.view(np.float32).reshape(-1, 3), t2[keep].astype(np.int32)


def _scene(name, urdf, base_pos=(0.0, 0.0, 0.0)):
    """assets/visual/<name>.json: the instance list of a URDF's opaque visuals (visual_instances); mesh files go once, welded, to
    assets/visual/mesh_<stem>_<sha1[:8]>.npz, primitives stay parametric.  `base_pos` bakes a fixed loadURDF position into link -1."""
    import hashlib
    import json
    out = []
    for rec in visual_instances(urdf):
        e = {"link": rec["link"], "R": np.asarray(rec["R"]).reshape(9).tolist(), "rgb": rec["rgb"], "scale": rec["scale"],
             "p": (np.asarray(rec["p"]) + (np.asarray(base_pos) if rec["link"] < 0 else 0.0)).tolist()}
        if rec["mesh"] is not None:
            h = hashlib.sha1(open(rec["mesh"], "rb").read()).hexdigest()[:8]
            key = f"mesh_{os.path.splitext(os.path.basename(rec['mesh']))[0].lower()}_{h}"
            dst = os.path.join(OUT, "visual", key + ".npz")
            if not os.path.isfile(dst):
                v, t = _weld(*load_mesh(rec["mesh"]))
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                np.savez_compressed(dst, verts=v, tris=t)
            e["mesh"] = key
        else:
            e["prim"] = [rec["prim"][0], rec["prim"][1]]
        out.append(e)
    os.makedirs(os.path.join(OUT, "visual"), exist_ok=True)
    with open(os.path.join(OUT, "visual", name + ".json"), "w") as f:
        json.dump({"source": os.path.relpath(urdf, REF), "instances": out}, f)
    return out


def scenes():
    """What the envs' scene camera (get_visual_obs, base_tactile_env.py:212-245) can see: plane + table (load_environment,
    base_tactile_env.py:131-139, positions baked), each robot URDF, each task object."""
    env = "shared_assets/environment_objects"
    _scene("world_plane", os.path.join(REF, env, "plane/plane.urdf"), (0.0, 0.0, -0.625))
    _scene("world_table", os.path.join(REF, env, "table/table.urdf"), (0.50, 0.0, -0.625))
    for arm, sensor, typ in ROBOTS:
        urdf = os.path.join(REF, "robot_assets", arm, sensor, f"{arm}_with_{typ}_{sensor}.urdf")
        if os.path.isfile(urdf):
            _scene(f"robot_{arm}_{typ}_{sensor}", urdf)
    for name, rel in (("long_edge", "rl_env_assets/exploration/edge_follow/edge_stimuli/long_edge_flat/long_edge.urdf"),
                      ("cube", "rl_env_assets/nonprehensile_manipulation/object_push/cube/cube.urdf"),
                      ("pole", "rl_env_assets/nonprehensile_manipulation/object_balance/pole/pole.urdf"),
                      ("sphere", "rl_env_assets/nonprehensile_manipulation/object_roll/sphere/sphere.urdf")):
        _scene("object_" + name, os.path.join(REF, rel))


def collision_boxes():
    """Broadphase guard boxes (tactile_gym_amd/csrc/tg_broadphase.hip, oracle/broadphase.py): one oriented box per URDF link
    that has <collision> geometry - every robot, the table, the ground plane, the edge stimuli, the free objects - in
    assets/collision/<name>.npz: names, link (moving link the box rides on; -1 = the body's base), center [k,3], rot [k,3,3], half [k,3].
    Objects: in the root link's INERTIAL frame (what get / resetBasePositionAndOrientation and this library's body_pos / body_rot speak)."""
    def write(name, boxes, **extra):
        save(os.path.join(OUT, "collision", f"{name}.npz"), names=np.array([b["name"] for b in boxes]), link=np.array([b["link"] for b in boxes], dtype=np.int32),
             center=np.array([b["center"] for b in boxes]).reshape(-1, 3), rot=np.array([b["rot"] for b in boxes]).reshape(-1, 3, 3),
             half=np.array([b["half"] for b in boxes]).reshape(-1, 3), **extra)
    for arm, sensor, typ in ROBOTS:
        urdf = os.path.join(REF, "robot_assets", arm, sensor, f"{arm}_with_{typ}_{sensor}.urdf")
        if not os.path.isfile(urdf):
            continue
        boxes = collision_boxes_of_urdf(urdf, MISSING)
        # third stage of the guard (robot link against the table top): the convex hull of the link's collision meshes (what Bullet collides: URDF
        # meshes become convex hulls) in the moving link's frame; a link whose collision is a primitive or a missing blob keeps its box's 8 corners
        hulls, off = [], [0]
        for b in boxes:
            try:
                _, hv = collision_hull_of_link(urdf, b["name"])
            except Exception:  # noqa: BLE001 - no mesh (primitive) or a missing large blob
                sgn = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
                hv = b["center"] + (sgn * b["half"]) @ b["rot"].T
            hulls.append(hv)
            off.append(off[-1] + len(hv))
        write(f"{arm}_{typ}_{sensor}", boxes, hull_off=np.array(off, dtype=np.int32), hull_verts=np.concatenate(hulls))
    env_objects = "shared_assets/environment_objects"
    write("table", collision_boxes_of_urdf(os.path.join(REF, env_objects, "table/table.urdf")), base_pos=np.array([0.50, 0.00, -0.625]))   # base_tactile_env.py:135-139
    write("plane", collision_boxes_of_urdf(os.path.join(REF, env_objects, "plane/plane.urdf")), base_pos=np.array([0.0, 0.0, -0.625]))     # :131-134
    edge = "rl_env_assets/exploration/edge_follow/edge_stimuli/long_edge_flat"
    for name in ("long_edge", "short_edge"):
        write(name, collision_boxes_of_urdf(os.path.join(REF, edge, f"{name}.urdf")))
    for name, rel in (("pole", "rl_env_assets/nonprehensile_manipulation/object_balance/pole/pole.urdf"),
                      ("round_plate", "rl_env_assets/nonprehensile_manipulation/object_balance/round_plate/round_plate.urdf"),
                      ("cube", "rl_env_assets/nonprehensile_manipulation/object_push/cube/cube.urdf"),
                      ("balance_ball", "rl_env_assets/nonprehensile_manipulation/object_balance/sphere/sphere.urdf"),
                      ("sphere", "rl_env_assets/nonprehensile_manipulation/object_roll/sphere/sphere.urdf")):
        urdf = os.path.join(REF, rel)
        links, joints = parse_urdf(urdf)
        root = [n for n in links if n not in {j.child for j in joints}][0]
        R0, p0 = rpy_to_mat(links[root].com_rpy), np.asarray(links[root].com_xyz, dtype=np.float64)
        boxes = collision_boxes_of_urdf(urdf)
        for b in boxes:                                   # root link frame -> root inertial frame
            b["center"], b["rot"] = R0.T @ (b["center"] - p0), R0.T @ b["rot"]
        if name == "round_plate":                         # a disc (<cylinder>, axis z): a second box, the first turned 45 degrees about the axis - the
            c45 = math.sqrt(0.5)                          # guard takes a pair with the plate for a hit only if BOTH boxes are reached (an octagon)
            Rz = np.array([[c45, -c45, 0.0], [c45, c45, 0.0], [0.0, 0.0, 1.0]])
            boxes.append(dict(boxes[0], name="round_plate@45", rot=boxes[0]["rot"] @ Rz))
        write(name, boxes)


if __name__ == "__main__":
    if "--scenes-only" in sys.argv:
        scenes()
        sys.exit(0)
    if "--collision-only" in sys.argv:
        collision_boxes()
        sys.exit(0)
    if "--spinning-plate-only" in sys.argv:
        spinning_plate()
        sys.exit(0)
    objects()
    sphere()
    balance_ball()
    spinning_plate()
    robots()
    sensors()
    stimuli()
    golden_views()
    scenes()
    collision_boxes()
