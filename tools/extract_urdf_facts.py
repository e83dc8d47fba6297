#!/usr/bin/env python3
"""Facts of the reference's robot URDFs, read with nothing but xml.etree / re / struct / numpy (NO import of tactile_gym_amd):

    python tools/extract_urdf_facts.py            (in the build container, where /root/reference exists)
    -> tests/golden/urdf_facts.json

Per URDF (every arm x sensor x sensor type the envs load, base_robot_arm.py:17-37 / robot.py:95-112): each link's mass, inertial
origin, written inertia and collision geometry (primitive sizes; for a mesh its file name, scale and the bounding box of its vertices,
read by this script's own OBJ / STL readers), each joint's type, parent, child, origin, axis, limits and dynamics - as written in the
file.  tests/test_urdf_facts.py compares the compiled blobs in tactile_gym_amd/assets/robots/ (tactile_gym_amd/urdf_compile.py,
which the product AND the oracle load) with these facts, so that masses, inertial frames, joint frames and the AABB-derived inertias
are pinned by something that shares no code with the compiler.

Number tokens such as `4.96E-09+0.035` (ur5_with_standard_digit.urdf:279) are read like C strtod (the longest leading number); every
such token is listed under "malformed_tokens".  PARITY_ASSUMPTIONS A9.
"""
import json
import math
import os
import re
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("TG_REFERENCE_ASSETS", "/root/reference/tactile_gym/assets")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "urdf_facts.json")

ROBOTS = [(arm, sensor, typ) for arm in ("ur5", "mg400") for sensor in ("tactip", "digit", "digitac")
          for typ in ("standard", "right_angle", "forward", "mini_right_angle", "flat")]

_NUM = re.compile(r"[+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?")
malformed = []


def num(tok, where):
    m = _NUM.match(tok)
    if m is None:
        raise ValueError(f"{where}: no number in {tok!r}")
    if m.end() != len(tok):
        malformed.append({"where": where, "token": tok, "read_as": m.group(0)})
    return float(m.group(0))


def vec(text, n, default, where):
    if text is None:
        return [default] * n
    out = [num(t, where) for t in text.split()]
    return (out + [default] * n)[:n]


def obj_vertices(path):
    pts = []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if line.startswith("v "):
                pts.append([float(t) for t in line.split()[1:4]])
    return np.asarray(pts, dtype=np.float64).reshape(-1, 3)


def stl_vertices(path):
    data = open(path, "rb").read()
    n_tri = struct.unpack_from("<I", data, 80)[0] if len(data) >= 84 else -1
    if n_tri >= 0 and 84 + 50 * n_tri == len(data):            # binary: 50-byte records of normal + 3 vertices (float32) + attribute
        rec = np.frombuffer(data, dtype=np.uint8, offset=84).reshape(n_tri, 50)[:, 12:48]
        return np.ascontiguousarray(rec).view("<f4").reshape(-1, 3).astype(np.float64)
    return np.asarray([[float(m.group(i)) for i in (1, 2, 3)] for m in re.finditer(rb"vertex\s+(\S+)\s+(\S+)\s+(\S+)", data)], dtype=np.float64).reshape(-1, 3)


def rot(rpy):
    """URDF fixed-axis roll, pitch, yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def find_mesh(urdf_dir, name):
    if name.startswith("package://"):
        name = name[len("package://"):]
    for up in (".", "..", "../..", "../../.."):                 # the UR5 URDFs name `collision/base.stl`, which lives one level up
        cand = os.path.normpath(os.path.join(urdf_dir, up, name))
        if os.path.isfile(cand):
            return cand
    return None


def geoms(link_el, tag, urdf_dir, where, inertial=None):
    """inertial = (xyz, rpy) of the link's inertial frame: each record then also carries the geometry's extent along the axes of that
    frame ("aabb_inertial": exact for a mesh (every vertex), |R| h for a box, cylinder as its bounding box, sphere +- r) - what Bullet
    recomputes the link inertia from when loadURDF is given no URDF_USE_INERTIA_FROM_FILE (PARITY_ASSUMPTIONS A3); no margin added."""
    out = []
    Ri, pi = (rot(inertial[1]), np.asarray(inertial[0])) if inertial is not None else (np.eye(3), np.zeros(3))
    for i, el in enumerate(link_el.findall(tag)):
        org, g = el.find("origin"), el.find("geometry")
        if g is None:
            continue
        rec = {"xyz": vec(org.get("xyz") if org is not None else None, 3, 0.0, where), "rpy": vec(org.get("rpy") if org is not None else None, 3, 0.0, where)}
        if g.find("mesh") is not None:
            m = g.find("mesh")
            rec.update(type="mesh", file=os.path.basename(m.get("filename")), scale=vec(m.get("scale"), 3, 1.0, where))
            path = find_mesh(urdf_dir, m.get("filename"))
            if path is None:
                rec["bounds"] = rec["aabb_inertial"] = None      # a missing large blob upstream (the standard TacTip body)
            else:
                v = (obj_vertices if path.lower().endswith(".obj") else stl_vertices)(path) * np.asarray(rec["scale"])
                rec["bounds"], rec["vertices"] = [v.min(0).tolist(), v.max(0).tolist()], int(v.shape[0])
                w = (v @ rot(rec["rpy"]).T + np.asarray(rec["xyz"]) - pi) @ Ri          # rows: Ri^T (Rg v + pg - pi)
                rec["aabb_inertial"] = [w.min(0).tolist(), w.max(0).tolist()]
        else:
            if g.find("box") is not None:
                rec.update(type="box", size=vec(g.find("box").get("size"), 3, 0.0, where))
                h = 0.5 * np.asarray(rec["size"])
            elif g.find("sphere") is not None:
                rec.update(type="sphere", radius=num(g.find("sphere").get("radius"), where))
                h = None
            elif g.find("cylinder") is not None:
                c = g.find("cylinder")
                rec.update(type="cylinder", radius=num(c.get("radius"), where), length=num(c.get("length"), where))
                h = np.array([rec["radius"], rec["radius"], 0.5 * rec["length"]])
            else:
                continue
            c0 = Ri.T @ (np.asarray(rec["xyz"]) - pi)
            ext = np.full(3, rec["radius"]) if h is None else np.abs(Ri.T @ rot(rec["rpy"])) @ h
            rec["aabb_inertial"] = [(c0 - ext).tolist(), (c0 + ext).tolist()]
        out.append(rec)
    return out


def facts(urdf):
    root = ET.parse(urdf).getroot()
    d = os.path.dirname(os.path.abspath(urdf))
    base = os.path.basename(urdf)
    links, joints = {}, []
    for el in root.findall("link"):
        where = f"{base}:{el.get('name')}"
        ine = el.find("inertial")
        rec = {"has_inertial": ine is not None}
        if ine is not None:
            org, mass, it = ine.find("origin"), ine.find("mass"), ine.find("inertia")
            rec["mass"] = num(mass.get("value"), where) if mass is not None else 0.0
            rec["inertial_xyz"] = vec(org.get("xyz") if org is not None else None, 3, 0.0, where)
            rec["inertial_rpy"] = vec(org.get("rpy") if org is not None else None, 3, 0.0, where)
            rec["inertia"] = [num(it.get(k, "0"), where) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")] if it is not None else [0.0] * 6
        rec["collisions"] = geoms(el, "collision", d, where, (rec["inertial_xyz"], rec["inertial_rpy"]) if ine is not None else None)
        links[el.get("name")] = rec
    for el in root.findall("joint"):
        where = f"{base}:{el.get('name')}"
        org, ax, lim, dyn = el.find("origin"), el.find("axis"), el.find("limit"), el.find("dynamics")
        joints.append({
            "name": el.get("name"), "type": el.get("type"), "parent": el.find("parent").get("link"), "child": el.find("child").get("link"),
            "xyz": vec(org.get("xyz") if org is not None else None, 3, 0.0, where), "rpy": vec(org.get("rpy") if org is not None else None, 3, 0.0, where),
            "axis": vec(ax.get("xyz"), 3, 0.0, where) if ax is not None else None,
            "limit": {k: num(lim.get(k), where) for k in ("lower", "upper", "effort", "velocity") if lim.get(k) is not None} if lim is not None else None,
            "dynamics": {k: num(dyn.get(k), where) for k in ("damping", "friction") if dyn.get(k) is not None} if dyn is not None else None,
        })
    return {"links": links, "joints": joints}


def main():
    out = {"_what": "facts of the reference's robot URDFs as written in the files; made by tools/extract_urdf_facts.py (xml.etree only)", "robots": {}}
    for arm, sensor, typ in ROBOTS:
        urdf = os.path.join(REF, "robot_assets", arm, sensor, f"{arm}_with_{typ}_{sensor}.urdf")
        if not os.path.isfile(urdf):
            continue
        out["robots"][f"{arm}_{typ}_{sensor}"] = dict(facts(urdf), urdf=os.path.relpath(urdf, REF))
    out["malformed_tokens"] = malformed
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"wrote {os.path.relpath(OUT, ROOT)}: {len(out['robots'])} URDFs, {len(malformed)} malformed number tokens, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    sys.exit(main())
