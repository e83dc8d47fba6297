"""URDF -> compact articulated-body model ("TGModel") compiler.

The reference hands a URDF to PyBullet (`robots/arms/robot.py:95-112`, `loadURDF(..., useFixedBase=True)`)
and every later call works on PyBullet's internal multibody.  The MI355X path has no URDF parser on the
device; instead this module flattens the URDF once, on the host, into a handful of small arrays that are
uploaded as constants:

* a tree of *moving* links (one per non-fixed joint) with their joint origin transform and axis,
* a list of rigid *bodies* (one per URDF link, fixed links included) each rigidly attached to a moving
  link (or to the static base), carrying mass, centre of mass and principal inertia,
* named frames (TCP, sensor body, ...) in PyBullet's *inertial* (COM) link frame convention, because the
  reference reads `getLinkState(...)[0:2]` (`base_robot_arm.py:146-147`, `tactile_sensor.py:153-155`).

Bullet semantics that are assumed rather than verified are listed in PARITY_ASSUMPTIONS.md (A1..).
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# PyBullet's URDF importer gives every convex collision child this margin (gUrdfDefaultCollisionMargin)
# [Bullet-knowledge, PARITY_ASSUMPTIONS A3].
URDF_COLLISION_MARGIN = 0.001


# ----------------------------------------------------------------------------- math helpers
def rpy_to_mat(rpy):
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix  R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ],
        dtype=np.float64,
    )


def _floats(text, n=3, default=0.0):
    """Parse a URDF numeric attribute.

    Some reference URDFs hold malformed tokens such as ``4.96E-09+0.035``
    (`ur5_with_standard_digit.urdf:279`).  A C ``strtod``-style scan stops at the first character that cannot
    continue the number, i.e. reads ``4.96E-09`` [PARITY_ASSUMPTIONS A9]; we do the same.
    """
    if text is None:
        return [default] * n
    out = []
    for tok in text.split():
        out.append(_strtod(tok))
    while len(out) < n:
        out.append(default)
    return out[:n]


def _strtod(tok):
    best = None
    for end in range(len(tok), 0, -1):
        try:
            best = float(tok[:end])
            break
        except ValueError:
            continue
    if best is None:
        raise ValueError(f"cannot parse number from {tok!r}")
    return best


# ----------------------------------------------------------------------------- mesh loading
def load_mesh(path):
    """Return (vertices f64[V,3], triangles i32[T,3]) for OBJ / binary or ASCII STL."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return _load_obj(path)
    if ext == ".stl":
        return _load_stl(path)
    raise ValueError(f"unsupported mesh type: {path}")


def _load_obj(path):
    verts, tris = [], []
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):  # polygon fan
                    tris.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(tris, dtype=np.int32).reshape(-1, 3)


def _load_stl(path):
    with open(path, "rb") as fh:
        data = fh.read()
    ntri = struct.unpack_from("<I", data, 80)[0] if len(data) >= 84 else -1
    if ntri >= 0 and 84 + 50 * ntri == len(data):
        rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", 9), ("a", "<u2")]), count=ntri, offset=84)
        verts = rec["v"].reshape(-1, 3).astype(np.float64)
    else:  # ASCII
        verts = []
        for line in data.decode("ascii", errors="replace").splitlines():
            s = line.strip()
            if s.startswith("vertex"):
                verts.append([float(x) for x in s.split()[1:4]])
        verts = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    tris = np.arange(verts.shape[0], dtype=np.int32).reshape(-1, 3)
    return verts, tris


def find_mesh_file(urdf_dir, filename):
    """Mesh lookup relative to the URDF directory, then its parents (PyBullet's UrdfFindMeshFile walks
    '.', '..', '../..' — the UR5 URDFs rely on this: `collision/base.stl` lives one level up)."""
    if filename.startswith("package://"):
        filename = filename[len("package://"):]
    for up in (".", "..", "../..", "../../.."):
        cand = os.path.normpath(os.path.join(urdf_dir, up, filename))
        if os.path.isfile(cand):
            return cand
    return None


# ----------------------------------------------------------------------------- URDF parsing
@dataclass
class _Geom:
    kind: str
    origin_xyz: list
    origin_rpy: list
    size: list = field(default_factory=list)  # box: xyz, sphere: [r], cylinder: [r, l]
    mesh: str | None = None
    scale: list = field(default_factory=lambda: [1.0, 1.0, 1.0])


@dataclass
class _Link:
    name: str
    mass: float
    com_xyz: list
    com_rpy: list
    inertia: list  # ixx ixy ixz iyy iyz izz
    has_inertial: bool
    visuals: list
    collisions: list


@dataclass
class _Joint:
    name: str
    jtype: str
    parent: str
    child: str
    xyz: list
    rpy: list
    axis: list


def _parse_geoms(link_el, tag):
    out = []
    for el in link_el.findall(tag):
        org = el.find("origin")
        xyz = _floats(org.get("xyz") if org is not None else None)
        rpy = _floats(org.get("rpy") if org is not None else None)
        g = el.find("geometry")
        if g is None:
            continue
        if g.find("mesh") is not None:
            m = g.find("mesh")
            out.append(_Geom("mesh", xyz, rpy, mesh=m.get("filename"), scale=_floats(m.get("scale"), 3, 1.0)))
        elif g.find("box") is not None:
            out.append(_Geom("box", xyz, rpy, size=_floats(g.find("box").get("size"))))
        elif g.find("sphere") is not None:
            out.append(_Geom("sphere", xyz, rpy, size=[float(g.find("sphere").get("radius"))]))
        elif g.find("cylinder") is not None:
            c = g.find("cylinder")
            out.append(_Geom("cylinder", xyz, rpy, size=[float(c.get("radius")), float(c.get("length"))]))
    return out


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links, joints = {}, []
    for el in root.findall("link"):
        ine = el.find("inertial")
        if ine is not None:
            org = ine.find("origin")
            mass = float(ine.find("mass").get("value")) if ine.find("mass") is not None else 0.0
            i_el = ine.find("inertia")
            inertia = [float(i_el.get(k, 0.0)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")] if i_el is not None else [0.0] * 6
            link = _Link(
                el.get("name"), mass,
                _floats(org.get("xyz") if org is not None else None),
                _floats(org.get("rpy") if org is not None else None),
                inertia, True, _parse_geoms(el, "visual"), _parse_geoms(el, "collision"),
            )
        else:
            # Bullet's UrdfParser: a link without <inertial> gets mass 1 / inertia 1 unless it is named "world"
            # [Bullet-knowledge, A2]; irrelevant for fixed-base roots.
            is_world = el.get("name") == "world"
            link = _Link(el.get("name"), 0.0 if is_world else 1.0, [0, 0, 0], [0, 0, 0],
                         [0.0] * 6 if is_world else [1, 0, 0, 1, 0, 1], False,
                         _parse_geoms(el, "visual"), _parse_geoms(el, "collision"))
        links[link.name] = link
    for el in root.findall("joint"):
        org = el.find("origin")
        ax = el.find("axis")
        joints.append(_Joint(
            el.get("name"), el.get("type"), el.find("parent").get("link"), el.find("child").get("link"),
            _floats(org.get("xyz") if org is not None else None),
            _floats(org.get("rpy") if org is not None else None),
            _floats(ax.get("xyz")) if ax is not None else [1.0, 0.0, 0.0],
        ))
    return links, joints


# ----------------------------------------------------------------------------- model
@dataclass
class TGModel:
    """Flattened articulated body (fixed base).  All arrays float64 unless noted."""

    name: str
    ndof: int
    parent: np.ndarray        # i32[ndof]   parent moving-link index, -1 = static base
    joint_pos: np.ndarray     # [ndof,3]    joint origin in parent moving-link frame
    joint_rot: np.ndarray     # [ndof,3,3]  joint orientation in parent moving-link frame
    joint_axis: np.ndarray    # [ndof,3]    unit axis in the joint (child link) frame
    joint_names: list
    # rigid bodies (every URDF link with mass or inertia)
    body_link: np.ndarray     # i32[nb]     moving link it is welded to (-1 = base)
    body_com: np.ndarray      # [nb,3]      COM in that link's frame
    body_rot: np.ndarray      # [nb,3,3]    inertial-frame orientation in that link's frame
    body_mass: np.ndarray     # [nb]
    body_inertia: np.ndarray  # [nb,3]      principal inertia (diagonal in body_rot frame)
    body_names: list
    # named frames in PyBullet's inertial-frame convention: (link, pos[3], rot[3,3]) relative to moving link
    frames: dict
    # URDF-order link list (PyBullet link index i <-> joints[i].child), for rest-pose vectors
    urdf_joint_names: list
    urdf_joint_types: list
    urdf_link_names: list
    dof_of_urdf_joint: np.ndarray  # i32[n_urdf_joints], -1 for fixed
    inertia_mode: str = "collision_aabb"

    def to_npz_dict(self):
        d = dict(
            name=np.array(self.name), ndof=np.array(self.ndof), parent=self.parent, joint_pos=self.joint_pos,
            joint_rot=self.joint_rot, joint_axis=self.joint_axis, joint_names=np.array(self.joint_names),
            body_link=self.body_link, body_com=self.body_com, body_rot=self.body_rot, body_mass=self.body_mass,
            body_inertia=self.body_inertia, body_names=np.array(self.body_names),
            urdf_joint_names=np.array(self.urdf_joint_names), urdf_joint_types=np.array(self.urdf_joint_types),
            urdf_link_names=np.array(self.urdf_link_names), dof_of_urdf_joint=self.dof_of_urdf_joint,
            inertia_mode=np.array(self.inertia_mode),
            frame_names=np.array(list(self.frames.keys())),
            frame_link=np.array([v[0] for v in self.frames.values()], dtype=np.int32),
            frame_pos=np.array([v[1] for v in self.frames.values()], dtype=np.float64).reshape(-1, 3),
            frame_rot=np.array([v[2] for v in self.frames.values()], dtype=np.float64).reshape(-1, 3, 3),
        )
        return d

    @staticmethod
    def from_npz(z):
        frames = {}
        for i, n in enumerate(z["frame_names"].tolist()):
            frames[str(n)] = (int(z["frame_link"][i]), z["frame_pos"][i].copy(), z["frame_rot"][i].copy())
        return TGModel(
            name=str(z["name"]), ndof=int(z["ndof"]), parent=z["parent"].astype(np.int32), joint_pos=z["joint_pos"],
            joint_rot=z["joint_rot"], joint_axis=z["joint_axis"], joint_names=[str(s) for s in z["joint_names"].tolist()],
            body_link=z["body_link"].astype(np.int32), body_com=z["body_com"], body_rot=z["body_rot"],
            body_mass=z["body_mass"], body_inertia=z["body_inertia"], body_names=[str(s) for s in z["body_names"].tolist()],
            frames=frames, urdf_joint_names=[str(s) for s in z["urdf_joint_names"].tolist()],
            urdf_joint_types=[str(s) for s in z["urdf_joint_types"].tolist()],
            urdf_link_names=[str(s) for s in z["urdf_link_names"].tolist()],
            dof_of_urdf_joint=z["dof_of_urdf_joint"].astype(np.int32), inertia_mode=str(z["inertia_mode"]),
        )


def _geom_aabb_in(frame_R, frame_p, geom, urdf_dir, missing_mesh_aabb):
    """AABB (min,max) of one collision geometry expressed in the inertial frame (frame_R, frame_p are the
    inertial frame in link coordinates).  Margin handling follows Bullet: convex hulls / boxes add the
    collision margin to their AABB [A3]."""
    Rg = rpy_to_mat(geom.origin_rpy)
    pg = np.asarray(geom.origin_xyz, dtype=np.float64)
    # geometry frame expressed in inertial frame
    R = frame_R.T @ Rg
    p = frame_R.T @ (pg - frame_p)
    if geom.kind == "mesh":
        path = find_mesh_file(urdf_dir, geom.mesh)
        if path is None:
            key = os.path.basename(geom.mesh)
            if missing_mesh_aabb is None or key not in missing_mesh_aabb:
                raise FileNotFoundError(f"collision mesh {geom.mesh} not found and no stand-in AABB given")
            lo, hi = (np.asarray(v, dtype=np.float64) for v in missing_mesh_aabb[key])
            corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
            pts = corners
        else:
            v, _ = load_mesh(path)
            pts = v * np.asarray(geom.scale, dtype=np.float64)
        w = pts @ R.T + p
        return w.min(0) - URDF_COLLISION_MARGIN, w.max(0) + URDF_COLLISION_MARGIN
    if geom.kind == "box":
        h = 0.5 * np.asarray(geom.size, dtype=np.float64)
        ext = np.abs(R) @ h  # btBoxShape::getAabb (half extents already include margin)
        return p - ext, p + ext
    if geom.kind == "sphere":
        r = geom.size[0]
        return p - r, p + r
    if geom.kind == "cylinder":
        r, l = geom.size
        h = np.array([r, r, 0.5 * l])
        ext = np.abs(R) @ h
        return p - ext, p + ext
    raise ValueError(geom.kind)


def compile_urdf(path, frames_of_interest=(), inertia_mode="collision_aabb", missing_mesh_aabb=None, name=None):
    """Flatten `path` into a TGModel.

    inertia_mode:
      "collision_aabb" — PyBullet's default when `loadURDF` is called without URDF_USE_INERTIA_FROM_FILE (the
          reference passes no flags, `robot.py:108-110`): inertia = box inertia of the link's compound collision
          AABB in the inertial frame [Bullet-knowledge, A3];
      "urdf" — take <inertia> as written (diagonal entries; off-diagonals must be zero for the reference assets).
    """
    links, joints = parse_urdf(path)
    urdf_dir = os.path.dirname(os.path.abspath(path))
    children = {j.child for j in joints}
    roots = [n for n in links if n not in children]
    assert len(roots) == 1, f"expected a single root link, got {roots}"
    root = roots[0]
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)

    # PyBullet numbers links depth-first in URDF child order; joint i's child is link i.
    order = []

    def dfs(link):
        for j in by_parent.get(link, []):
            order.append(j)
            dfs(j.child)

    dfs(root)

    parent, jpos, jrot, jaxis, jnames = [], [], [], [], []
    body_link, body_com, body_rot, body_mass, body_inertia, body_names = [], [], [], [], [], []
    frames = {}
    dof_of_joint = []
    # per URDF link: (moving link index, R, p) of the URDF link frame in that moving link's frame
    link_attach = {root: (-1, np.eye(3), np.zeros(3))}

    def add_body(link_name):
        L = links[link_name]
        mi, R, p = link_attach[link_name]
        Rc = rpy_to_mat(L.com_rpy)
        pc = np.asarray(L.com_xyz, dtype=np.float64)
        R_in = R @ Rc          # inertial frame in moving-link frame
        p_in = R @ pc + p
        if link_name in frames_of_interest:
            frames[link_name] = (mi, p_in.copy(), R_in.copy())
        if inertia_mode == "urdf":
            ixx, ixy, ixz, iyy, iyz, izz = L.inertia
            if max(abs(ixy), abs(ixz), abs(iyz)) > 1e-12:
                Im = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
                w, V = np.linalg.eigh(Im)
                if np.linalg.det(V) < 0:
                    V[:, 0] = -V[:, 0]
                R_in = R_in @ V
                diag = w
            else:
                diag = np.array([ixx, iyy, izz])
        else:
            lo = hi = None
            for g in L.collisions:
                a, b = _geom_aabb_in(Rc, pc, g, urdf_dir, missing_mesh_aabb)
                lo = a if lo is None else np.minimum(lo, a)
                hi = b if hi is None else np.maximum(hi, b)
            if lo is None:
                diag = np.zeros(3)
            else:
                l = hi - lo
                diag = L.mass / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2])
        if mi >= 0 and L.mass > 0.0:
            body_link.append(mi)
            body_com.append(p_in)
            body_rot.append(R_in)
            body_mass.append(L.mass)
            body_inertia.append(diag)
            body_names.append(link_name)

    add_body(root)
    for j in order:
        mi, R, p = link_attach[j.parent]
        Rj = rpy_to_mat(j.rpy)
        pj = np.asarray(j.xyz, dtype=np.float64)
        if j.jtype == "fixed":
            link_attach[j.child] = (mi, R @ Rj, R @ pj + p)
            dof_of_joint.append(-1)
        elif j.jtype in ("revolute", "continuous"):
            idx = len(parent)
            parent.append(mi)
            jpos.append(R @ pj + p)
            jrot.append(R @ Rj)
            ax = np.asarray(j.axis, dtype=np.float64)
            jaxis.append(ax / np.linalg.norm(ax))
            jnames.append(j.name)
            link_attach[j.child] = (idx, np.eye(3), np.zeros(3))
            dof_of_joint.append(idx)
        else:
            raise NotImplementedError(f"joint type {j.jtype} ({j.name})")
        add_body(j.child)

    nd = len(parent)
    return TGModel(
        name=name or os.path.splitext(os.path.basename(path))[0], ndof=nd,
        parent=np.asarray(parent, dtype=np.int32), joint_pos=np.asarray(jpos).reshape(nd, 3),
        joint_rot=np.asarray(jrot).reshape(nd, 3, 3), joint_axis=np.asarray(jaxis).reshape(nd, 3), joint_names=jnames,
        body_link=np.asarray(body_link, dtype=np.int32), body_com=np.asarray(body_com).reshape(-1, 3),
        body_rot=np.asarray(body_rot).reshape(-1, 3, 3), body_mass=np.asarray(body_mass, dtype=np.float64),
        body_inertia=np.asarray(body_inertia).reshape(-1, 3), body_names=body_names, frames=frames,
        urdf_joint_names=[j.name for j in order], urdf_joint_types=[j.jtype for j in order],
        urdf_link_names=[j.child for j in order], dof_of_urdf_joint=np.asarray(dof_of_joint, dtype=np.int32),
        inertia_mode=inertia_mode,
    )


def visual_meshes_of_link(path, link_name):
    """Triangles of a link's <visual> geometry in the link's *inertial* frame (for fixture pinning renders)."""
    links, _ = parse_urdf(path)
    L = links[link_name]
    urdf_dir = os.path.dirname(os.path.abspath(path))
    Rc, pc = rpy_to_mat(L.com_rpy), np.asarray(L.com_xyz, dtype=np.float64)
    out_v, out_t, base = [], [], 0
    for g in L.visuals:
        if g.kind != "mesh":
            continue
        mp = find_mesh_file(urdf_dir, g.mesh)
        if mp is None:
            continue
        v, t = load_mesh(mp)
        v = v * np.asarray(g.scale, dtype=np.float64)
        Rg, pg = rpy_to_mat(g.origin_rpy), np.asarray(g.origin_xyz, dtype=np.float64)
        v = v @ Rg.T + pg            # link frame
        v = (v - pc) @ Rc            # inertial frame
        out_v.append(v)
        out_t.append(t + base)
        base += v.shape[0]
    if not out_v:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int32)
    return np.concatenate(out_v), np.concatenate(out_t)


# ----------------------------------------------------------------------------- visual scene (RGB scene camera)
def _primitive_mesh(g):
    """Triangles of a URDF primitive <visual> in its own frame: box 12, cylinder (axis z) 32 segments, sphere: a 16 x 32 lat-long grid.
    (PyBullet tessellates primitives in its renderer; the tessellation here is this repo's own, PARITY_ASSUMPTIONS A32.)"""
    if g.kind == "box":
        return _box_mesh(g.size)
    if g.kind == "cylinder":
        r, l, n = float(g.size[0]), float(g.size[1]), 32
        ang = 2.0 * math.pi * np.arange(n) / n
        ring = np.stack([r * np.cos(ang), r * np.sin(ang)], 1)
        v = np.concatenate([np.c_[ring, np.full(n, -l / 2)], np.c_[ring, np.full(n, l / 2)], [[0, 0, -l / 2], [0, 0, l / 2]]])
        t = []
        for i in range(n):
            j = (i + 1) % n
            t += [(i, j, n + j), (i, n + j, n + i), (2 * n, j, i), (2 * n + 1, n + i, n + j)]
        return v, np.asarray(t, dtype=np.int32)
    if g.kind == "sphere":
        r, nl, nm = float(g.size[0]), 16, 32
        v = [[0.0, 0.0, r]]
        for a in range(1, nl):
            th = math.pi * a / nl
            for b in range(nm):
                ph = 2.0 * math.pi * b / nm
                v.append([r * math.sin(th) * math.cos(ph), r * math.sin(th) * math.sin(ph), r * math.cos(th)])
        v.append([0.0, 0.0, -r])
        t = []
        for b in range(nm):
            t.append((0, 1 + b, 1 + (b + 1) % nm))
            t.append((len(v) - 1, 1 + (nl - 2) * nm + (b + 1) % nm, 1 + (nl - 2) * nm + b))
        for a in range(nl - 2):
            for b in range(nm):
                p00, p01 = 1 + a * nm + b, 1 + a * nm + (b + 1) % nm
                p10, p11 = p00 + nm, p01 + nm
                t += [(p00, p10, p11), (p00, p11, p01)]
        return np.asarray(v), np.asarray(t, dtype=np.int32)
    raise ValueError(g.kind)


def visual_instances(path, default_rgba=(1.0, 1.0, 1.0, 1.0)):
    """What a scene camera sees of a URDF (getCameraImage draws every <visual>): one record per opaque <visual>,
    {"mesh": absolute mesh path or None, "prim": (kind, size) or None, "scale", "link": moving-link index (-1 = welded to the base),
     "R", "p": pose of the visual's geometry frame in that moving link's frame, "rgb": u8[3] from the <material> rgba (inline, or by
    name; `default_rgba` without one)}.  Transparent visuals (alpha < 1: TCP markers, goal indicators) are skipped [PARITY_ASSUMPTIONS A33]."""
    root_el = ET.parse(path).getroot()
    named = {}
    for m in root_el.iter("material"):
        c = m.find("color")
        if c is not None and m.get("name"):
            named.setdefault(m.get("name"), _floats(c.get("rgba"), 4, 1.0))
    links, joints = parse_urdf(path)
    urdf_dir = os.path.dirname(os.path.abspath(path))
    rgba_of = {}                                   # (link name, visual index) -> rgba
    for el in root_el.findall("link"):
        vi = 0
        for v in el.findall("visual"):
            if v.find("geometry") is None:
                continue
            m = v.find("material")
            rgba = None
            if m is not None:
                c = m.find("color")
                rgba = _floats(c.get("rgba"), 4, 1.0) if c is not None else named.get(m.get("name"))
            rgba_of[(el.get("name"), vi)] = rgba if rgba is not None else list(default_rgba)
            vi += 1
    children = {j.child for j in joints}
    root = [n for n in links if n not in children][0]
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)
    order = []

    def dfs(link):
        for j in by_parent.get(link, []):
            order.append(j)
            dfs(j.child)
    dfs(root)
    attach = {root: (-1, np.eye(3), np.zeros(3))}  # same numbering as compile_urdf: moving links in depth-first URDF child order
    n_moving = 0
    for j in order:
        mi, R, p = attach[j.parent]
        Rj, pj = rpy_to_mat(j.rpy), np.asarray(j.xyz, dtype=np.float64)
        if j.jtype == "fixed":
            attach[j.child] = (mi, R @ Rj, R @ pj + p)
        else:
            attach[j.child] = (n_moving, np.eye(3), np.zeros(3))
            n_moving += 1
    out = []
    for name, L in links.items():
        mi, R, p = attach[name]
        for vi, g in enumerate(L.visuals):
            rgba = rgba_of.get((name, vi), list(default_rgba))
            if rgba[3] < 1.0:
                continue
            rec = {"mesh": None, "prim": None, "scale": [1.0, 1.0, 1.0], "link": int(mi)}
            if g.kind == "mesh":
                mp = find_mesh_file(urdf_dir, g.mesh)
                if mp is None:
                    continue                       # a large blob missing upstream
                rec["mesh"], rec["scale"] = mp, [float(x) for x in g.scale]
            else:
                rec["prim"] = (g.kind, [float(x) for x in np.atleast_1d(g.size)])
            Rg, pg = rpy_to_mat(g.origin_rpy), np.asarray(g.origin_xyz, dtype=np.float64)
            rec["R"], rec["p"] = R @ Rg, R @ pg + p
            rec["rgb"] = [int(x) for x in np.clip(np.round(np.asarray(rgba[:3]) * 255.0), 0, 255)]
            out.append(rec)
    return out


def instance_mesh(rec):
    """(verts, tris) of one visual_instances record in its geometry frame, scale applied."""
    if rec["mesh"] is not None:
        v, t = load_mesh(rec["mesh"])
        return np.asarray(v, dtype=np.float64) * np.asarray(rec["scale"], dtype=np.float64), np.asarray(t, dtype=np.int32)
    kind, size = rec["prim"]
    return _primitive_mesh(_Geom(kind=kind, origin_xyz=[0.0] * 3, origin_rpy=[0.0] * 3, size=list(size)))


def visual_scene(path, default_rgba=(1.0, 1.0, 1.0, 1.0)):
    """visual_instances composed: (verts f32 [nv, 3] in the owning moving link's frame, tris i32 [nt, 3], tri_link i8 [nt], tri_rgb u8 [nt, 3])."""
    vs, ts, ls, cs, base = [], [], [], [], 0
    for rec in visual_instances(path, default_rgba):
        v, t = instance_mesh(rec)
        v = v @ np.asarray(rec["R"]).T + np.asarray(rec["p"])
        vs.append(v); ts.append(t + base); base += len(v)
        ls.append(np.full(len(t), rec["link"], dtype=np.int8)); cs.append(np.tile(np.asarray(rec["rgb"], dtype=np.uint8), (len(t), 1)))
    return (np.concatenate(vs).astype(np.float32), np.concatenate(ts).astype(np.int32), np.concatenate(ls), np.concatenate(cs))


# ----------------------------------------------------------------------------- free objects (pole, cube, ...)
def _box_mesh(size):
    hx, hy, hz = (0.5 * float(s) for s in size)
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    # vertex index = 4*ix + 2*iy + iz
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    t = []
    for a, b, c, d in quads:
        t += [(a, b, c), (a, c, d)]
    return v, np.asarray(t, dtype=np.int32)


def compile_free_body(path, inertia_mode="collision_aabb"):
    """Flatten a URDF whose links are all welded together (fixed joints only) into one rigid body.

    Returns a dict in the *inertial frame of the root link* — the frame PyBullet's get/resetBasePositionAndOrientation
    report: mass, com [3], inertia [3,3] about com (root inertial axes), verts float32 [V,3] / tris int32 [T,3] of the
    visual geometry (boxes and meshes), link_masses.  Used for the object_balance pole and the object_push cube."""
    links, joints = parse_urdf(path)
    assert all(j.jtype == "fixed" for j in joints), "compile_free_body: only welded objects are supported"
    urdf_dir = os.path.dirname(os.path.abspath(path))
    children = {j.child for j in joints}
    root = [n for n in links if n not in children][0]
    # pose of every link frame in the root link frame
    pose = {root: (np.eye(3), np.zeros(3))}
    pending = list(joints)
    while pending:
        for j in list(pending):
            if j.parent in pose:
                R, p = pose[j.parent]
                pose[j.child] = (R @ rpy_to_mat(j.rpy), R @ np.asarray(j.xyz, dtype=np.float64) + p)
                pending.remove(j)
    Lr = links[root]
    R0, p0 = rpy_to_mat(Lr.com_rpy), np.asarray(Lr.com_xyz, dtype=np.float64)   # root inertial frame in root link frame

    def to_root_inertial(R, p):   # link-frame pose (R, p in root link frame) -> root inertial frame
        return R0.T @ R, R0.T @ (p - p0)

    parts, verts, tris, base = [], [], [], 0
    for name, L in links.items():
        Rl, pl = pose[name]
        Rc, pc = rpy_to_mat(L.com_rpy), np.asarray(L.com_xyz, dtype=np.float64)
        Ri, pi = to_root_inertial(Rl @ Rc, Rl @ pc + pl)                        # this link's inertial frame
        if inertia_mode == "urdf":
            ixx, ixy, ixz, iyy, iyz, izz = L.inertia
            I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        else:
            lo = hi = None
            for g in L.collisions:
                a, b = _geom_aabb_in(Rc, pc, g, urdf_dir, None)
                lo = a if lo is None else np.minimum(lo, a)
                hi = b if hi is None else np.maximum(hi, b)
            l = (hi - lo) if lo is not None else np.zeros(3)
            I = np.diag(L.mass / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2]))
        if L.mass > 0:
            parts.append((L.mass, pi, Ri @ I @ Ri.T))
        for g in L.visuals:
            Rg, pg = rpy_to_mat(g.origin_rpy), np.asarray(g.origin_xyz, dtype=np.float64)
            if g.kind == "box":
                v, t = _box_mesh(g.size)
            elif g.kind == "mesh" and find_mesh_file(urdf_dir, g.mesh):
                v, t = load_mesh(find_mesh_file(urdf_dir, g.mesh))
                v = v * np.asarray(g.scale, dtype=np.float64)
            elif g.kind in ("cylinder", "sphere"):           # primitives: this repo's tessellation (A32); the round plate of object_balance
                v, t = _primitive_mesh(g)
            else:
                continue
            Rv, pv = to_root_inertial(Rl @ Rg, Rl @ pg + pl)
            verts.append(v @ Rv.T + pv)
            tris.append(t + base)
            base += v.shape[0]
    mass = sum(m for m, _, _ in parts)
    com = sum(m * p for m, p, _ in parts) / mass
    inertia = np.zeros((3, 3))
    for m, p, I in parts:
        d = p - com
        inertia += I + m * ((d @ d) * np.eye(3) - np.outer(d, d))
    return dict(mass=np.array(mass), com=com, inertia=inertia, verts=np.concatenate(verts).astype(np.float32),
                tris=np.concatenate(tris).astype(np.int32), link_masses=np.array([m for m, _, _ in parts]),
                root_inertial_pos=p0, root_inertial_rot=R0)


def collision_hull_of_link(path, link_name):
    """Convex-hull vertices of a link's <collision> meshes, expressed in the frame of the *moving* link the URDF link is
    welded to (PyBullet turns URDF collision meshes into convex hulls unless flagged concave).  Returns
    (moving_link_index, float64 [V,3])."""
    from scipy.spatial import ConvexHull
    links, joints = parse_urdf(path)
    urdf_dir = os.path.dirname(os.path.abspath(path))
    children = {j.child for j in joints}
    root = [n for n in links if n not in children][0]
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)
    attach = {root: (-1, np.eye(3), np.zeros(3))}
    counter = [0]

    def dfs(link):
        for j in by_parent.get(link, []):
            mi, R, p = attach[j.parent]
            Rj, pj = rpy_to_mat(j.rpy), np.asarray(j.xyz, dtype=np.float64)
            if j.jtype == "fixed":
                attach[j.child] = (mi, R @ Rj, R @ pj + p)
            else:
                attach[j.child] = (counter[0], np.eye(3), np.zeros(3))
                counter[0] += 1
            dfs(j.child)

    dfs(root)
    mi, R, p = attach[link_name]
    pts = []
    for g in links[link_name].collisions:
        if g.kind != "mesh":
            continue
        v, _ = load_mesh(find_mesh_file(urdf_dir, g.mesh))
        v = v * np.asarray(g.scale, dtype=np.float64)
        v = v @ rpy_to_mat(g.origin_rpy).T + np.asarray(g.origin_xyz, dtype=np.float64)   # URDF link frame
        pts.append(v @ R.T + p)                                                              # moving-link frame
    pts = np.concatenate(pts)
    hull = ConvexHull(pts)
    return mi, pts[hull.vertices]


def collision_cylinder_of_link(path, link_name):
    """A link's <collision><cylinder> (axis = its local z) expressed in the frame of the *moving* link the URDF link is welded to:
    (moving_link_index, rot float64 [3,3], pos float64 [3], radius, length).  Used for the flat TacTip's tip (object_roll)."""
    links, joints = parse_urdf(path)
    children = {j.child for j in joints}
    root = [n for n in links if n not in children][0]
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)
    attach = {root: (-1, np.eye(3), np.zeros(3))}
    counter = [0]

    def dfs(link):
        for j in by_parent.get(link, []):
            mi, R, p = attach[j.parent]
            Rj, pj = rpy_to_mat(j.rpy), np.asarray(j.xyz, dtype=np.float64)
            if j.jtype == "fixed":
                attach[j.child] = (mi, R @ Rj, R @ pj + p)
            else:
                attach[j.child] = (counter[0], np.eye(3), np.zeros(3))
                counter[0] += 1
            dfs(j.child)

    dfs(root)
    mi, R, p = attach[link_name]
    g = [c for c in links[link_name].collisions if c.kind == "cylinder"][0]
    Rg, pg = rpy_to_mat(g.origin_rpy), np.asarray(g.origin_xyz, dtype=np.float64)
    return mi, R @ Rg, R @ pg + p, float(g.size[0]), float(g.size[1])


# ---------------------------------------------------------------------------------------------------- broadphase guard boxes
def collision_boxes_of_urdf(path, missing_mesh_aabb=None):
    """Every URDF link's <collision> geometry as ONE oriented box per link, for the broadphase guard: the axis-aligned box,
    in the URDF link's own frame, of all the link's collision geometries (margins as Bullet's getAabb adds them, `_geom_aabb_in`), carried
    into the frame of the MOVING link the URDF link is welded to (-1: the fixed base).  Returns a list of dicts
    {name, link, center [3], rot [3,3], half [3]}, in URDF link order (the root first), links without collision geometry left out.
    A robot (revolute / fixed joints) or a welded object (then every box has link -1 and the frame is the root LINK frame)."""
    links, joints = parse_urdf(path)
    urdf_dir = os.path.dirname(os.path.abspath(path))
    children = {j.child for j in joints}
    root = [n for n in links if n not in children][0]
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)
    attach = {root: (-1, np.eye(3), np.zeros(3))}
    order, n_moving = [root], 0

    def dfs(link):
        nonlocal n_moving
        for j in by_parent.get(link, []):
            mi, R, p = attach[j.parent]
            Rj, pj = rpy_to_mat(j.rpy), np.asarray(j.xyz, dtype=np.float64)
            if j.jtype == "fixed":
                attach[j.child] = (mi, R @ Rj, R @ pj + p)
            elif j.jtype in ("revolute", "continuous"):
                attach[j.child] = (n_moving, np.eye(3), np.zeros(3))
                n_moving += 1
            else:
                raise NotImplementedError(j.jtype)
            order.append(j.child)
            dfs(j.child)

    dfs(root)
    out = []
    for name in order:
        L = links[name]
        lo = hi = None
        for g in L.collisions:
            a, b = _geom_aabb_in(np.eye(3), np.zeros(3), g, urdf_dir, missing_mesh_aabb)
            lo = a if lo is None else np.minimum(lo, a)
            hi = b if hi is None else np.maximum(hi, b)
        if lo is None:
            continue
        mi, R, p = attach[name]
        out.append(dict(name=name, link=mi, center=R @ (0.5 * (lo + hi)) + p, rot=R.copy(), half=0.5 * (hi - lo)))
    return out
