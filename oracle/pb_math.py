"""Restatement of the PyBullet transform helpers the reference calls (TEST INFRASTRUCTURE).

Quaternions are [x, y, z, w] like PyBullet.  Formulas follow pybullet.c's getQuaternionFromEuler /
getEulerFromQuaternion and Bullet's btTransform algebra [Bullet-knowledge, PARITY_ASSUMPTIONS A1].
Call sites in the reference: base_robot_arm.py:39-118 (work-frame transforms), tactile_sensor.py:184-210.
"""
import math

import numpy as np


def quat_from_euler(rpy):
    phi, the, psi = (0.5 * float(v) for v in rpy)
    q = np.array([
        math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi),
        math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi),
        math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi),
        math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi),
    ])
    return q / math.sqrt(float(q @ q))


def euler_from_quat(q):
    x, y, z, w = (float(v) for v in q)
    sqx, sqy, sqz, sqw = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return np.array([0.0, -0.5 * math.pi, 2.0 * math.atan2(x, -y)])
    if sarg >= 0.99999:
        return np.array([0.0, 0.5 * math.pi, 2.0 * math.atan2(-x, y)])
    return np.array([
        math.atan2(2.0 * (y * z + w * x), sqw - sqx - sqy + sqz),
        math.asin(sarg),
        math.atan2(2.0 * (x * y + w * z), sqw + sqx - sqy - sqz),
    ])


def mat_from_quat(q):
    x, y, z, w = (float(v) for v in q)
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1.0 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1.0 - (xx + yy)]])


def quat_from_mat(R):
    """btMatrix3x3::getRotation."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if tr > 0.0:
        s = math.sqrt(tr + 1.0)
        q[3] = 0.5 * s
        s = 0.5 / s
        q[0] = (R[2, 1] - R[1, 2]) * s
        q[1] = (R[0, 2] - R[2, 0]) * s
        q[2] = (R[1, 0] - R[0, 1]) * s
    else:
        i = 0 if R[0, 0] >= R[1, 1] else 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    return q


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def multiply_transforms(pa, qa, pb, qb):
    pos = np.asarray(pa, dtype=np.float64) + mat_from_quat(qa) @ np.asarray(pb, dtype=np.float64)
    return pos, quat_mul(np.asarray(qa, dtype=np.float64), np.asarray(qb, dtype=np.float64))


def invert_transform(p, q):
    qi = np.array([-q[0], -q[1], -q[2], q[3]], dtype=np.float64)
    return -(mat_from_quat(qi) @ np.asarray(p, dtype=np.float64)), qi
