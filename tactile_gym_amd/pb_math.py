"""Vectorised (leading batch axis) numpy versions of the PyBullet frame helpers the reference's observation code chains:
getQuaternionFromEuler, getEulerFromQuaternion, getMatrixFromQuaternion, multiplyTransforms, invertTransform
(tactile_gym/robots/arms/base_robot_arm.py:46-118; PARITY_ASSUMPTIONS A1).  Quaternions are (x, y, z, w).  Host side only:
used for the `oracle` observation vectors, which the device tactile path does not need."""
import numpy as np


def quat_from_euler(rpy):
    rpy = np.asarray(rpy, dtype=np.float64)
    hr, hp, hy = 0.5 * rpy[..., 0], 0.5 * rpy[..., 1], 0.5 * rpy[..., 2]
    sr, cr, sp, cp, sy, cy = np.sin(hr), np.cos(hr), np.sin(hp), np.cos(hp), np.sin(hy), np.cos(hy)
    q = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], axis=-1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def euler_from_quat(q):
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    sqx, sqy, sqz, sqw = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    roll = np.arctan2(2.0 * (y * z + w * x), sqw - sqx - sqy + sqz)
    pitch = np.arcsin(np.clip(sarg, -1.0, 1.0))
    yaw = np.arctan2(2.0 * (x * y + w * z), sqw + sqx - sqy - sqz)
    lo, hi = sarg <= -0.99999, sarg >= 0.99999          # gimbal branches of btMatrix3x3 / pybullet.c
    roll = np.where(lo | hi, 0.0, roll)
    pitch = np.where(lo, -0.5 * np.pi, np.where(hi, 0.5 * np.pi, pitch))
    yaw = np.where(lo, 2.0 * np.arctan2(x, -y), np.where(hi, 2.0 * np.arctan2(-x, y), yaw))
    return np.stack([roll, pitch, yaw], axis=-1)


def mat_from_quat(q):
    q = np.asarray(q, dtype=np.float64)
    d = np.sum(q * q, axis=-1)
    s = 2.0 / d
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    R = np.stack([1.0 - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0 - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0 - (xx + yy)], axis=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def quat_from_mat(R):
    """btMatrix3x3::getRotation, batched."""
    R = np.asarray(R, dtype=np.float64)
    flat = R.reshape(-1, 3, 3)
    out = np.zeros((flat.shape[0], 4))
    for n, m in enumerate(flat):
        tr = m[0, 0] + m[1, 1] + m[2, 2]
        if tr > 0.0:
            s = np.sqrt(tr + 1.0)
            w = 0.5 * s
            s = 0.5 / s
            out[n] = [(m[2, 1] - m[1, 2]) * s, (m[0, 2] - m[2, 0]) * s, (m[1, 0] - m[0, 1]) * s, w]
        else:
            i = (2 if m[1, 1] < m[2, 2] else 1) if m[0, 0] < m[1, 1] else (2 if m[0, 0] < m[2, 2] else 0)
            j, k = (i + 1) % 3, (i + 2) % 3
            s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
            t = [0.0, 0.0, 0.0, 0.0]
            t[i] = 0.5 * s
            s = 0.5 / s
            t[3] = (m[k, j] - m[j, k]) * s
            t[j] = (m[j, i] + m[i, j]) * s
            t[k] = (m[k, i] + m[i, k]) * s
            out[n] = t
    return out.reshape(R.shape[:-2] + (4,))


def quat_mul(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def invert_transform(p, q):
    q = np.asarray(q, dtype=np.float64)
    qi = q * np.array([-1.0, -1.0, -1.0, 1.0])
    return -np.einsum("...ij,...j->...i", mat_from_quat(qi), np.asarray(p, dtype=np.float64)), qi


def multiply_transforms(pa, qa, pb, qb):
    return np.asarray(pa, dtype=np.float64) + np.einsum("...ij,...j->...i", mat_from_quat(qa), np.asarray(pb, dtype=np.float64)), quat_mul(qa, qb)


class WorkFrame:
    """worldframe_to_workframe / worldvel_to_workvel of BaseRobotArm for a batch of poses."""

    def __init__(self, pos, rpy):
        self.pos, self.rpy = np.asarray(pos, dtype=np.float64), np.asarray(rpy, dtype=np.float64)
        self.orn = quat_from_euler(self.rpy)
        self.inv_pos, self.inv_orn = invert_transform(self.pos, self.orn)
        self.Rinv = mat_from_quat(self.inv_orn)

    def pose(self, pos, rpy):
        p, q = multiply_transforms(self.inv_pos, self.inv_orn, pos, quat_from_euler(rpy))
        return p, euler_from_quat(q)

    def vec(self, v):
        return np.einsum("ij,...j->...i", self.Rinv, np.asarray(v, dtype=np.float64))
