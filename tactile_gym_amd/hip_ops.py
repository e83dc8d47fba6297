"""Function-level entry points of the HIP library (batched, numpy in / numpy out).

Each wraps one device implementation of a PyBullet call on the reference's hot path; the GPU parity tests compare
them with the CPU oracle.  They raise if the HIP library or a GPU is missing — there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _prep(x, n, nd):
    a = np.ascontiguousarray(x, dtype=np.float64).reshape(n, nd)
    return a


def inverse_dynamics(robot, q, qd, qdd, dtype="f64"):
    """calculateInverseDynamics (base_robot_arm.py:176-178) for a batch [n, ndof]."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    n, nd = q.shape
    q, qd, qdd = _prep(q, n, nd), _prep(qd, n, nd), _prep(qdd, n, nd)
    tau = np.zeros((n, nd))
    capi.check(capi.lib().tg_inverse_dynamics(C.byref(robot), capi.PHYSICS[dtype], n, _dp(q), _dp(qd), _dp(qdd), _dp(tau)))
    return tau


def mass_matrix(robot, q, dtype="f64"):
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    n, nd = q.shape
    q = _prep(q, n, nd)
    M = np.zeros((n, nd, nd))
    capi.check(capi.lib().tg_mass_matrix(C.byref(robot), capi.PHYSICS[dtype], n, _dp(q), _dp(M)))
    return M


def jacobian_tcp(robot, q, dtype="f64"):
    """calculateJacobian at the TCP frame (base_robot_arm.py:300-307): (J [n,6,ndof], pos [n,3], rot [n,3,3])."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    n, nd = q.shape
    q = _prep(q, n, nd)
    J, pos, rot = np.zeros((n, 6, nd)), np.zeros((n, 3)), np.zeros((n, 3, 3))
    capi.check(capi.lib().tg_jacobian_tcp(C.byref(robot), capi.PHYSICS[dtype], n, _dp(q), _dp(J), _dp(pos), _dp(rot)))
    return J, pos, rot


def sim_ticks(robot, q, qd, n_ticks, motor_mode, q_des=None, qd_des=None, max_force=1000.0, dt=1.0 / 240.0, iters=150, dtype="f64"):
    """n_ticks x (gravity compensation + stepSimulation) (robot.py:131-141); returns (q, qd)."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    n, nd = q.shape
    q, qd = _prep(q, n, nd).copy(), _prep(qd, n, nd).copy()
    qdes = _prep(q_des, n, nd) if q_des is not None else None
    vdes = _prep(qd_des, n, nd) if qd_des is not None else None
    null = C.POINTER(C.c_double)()
    capi.check(capi.lib().tg_sim_ticks(C.byref(robot), capi.PHYSICS[dtype], n, int(n_ticks), int(iters), float(dt), int(motor_mode),
                                       _dp(qdes) if qdes is not None else null, _dp(vdes) if vdes is not None else null,
                                       float(max_force), _dp(q), _dp(qd)))
    return q, qd


def inverse_kinematics(robot, q0, target_pos, target_rot, max_iters=100, threshold=1e-8, dtype="f64"):
    """calculateInverseKinematics at the TCP frame (base_robot_arm.py:201-209); returns (q [n,ndof], iterations [n])."""
    q0 = np.atleast_2d(np.asarray(q0, dtype=np.float64))
    n, nd = q0.shape
    q0 = _prep(q0, n, nd)
    tp = np.ascontiguousarray(target_pos, dtype=np.float64).reshape(n, 3)
    tr = np.ascontiguousarray(target_rot, dtype=np.float64).reshape(n, 9)
    out, iters = np.zeros((n, nd)), np.zeros(n, dtype=np.int32)
    capi.check(capi.lib().tg_inverse_kinematics(C.byref(robot), capi.PHYSICS[dtype], n, _dp(q0), _dp(tp), _dp(tr), int(max_iters),
                                                float(threshold), _dp(out), iters.ctypes.data_as(C.POINTER(C.c_int32))))
    return out, iters


def render_tactile(sensor_desc, mesh_desc, cam_from_obj):
    """getCameraImage depth + t_s_camera (tactile_sensor.py:239-294) for transforms [n,12] -> uint8 [n,H,W]."""
    xf = np.ascontiguousarray(cam_from_obj, dtype=np.float32).reshape(-1, 12)
    n = xf.shape[0]
    h, w = sensor_desc.struct.image_h, sensor_desc.struct.image_w
    out = np.zeros((n, h, w), dtype=np.uint8)
    capi.check(capi.lib().tg_render_tactile(C.byref(sensor_desc.struct), C.byref(mesh_desc.struct), n,
                                            xf.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def gen_heightfield(seeds, rows=64, cols=64, interp=0.05, height_range=0.025):
    """gen_heigtfield_simplex_2d (base_surface_env.py:319-337) on the device: (heights [n,rows,cols] f64, zoff [n] f32)."""
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    n = seeds.shape[0]
    h, z = np.zeros((n, rows, cols)), np.zeros(n, dtype=np.float32)
    capi.check(capi.lib().tg_gen_heightfield(n, seeds.ctypes.data_as(C.POINTER(C.c_int64)), rows, cols, float(interp), float(height_range),
                                             _dp(h), z.ctypes.data_as(C.POINTER(C.c_float))))
    return h, z


def render_tactile_heightfield(sensor_desc, heights, zoff, cam_from_obj, grid_scale=0.006):
    """Tactile image of per-image heightfields (base_surface_env.py:402-432 + tactile_sensor.py:239-294)."""
    h = np.ascontiguousarray(heights, dtype=np.float64)
    n, rows, cols = h.shape
    z = np.ascontiguousarray(zoff, dtype=np.float32).reshape(n)
    xf = np.ascontiguousarray(cam_from_obj, dtype=np.float32).reshape(n, 12)
    out = np.zeros((n, sensor_desc.struct.image_h, sensor_desc.struct.image_w), dtype=np.uint8)
    capi.check(capi.lib().tg_render_tactile_heightfield(C.byref(sensor_desc.struct), rows, cols, float(grid_scale), n, _dp(h),
                                                        z.ctypes.data_as(C.POINTER(C.c_float)), xf.ctypes.data_as(C.POINTER(C.c_float)),
                                                        out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out
