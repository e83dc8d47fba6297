"""ctypes binding of oracle/libminibullet.so (TEST INFRASTRUCTURE ONLY — see oracle/minibullet.h).

`Arm` bundles a model + state and exposes the PyBullet calls the reference makes on its robot
(robots/arms/base_robot_arm.py, robots/arms/robot.py) under PyBullet-like names.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

from . import pb_math as pm

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_DOF, MAX_BODIES = 8, 24


class MBModel(C.Structure):
    _fields_ = [
        ("ndof", C.c_int32), ("nbodies", C.c_int32), ("parent", C.c_int32 * MAX_DOF),
        ("joint_pos", (C.c_double * 3) * MAX_DOF), ("joint_rot", (C.c_double * 9) * MAX_DOF),
        ("joint_axis", (C.c_double * 3) * MAX_DOF), ("body_link", C.c_int32 * MAX_BODIES),
        ("body_com", (C.c_double * 3) * MAX_BODIES), ("body_rot", (C.c_double * 9) * MAX_BODIES),
        ("body_mass", C.c_double * MAX_BODIES), ("body_inertia", (C.c_double * 3) * MAX_BODIES),
        ("gravity", C.c_double * 3), ("linear_damping", C.c_double), ("angular_damping", C.c_double),
        ("joint_damping", C.c_double),
    ]


class MBState(C.Structure):
    _fields_ = [
        ("q", C.c_double * MAX_DOF), ("qd", C.c_double * MAX_DOF), ("applied_torque", C.c_double * MAX_DOF),
        ("motor_mode", C.c_int32 * MAX_DOF), ("motor_q_des", C.c_double * MAX_DOF), ("motor_qd_des", C.c_double * MAX_DOF),
        ("motor_kp", C.c_double * MAX_DOF), ("motor_kd", C.c_double * MAX_DOF), ("motor_max_force", C.c_double * MAX_DOF),
    ]


class MBBody(C.Structure):
    _fields_ = [
        ("mass", C.c_double), ("com", C.c_double * 3), ("inertia", C.c_double * 9), ("pos", C.c_double * 3), ("rot", C.c_double * 9),
        ("linvel", C.c_double * 3), ("angvel", C.c_double * 3), ("ext_force", C.c_double * 3), ("ext_pos", C.c_double * 3),
        ("ext_pending", C.c_int32),
    ]


class MBP2P(C.Structure):
    _fields_ = [("link", C.c_int32), ("pivot_a", C.c_double * 3), ("pivot_b", C.c_double * 3), ("erp", C.c_double),
                ("max_impulse", C.c_double)]


class MBBall(C.Structure):
    _fields_ = [("radius", C.c_double), ("mass", C.c_double), ("inertia", C.c_double), ("pos", C.c_double * 3), ("linvel", C.c_double * 3),
                ("angvel", C.c_double * 3), ("ext_torque", C.c_double * 3), ("ext_pending", C.c_int32), ("mu", C.c_double),
                ("plate_radius", C.c_double), ("plate_half_len", C.c_double), ("breaking", C.c_double), ("erp", C.c_double),
                ("in_contact", C.c_int32), ("depth", C.c_double), ("normal_impulse", C.c_double)]


class MBManifold(C.Structure):
    _fields_ = [("n", C.c_int32), ("la", (C.c_double * 3) * 4), ("lb", (C.c_double * 3) * 4), ("nrm", (C.c_double * 3) * 4),
                ("pa", (C.c_double * 3) * 4), ("pb", (C.c_double * 3) * 4), ("depth", C.c_double * 4)]


class MBSpin(C.Structure):
    _fields_ = [("dish", MBBody), ("ext_torque", C.c_double * 3), ("torque_pending", C.c_int32), ("n_dish", C.c_int32), ("n_spool", C.c_int32),
                ("dish_hull", C.POINTER(C.c_double)), ("spool_hull", C.POINTER(C.c_double)), ("margin", C.c_double), ("breaking", C.c_double),
                ("erp", C.c_double), ("mu", C.c_double), ("lin_damp", C.c_double), ("ang_damp", C.c_double), ("mani", MBManifold),
                ("n_contacts", C.c_int32), ("normal_impulse", C.c_double)]


class MBPushScene(C.Structure):
    _fields_ = [
        ("table_z", C.c_double), ("half", C.c_double * 3), ("mu_table", C.c_double), ("mu_tip", C.c_double),
        ("margin_cube", C.c_double), ("margin_tip", C.c_double), ("breaking", C.c_double), ("erp", C.c_double),
        ("tip_stiffness", C.c_double), ("tip_damping", C.c_double), ("lin_damp", C.c_double), ("ang_damp", C.c_double),
        ("tip_link", C.c_int32), ("n_tip", C.c_int32), ("tip_verts", C.POINTER(C.c_double)), ("cone_friction", C.c_int32),
        ("n_contacts", C.c_int32), ("tip_depth", C.c_double), ("tip_normal", C.c_double * 3), ("tip_impulse", C.c_double),
        ("residual_threshold", C.c_double), ("sweeps_used", C.c_int32),
        ("shape", C.c_int32), ("radius", C.c_double), ("cyl_pos", C.c_double * 3), ("cyl_rot", C.c_double * 9),
        ("cyl_half_len", C.c_double), ("cyl_radius", C.c_double), ("contact_ids", C.c_int32 * 8),
        ("narrowphase", C.c_int32), ("mani", MBManifold),
    ]


MOTOR_OFF, MOTOR_VELOCITY, MOTOR_POSITION = 0, 1, 2
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libminibullet.so")
        if not os.path.isfile(so):
            build()
        _lib = C.CDLL(so)
        dp, fp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        mp, sp = C.POINTER(MBModel), C.POINTER(MBState)
        _lib.mb_fk.argtypes = [mp, dp, dp, dp]
        _lib.mb_frame_state.argtypes = [mp, dp, dp, C.c_int, dp, dp, dp, dp, dp, dp]
        _lib.mb_inverse_dynamics.argtypes = [mp, dp, dp, dp, dp]
        _lib.mb_mass_matrix.argtypes = [mp, dp, dp]
        _lib.mb_jacobian.argtypes = [mp, dp, C.c_int, dp, dp]
        _lib.mb_step.argtypes = [mp, sp, C.c_double, C.c_int]
        _lib.mb_set_solver_residual_threshold.argtypes = [C.c_double]
        _lib.mb_get_solver_residual_threshold.restype = C.c_double
        _lib.mb_step_body.argtypes = [mp, sp, C.POINTER(MBBody), C.POINTER(MBP2P), C.c_double, C.c_int]
        _lib.mb_step_push.argtypes = [mp, sp, C.POINTER(MBBody), C.POINTER(MBPushScene), C.c_double, C.c_int]
        _lib.mb_step_body_ball.argtypes = [mp, sp, C.POINTER(MBBody), C.POINTER(MBP2P), C.POINTER(MBBall), C.c_double, C.c_int]
        _lib.mb_gjk_epa_hull_box.argtypes = [dp, C.c_int, dp, dp, dp, dp, dp]
        _lib.mb_gjk_epa_hull_box.restype = C.c_int
        _lib.mb_gjk_epa_hull_hull.argtypes = [dp, C.c_int, dp, C.c_int, dp, dp, dp, dp]
        _lib.mb_gjk_epa_hull_hull.restype = C.c_int
        _lib.mb_step_spin.argtypes = [mp, sp, C.POINTER(MBBody), C.POINTER(MBP2P), C.POINTER(MBSpin), C.c_double, C.c_int]
        _lib.mb_opensimplex_perm.argtypes = [C.c_int64, C.POINTER(C.c_int16)]
        _lib.mb_opensimplex_noise2.argtypes = [C.POINTER(C.c_int16), C.c_double, C.c_double]
        _lib.mb_opensimplex_noise2.restype = C.c_double
        _lib.mb_ik.argtypes = [mp, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_double]
        _lib.mb_ik.restype = C.c_int
        _lib.mb_render_depth.argtypes = [fp, C.c_int, ip, C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, fp]
        _lib.mb_t_s_camera.argtypes = [fp, fp, fp, u8p, C.c_int, C.c_int, u8p]
        _lib.mb_render_scene.argtypes = [fp, ip, u8p, u8p, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, u8p,
                                         C.POINTER(C.c_uint64), u8p]
        _lib.mb_blend_spheres.argtypes = [fp, C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_uint64), u8p]
    return _lib


def set_solver_residual_threshold(t):
    """btContactSolverInfo::m_leastSquaresResidualThreshold for every mb_step* of this process (PARITY A7b; 0 = exact fixed point)."""
    lib().mb_set_solver_residual_threshold(float(t))


def solver_residual_threshold():
    return float(lib().mb_get_solver_residual_threshold())


def last_sweeps():
    """PGS sweeps the last mb_step / mb_step_body / mb_step_body_ball / mb_step_push executed."""
    return int(C.c_int.in_dll(lib(), "mb_last_sweeps").value)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_model(tg, gravity=(0.0, 0.0, -9.81), linear_damping=0.04, angular_damping=0.04, joint_damping=0.01):
    """tg: tactile_gym_amd.urdf_compile.TGModel (plain data)."""
    m = MBModel()
    m.ndof, m.nbodies = tg.ndof, len(tg.body_mass)
    assert m.ndof <= MAX_DOF and m.nbodies <= MAX_BODIES
    for i in range(tg.ndof):
        m.parent[i] = int(tg.parent[i])
        for k in range(3):
            m.joint_pos[i][k] = float(tg.joint_pos[i][k])
            m.joint_axis[i][k] = float(tg.joint_axis[i][k])
        for k in range(9):
            m.joint_rot[i][k] = float(tg.joint_rot[i].reshape(9)[k])
    for b in range(m.nbodies):
        m.body_link[b] = int(tg.body_link[b])
        m.body_mass[b] = float(tg.body_mass[b])
        for k in range(3):
            m.body_com[b][k] = float(tg.body_com[b][k])
            m.body_inertia[b][k] = float(tg.body_inertia[b][k])
        for k in range(9):
            m.body_rot[b][k] = float(tg.body_rot[b].reshape(9)[k])
    for k in range(3):
        m.gravity[k] = float(gravity[k])
    m.linear_damping, m.angular_damping, m.joint_damping = linear_damping, angular_damping, joint_damping
    return m


class Arm:
    """One simulated arm: the subset of the PyBullet API used on the reference's hot path."""

    def __init__(self, tg, **dyn):
        self.tg = tg
        self.n = tg.ndof
        self.model = make_model(tg, **dyn)
        self.state = MBState()
        self.L = lib()

    # -- state ------------------------------------------------------------------------------------
    @property
    def q(self):
        return np.array(self.state.q[: self.n])

    @property
    def qd(self):
        return np.array(self.state.qd[: self.n])

    def reset_joint_states(self, q):  # resetJointState: position set, velocity zeroed (base_robot_arm.py:22-23)
        for i in range(self.n):
            self.state.q[i] = float(q[i])
            self.state.qd[i] = 0.0

    def set_gravity(self, g):
        for k in range(3):
            self.model.gravity[k] = float(g[k])

    # -- motors -----------------------------------------------------------------------------------
    def set_motors_velocity(self, qd_des, kd, max_force):  # VELOCITY_CONTROL (base_robot_arm.py:325-332)
        for i in range(self.n):
            self.state.motor_mode[i] = MOTOR_VELOCITY
            self.state.motor_qd_des[i] = float(qd_des[i])
            self.state.motor_q_des[i] = 0.0
            self.state.motor_kp[i] = 0.0
            self.state.motor_kd[i] = float(kd)
            self.state.motor_max_force[i] = float(max_force)

    def set_motors_position(self, q_des, qd_des, kp, kd, max_force):  # POSITION_CONTROL (base_robot_arm.py:211-220)
        for i in range(self.n):
            self.state.motor_mode[i] = MOTOR_POSITION
            self.state.motor_q_des[i] = float(q_des[i])
            self.state.motor_qd_des[i] = float(qd_des[i])
            self.state.motor_kp[i] = float(kp)
            self.state.motor_kd[i] = float(kd)
            self.state.motor_max_force[i] = float(max_force)

    def apply_torques(self, tau):  # TORQUE_CONTROL adds a feed-forward torque, motors stay on [A5]
        for i in range(self.n):
            self.state.applied_torque[i] += float(tau[i])

    # -- queries ----------------------------------------------------------------------------------
    def link_poses(self, q=None):
        """World pose (R [3,3], p [3]) of every moving link's frame (origin at its joint)."""
        q = np.ascontiguousarray(self.q if q is None else q, dtype=np.float64)
        R, p = np.zeros((self.n, 9)), np.zeros((self.n, 3))
        self.L.mb_fk(C.byref(self.model), _dp(q), _dp(R), _dp(p))
        return [(R[i].reshape(3, 3), p[i]) for i in range(self.n)]

    def link_state(self, frame, q=None, qd=None):
        """getLinkState(..., computeLinkVelocity=1): (pos, quat, lin_vel, ang_vel) of a named frame."""
        link, fpos, frot = self.tg.frames[frame]
        q = np.ascontiguousarray(self.q if q is None else q, dtype=np.float64)
        qd = np.ascontiguousarray(self.qd if qd is None else qd, dtype=np.float64)
        fpos = np.ascontiguousarray(fpos, dtype=np.float64)
        frot = np.ascontiguousarray(frot, dtype=np.float64)
        pos, rot, lv, av = np.zeros(3), np.zeros(9), np.zeros(3), np.zeros(3)
        self.L.mb_frame_state(C.byref(self.model), _dp(q), _dp(qd), link, _dp(fpos), _dp(frot), _dp(pos), _dp(rot), _dp(lv), _dp(av))
        return pos, pm.quat_from_mat(rot.reshape(3, 3)), lv, av, rot.reshape(3, 3)

    def inverse_dynamics(self, q, qd, qdd):
        q, qd, qdd = (np.ascontiguousarray(v, dtype=np.float64) for v in (q, qd, qdd))
        tau = np.zeros(self.n)
        self.L.mb_inverse_dynamics(C.byref(self.model), _dp(q), _dp(qd), _dp(qdd), _dp(tau))
        return tau

    def mass_matrix(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        M = np.zeros((self.n, self.n))
        self.L.mb_mass_matrix(C.byref(self.model), _dp(q), _dp(M))
        return M

    def jacobian(self, frame, q):
        link, fpos, _ = self.tg.frames[frame]
        q = np.ascontiguousarray(q, dtype=np.float64)
        fpos = np.ascontiguousarray(fpos, dtype=np.float64)
        J = np.zeros((6, self.n))
        self.L.mb_jacobian(C.byref(self.model), _dp(q), link, _dp(fpos), _dp(J))
        return J

    def inverse_kinematics(self, frame, target_pos, target_quat, max_iters=100, residual_threshold=1e-8):
        link, fpos, frot = self.tg.frames[frame]
        q = self.q.copy()
        tp = np.ascontiguousarray(target_pos, dtype=np.float64)
        tr = np.ascontiguousarray(pm.mat_from_quat(target_quat), dtype=np.float64)
        fpos = np.ascontiguousarray(fpos, dtype=np.float64)
        frot = np.ascontiguousarray(frot, dtype=np.float64)
        self.L.mb_ik(C.byref(self.model), link, _dp(fpos), _dp(frot), _dp(tp), _dp(tr), _dp(q), max_iters, residual_threshold)
        return q

    def step_simulation(self, dt=1.0 / 240.0, iters=150):
        self.L.mb_step(C.byref(self.model), C.byref(self.state), dt, iters)

    def step_simulation_push(self, cube, scene, dt=1.0 / 240.0, iters=150):
        """stepSimulation with a free cube on the table pushed by the tip's collision core."""
        self.L.mb_step_push(C.byref(self.model), C.byref(self.state), C.byref(cube), C.byref(scene), dt, iters)

    def step_simulation_body_ball(self, plate, p2p, ball, dt=1.0 / 240.0, iters=150):
        """stepSimulation with the round plate tied to the arm and the ball on it (object_balance, ball_on_plate)."""
        self.L.mb_step_body_ball(C.byref(self.model), C.byref(self.state), C.byref(plate), C.byref(p2p), C.byref(ball), dt, iters)

    def step_simulation_spin(self, spool, p2p, spin, dt=1.0 / 240.0, iters=150):
        """stepSimulation with the spool tied to the arm and the dish standing on it (object_balance, spinning_plate)."""
        self.L.mb_step_spin(C.byref(self.model), C.byref(self.state), C.byref(spool), C.byref(p2p), C.byref(spin), dt, iters)

    def step_simulation_body(self, body, p2p, dt=1.0 / 240.0, iters=150):
        """stepSimulation with a free rigid body tied to the arm by a point-to-point constraint."""
        self.L.mb_step_body(C.byref(self.model), C.byref(self.state), C.byref(body), C.byref(p2p), dt, iters)


# --------------------------------------------------------------------------------------------------- camera
def render_depth(verts, tris, cam_from_obj, fov, near, far, w, h, depth):
    """z-test a triangle soup into `depth` (float32[h,w], in place)."""
    v = np.ascontiguousarray(verts, dtype=np.float32)
    t = np.ascontiguousarray(tris, dtype=np.int32)
    M = np.ascontiguousarray(cam_from_obj, dtype=np.float32).reshape(12)
    assert depth.dtype == np.float32 and depth.flags.c_contiguous
    fp = C.POINTER(C.c_float)
    lib().mb_render_depth(v.ctypes.data_as(fp), v.shape[0], t.ctypes.data_as(C.POINTER(C.c_int32)), t.shape[0],
                          M.ctypes.data_as(fp), fov, near, far, w, h, depth.ctypes.data_as(fp))
    return depth


def scene_view_matrix(target, dist, yaw_deg, pitch_deg):
    """computeViewMatrixFromYawPitchRoll(cameraTargetPosition, distance, yaw, pitch, roll=0, upAxisIndex=2) as the reference calls it
    (base_tactile_env.py:217-224), returned as world -> eye (R [3,3], t [3]).  Bullet [PARITY_ASSUMPTIONS A31]: the eye sits at
    target + Rz(yaw) Rx(pitch) (0, -distance, 0) with up = Rz(yaw) Rx(pitch) (0, 0, 1), then the look-at matrix of (eye, target, up)."""
    y, p = math.radians(yaw_deg), math.radians(pitch_deg)
    Rz = np.array([[math.cos(y), -math.sin(y), 0.0], [math.sin(y), math.cos(y), 0.0], [0.0, 0.0, 1.0]])
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, math.cos(p), -math.sin(p)], [0.0, math.sin(p), math.cos(p)]])
    E = Rz @ Rx
    eye = np.asarray(target, dtype=np.float64) + E @ np.array([0.0, -float(dist), 0.0])
    up = E @ np.array([0.0, 0.0, 1.0])
    f = np.asarray(target, dtype=np.float64) - eye
    f = f / np.linalg.norm(f)
    s = np.cross(f, up)
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.stack([s, u, -f])
    return V, -V @ eye


def render_scene(verts, tris, tri_frame, tri_rgb, frames, view, light_dir_world, fov, near, far, w, h, background, spheres=None):
    """Scene camera rgb (uint8 [h, w, 3]).  frames: world poses [(R [3,3], p [3]), ...] that tri_frame indexes; view = scene_view_matrix(...).
    spheres: optional list of (world centre [3], radius, (r, g, b) 0..255, alpha) blended over the opaque image in list order (mb_blend_spheres)."""
    V, tv = view
    xf = np.zeros((len(frames), 12), dtype=np.float32)
    for i, (R, p) in enumerate(frames):
        xf[i, :9] = (V @ np.asarray(R, dtype=np.float64)).reshape(9)
        xf[i, 9:] = V @ np.asarray(p, dtype=np.float64) + tv
    L = np.asarray(light_dir_world, dtype=np.float64)
    le = np.ascontiguousarray(V @ (L / np.linalg.norm(L)), dtype=np.float32)
    v = np.ascontiguousarray(verts, dtype=np.float32)
    t = np.ascontiguousarray(tris, dtype=np.int32)
    tf = np.ascontiguousarray(tri_frame, dtype=np.uint8)
    tc = np.ascontiguousarray(tri_rgb, dtype=np.uint8)
    bg = np.ascontiguousarray(background, dtype=np.uint8)
    z = np.zeros(w * h, dtype=np.uint64)
    out = np.zeros((h, w, 3), dtype=np.uint8)
    fp, u8 = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    lib().mb_render_scene(v.ctypes.data_as(fp), t.ctypes.data_as(C.POINTER(C.c_int32)), tf.ctypes.data_as(u8), tc.ctypes.data_as(u8), t.shape[0],
                          xf.ctypes.data_as(fp), le.ctypes.data_as(fp), fov, near, far, w, h, bg.ctypes.data_as(u8),
                          z.ctypes.data_as(C.POINTER(C.c_uint64)), out.ctypes.data_as(u8))
    if spheres:
        sp = np.zeros((len(spheres), 8), dtype=np.float32)
        for i, (c, r, rgb, alpha) in enumerate(spheres):
            sp[i, :3] = V @ np.asarray(c, dtype=np.float64) + tv          # eye-space centre, rounded to float like the frames' translations
            sp[i, 3], sp[i, 4:7], sp[i, 7] = r, rgb, alpha
        lib().mb_blend_spheres(sp.ctypes.data_as(fp), len(spheres), le.ctypes.data_as(fp), fov, near, far, w, h, z.ctypes.data_as(C.POINTER(C.c_uint64)),
                               out.ctypes.data_as(u8))
    return out


def t_s_camera(cur_dep, nodef_dep, nodef_gray, border_mask, turn_off_border=False):
    cur = np.ascontiguousarray(cur_dep, dtype=np.float32)
    nd = np.ascontiguousarray(nodef_dep, dtype=np.float32)
    ng = np.ascontiguousarray(nodef_gray, dtype=np.float32)
    bm = np.ascontiguousarray(border_mask, dtype=np.uint8)
    out = np.zeros(cur.shape, dtype=np.uint8)
    fp, u8 = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    lib().mb_t_s_camera(cur.ctypes.data_as(fp), nd.ctypes.data_as(fp), ng.ctypes.data_as(fp), bm.ctypes.data_as(u8),
                        cur.size, int(turn_off_border), out.ctypes.data_as(u8))
    return out


def cam_from_obj_matrix(cam_pos, cam_rot, obj_pos, obj_rot):
    """[R|t] (12 floats, R row-major then t) taking object coordinates to GL eye space of the tactile camera.

    Camera convention (tactile_sensor.py:221-229): forward = R_cam[:,0], up = R_cam[:,2]; computeViewMatrix builds
    right = forward x up, so eye-space axes are (right, up, -forward)."""
    cam_rot = np.asarray(cam_rot, dtype=np.float64)
    fwd, up = cam_rot[:, 0], cam_rot[:, 2]
    f = fwd / np.linalg.norm(fwd)
    s = np.cross(f, up)
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.stack([s, u, -f])  # world -> eye rotation
    R = V @ np.asarray(obj_rot, dtype=np.float64)
    t = V @ (np.asarray(obj_pos, dtype=np.float64) - np.asarray(cam_pos, dtype=np.float64))
    return np.concatenate([R.reshape(9), t]).astype(np.float32)
