"""PyBullet cross-check kit (VERDICT r4 "Next 7").  Needs NO tactile_gym source: raw `pybullet` + a checkout of the reference's ASSETS
directory (`tactile_gym/assets`: URDFs and meshes only).

    python tools/pybullet_probe.py --backend pybullet --assets /path/to/tactile_gym/assets --out tests/golden
    python tools/pybullet_probe.py --backend oracle --out /tmp/probe            # the same scenarios through oracle/ (format check)

What it does: runs nine scenarios of the step's hot path - the PyBullet calls `BaseTactileEnv.step` makes, restated as a script of
backend-neutral operations - through one of two backends and writes `pybullet_<scenario>.npz` (inputs + recorded outputs):

    arm_statics       calculateInverseDynamics(q, 0, 0), calculateMassMatrix(q), calculateJacobian(TCP) at three poses
                      -> closes PARITY_ASSUMPTIONS A1-A3 (inertial frames, AABB-derived inertias, link-frame conventions)
    arm_velocity      resetJointState(rest), changeDynamics(damping), 48 x [gravity compensation (TORQUE_CONTROL) + VELOCITY_CONTROL motors +
                      stepSimulation] (base_tactile_env.py:125-139, robot.py:131-141, base_robot_arm.py:174-189, 325-332); q, qd and the TCP
                      link state per tick -> A4-A7 (integration order, damping, motor constraint, A7b the solver's exit / residual threshold)
    reset_move        Robot.reset (robot.py:114-125): calculateInverseKinematics to edge_follow's start pose + blocking_move under
                      POSITION_CONTROL (robot.py:188-260); IK solution, tick count, final q -> A9-A11
    tactile_depth     the in-sensor camera (tactile_sensor.py:150-246): view / projection from the sensor body link, getCameraImage depth of the
                      edge stimulus at a given pose, 128 x 128 -> A12-A16 (camera model, depth buffer convention, raster rules)
    push_contacts     object_push's contact path on a UR5 + right-angle TacTip (tip collision core on, object_push_env.py:40-56, 196-227): Robot.reset
                      to the work-frame origin, the cube at its start pose with the env's changeDynamics, then 10 control steps of a constant
                      work-frame push (24 ticks each); per tick the cube's pose and velocity, which cube - table and cube - tip contacts
                      exist (getContactPoints), the tip contact's normal and distance -> A23-A29 (contact sets, margins, soft tip contact,
                      cone friction) and row n1 of the survey (contact-pair indices)

    balance_constraint  object_balance's pole on its point-to-point constraint (object_balance_env.py:173-199, 261-294, 328-381): UR5 + standard
                      TacTip pointing up (tip collisions off: "no_core"), Robot.reset to the work-frame origin, the pole teleported onto the tip
                      with zero damping, gravity -0.5, the one-shot 0.1 N push at a fixed off-centre point, then 10 control steps of 12
                      ticks under a constant work-frame twist (x, y and a tilt about x); per tick the pole's pose and velocity, the arm's joints
                      and the pivot gap -> A18-A21 (the constraint's erp and row order, free-body integration, the one-shot force, which frame
                      get/resetBasePositionAndOrientation speak)

    ball_on_plate     object_balance's other object (:187-199, 241-260, 350-353, 393-401): the round plate on the same constraint, the ball
                      (sphere.urdf x globalScaling 7.5, lateralFriction 10) put on it, the one-shot 0.001 N m torque on the ball, then 10 control
                      steps under the same twist; per tick the plate's pose, the ball's position and velocities, the joints -> A39 (and A26 / A30:
                      friction combination, what globalScaling scales)

    push_manifold     the same push, the same PyBullet world; the ORACLE side runs its general narrowphase (GJK / EPA + a Bullet-style persistent
                      manifold of up to four tip - cube points, oracle/narrowphase.c) instead of the closed form, and the number of tip - cube
                      contact points per tick is compared as well -> A35-A38

    roll_contacts     object_roll's marble (object_roll_env.py:55-71, 156-237): UR5 + flat TacTip (tip core on, soft contact 10 / 100, friction 10),
                      sphere.urdf on the table with the env's changeDynamics, Robot.reset onto it (embed 2.5 mm), 5 control steps of 24 ticks
                      under an x / y work-frame velocity; per tick the marble's position and velocities, whether it touches table and tip, the
                      tip point's distance, the joints -> A30 (sphere - plane and sphere - cylinder contacts, sphere inertia, friction products)

tests/test_pybullet_golden.py compares oracle/ with every `tests/golden/pybullet_*.npz` it finds (tolerances and the assumption each
comparison closes are in the test) and always runs the oracle backend against itself through a temporary directory, so the file format and
the comparison code are exercised without PyBullet.  bench.py's `cpu_baseline` times `arm_velocity` through PyBullet when it is importable and
TG_PYBULLET_ASSETS points at the assets.

This container has no pybullet: the PyBullet backends below are written against PyBullet's documented API and the reference's call sites.  They
have been executed only against a stub of that API (tests/test_pybullet_stub.py: the Quickstart Guide's parameter names as real signatures, link
names from the robots' URDFs) - call names, keywords, argument counts and the files' fields / shapes are checked, nothing PyBullet computes."""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIM_DT, SOLVER_ITERS = 1.0 / 240.0, 150
UR5_REST = [0.166827, -2.16515, -1.64365, -0.90317, 1.57315, 1.74001]           # edge_follow/rest_poses.py (ur5, tactip, standard)
POSES = [UR5_REST, [0.3, -1.9, -1.2, -1.4, 1.2, 0.4], [-0.5, -1.2, -2.0, -0.3, 1.9, -1.0]]
JOINT_VEL = [0.02, -0.015, 0.01, 0.025, -0.02, 0.03]                            # rad/s targets of arm_velocity (well inside the motor limits)
MAX_FORCE, POS_GAIN, VEL_GAIN = 1000.0, 1.0, 1.0                                # ur5.py:19-21
WORKFRAME = ([0.65, 0.0, 0.035], [-math.pi, 0.0, math.pi / 2])                   # edge_follow_env.py:106-107
EDGE_POS, EDGE_ANG, EMBED = [0.65, 0.0, 0.0], 0.7, 0.0035


# ----------------------------------------------------------------------------------------------------------------- backends
class OracleBackend:
    """oracle/minibullet (the CPU restatement this repo's GPU path is tested against)."""
    name = "oracle"

    def __init__(self, assets=None):
        from oracle import minibullet as mb, pb_math as pm
        from oracle.ref_env import load_tg, sensor_camera
        self.mb, self.pm = mb, pm
        self.tg = load_tg("ur5_standard_tactip")
        self.arm = mb.Arm(self.tg)
        self.cam = sensor_camera("tactip", "standard")
        self.edge = np.load(os.path.join(ROOT, "tactile_gym_amd", "assets", "stimuli", "long_edge.npz"))

    def reset_joints(self, q):
        self.arm.reset_joint_states(q)
        self.arm.set_motors_position(q, np.zeros(self.arm.n), POS_GAIN, VEL_GAIN, MAX_FORCE)    # base_robot_arm.py:26-37 hold

    def joints(self):
        return self.arm.q, self.arm.qd

    def inverse_dynamics(self, q):
        return self.arm.inverse_dynamics(q, np.zeros(self.arm.n), np.zeros(self.arm.n))

    def mass_matrix(self, q):
        return self.arm.mass_matrix(q)

    def jacobian_tcp(self, q):
        return self.arm.jacobian("tcp_link", np.asarray(q, dtype=np.float64))

    def tcp_state(self):
        pos, quat, lv, av, _ = self.arm.link_state("tcp_link")
        return np.concatenate([pos, quat, lv, av])

    def motors_velocity(self, qd_des):
        self.arm.set_motors_velocity(qd_des, VEL_GAIN, MAX_FORCE)

    def motors_position(self, q_des, max_force):
        self.arm.set_motors_position(q_des, np.zeros(self.arm.n), POS_GAIN, VEL_GAIN, max_force)

    def tick(self):
        q, qd = self.joints()
        self.arm.apply_torques(self.arm.inverse_dynamics(q, qd, np.zeros(self.arm.n)))          # apply_gravity_compensation
        self.arm.step_simulation(SIM_DT, SOLVER_ITERS)

    def ik_tcp(self, pos, quat):
        return self.arm.inverse_kinematics("tcp_link", pos, quat, 100, 1e-8)

    def depth_of_edge(self, size):
        bpos, bquat, _, _, _ = self.arm.link_state("tactip_body_link")
        cpos, cquat = self.pm.multiply_transforms(bpos, bquat, self.cam["pos"], self.pm.quat_from_euler(self.cam["rpy"]))
        c, s = math.cos(EDGE_ANG), math.sin(EDGE_ANG)
        rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        M = self.mb.cam_from_obj_matrix(cpos, self.pm.mat_from_quat(cquat), np.array(EDGE_POS), rot)
        dep = np.ones((size, size), dtype=np.float32)                                           # far plane = 1.0 in the depth buffer
        self.mb.render_depth(self.edge["verts"], self.edge["tris"], M, self.cam["fov"], self.cam["near"], self.cam["far"], size, size, dep)
        return dep


class PyBulletBackend:
    """Raw pybullet on the reference's assets (loadURDF + the engine parameters of base_tactile_env.py:125-139)."""
    name = "pybullet"

    def __init__(self, assets):
        import pybullet as p
        self.p = p
        if not assets or not os.path.isdir(assets):
            raise SystemExit("--assets must point at the reference's `tactile_gym/assets` directory")
        self.assets = assets
        self.cid = p.connect(p.DIRECT)
        p.setGravity(0, 0, -9.81)
        p.setPhysicsEngineParameter(fixedTimeStep=SIM_DT, numSolverIterations=SOLVER_ITERS, enableConeFriction=1, contactBreakingThreshold=0.0001)
        p.loadURDF(os.path.join(assets, "shared_assets/environment_objects/plane/plane.urdf"), [0, 0, -0.625])
        p.loadURDF(os.path.join(assets, "shared_assets/environment_objects/table/table.urdf"), [0.50, 0.00, -0.625], [0.0, 0.0, 0.0, 1.0])
        self.robot = p.loadURDF(os.path.join(assets, "robot_assets/ur5/tactip/ur5_with_standard_tactip.urdf"), [0, 0, 0], [0, 0, 0, 1],
                                useFixedBase=True)                                              # robot.py:95-112
        self.n_all = p.getNumJoints(self.robot)
        info = [p.getJointInfo(self.robot, i) for i in range(self.n_all)]
        self.link = {inf[12].decode(): i for i, inf in enumerate(info)}
        self.ctrl = [i for i, inf in enumerate(info) if inf[2] != p.JOINT_FIXED]               # the six revolute joints, in URDF order
        self.tcp = self.link["tcp_link"]
        self.body = self.link["tactip_body_link"]
        for i in range(self.n_all):                                                            # base_robot_arm.py:22-25
            p.changeDynamics(self.robot, i, linearDamping=0.04, angularDamping=0.04)
            p.changeDynamics(self.robot, i, jointDamping=0.01)
        self.edge = None

    def reset_joints(self, q):
        p = self.p
        for k, i in enumerate(self.ctrl):
            p.resetJointState(self.robot, i, q[k])
        p.setJointMotorControlArray(self.robot, self.ctrl, p.POSITION_CONTROL, targetPositions=list(q), targetVelocities=[0] * len(self.ctrl),
                                    positionGains=[POS_GAIN] * len(self.ctrl), velocityGains=[VEL_GAIN] * len(self.ctrl),
                                    forces=[MAX_FORCE] * len(self.ctrl))

    def joints(self):
        st = self.p.getJointStates(self.robot, self.ctrl)
        return np.array([s[0] for s in st]), np.array([s[1] for s in st])

    def inverse_dynamics(self, q):
        n = len(self.ctrl)
        return np.array(self.p.calculateInverseDynamics(self.robot, list(q), [0.0] * n, [0.0] * n))

    def mass_matrix(self, q):
        return np.array(self.p.calculateMassMatrix(self.robot, list(q)))

    def jacobian_tcp(self, q):
        n = len(self.ctrl)
        jt, jr = self.p.calculateJacobian(self.robot, self.tcp, [0, 0, 0], list(q), [0.0] * n, [0.0] * n)      # base_robot_arm.py:300-310
        return np.concatenate([np.array(jt), np.array(jr)])

    def tcp_state(self):
        s = self.p.getLinkState(self.robot, self.tcp, computeLinkVelocity=True, computeForwardKinematics=True)   # base_robot_arm.py:136-151
        return np.concatenate([s[0], s[1], s[6], s[7]])

    def motors_velocity(self, qd_des):
        p = self.p
        p.setJointMotorControlArray(self.robot, self.ctrl, p.VELOCITY_CONTROL, targetVelocities=list(qd_des),
                                    velocityGains=[VEL_GAIN] * len(self.ctrl), forces=[MAX_FORCE] * len(self.ctrl))

    def motors_position(self, q_des, max_force):
        p = self.p
        kw = {} if max_force is None else {"forces": [max_force] * len(self.ctrl)}
        p.setJointMotorControlArray(self.robot, self.ctrl, p.POSITION_CONTROL, targetPositions=list(q_des),
                                    targetVelocities=[0] * len(self.ctrl), positionGains=[POS_GAIN] * len(self.ctrl),
                                    velocityGains=[VEL_GAIN] * len(self.ctrl), **kw)

    def tick(self):
        p = self.p
        q, qd = self.joints()
        tau = p.calculateInverseDynamics(self.robot, list(q), list(qd), [0.0] * len(self.ctrl))    # compute_gravity_compensation :174-179
        p.setJointMotorControlArray(self.robot, self.ctrl, p.TORQUE_CONTROL, forces=list(tau))
        p.stepSimulation()

    def ik_tcp(self, pos, quat):
        sol = self.p.calculateInverseKinematics(self.robot, self.tcp, list(pos), list(quat), restPoses=list(UR5_REST), maxNumIterations=100,
                                                residualThreshold=1e-8)                         # base_robot_arm.py:201-209
        return np.array(sol)[: len(self.ctrl)]

    def depth_of_edge(self, size):
        p = self.p
        if self.edge is None:                                                                  # edge_follow_env.py:218-235
            orn = p.getQuaternionFromEuler([0, 0, EDGE_ANG])
            self.edge = p.loadURDF(os.path.join(self.assets, "rl_env_assets/exploration/edge_follow/edge_stimuli/long_edge_flat/long_edge.urdf"),
                                   EDGE_POS, orn, useFixedBase=True)
        bpos, born = p.getLinkState(self.robot, self.body, computeForwardKinematics=True)[:2]   # tactile_sensor.py:150-187
        cpos, corn = p.multiplyTransforms(bpos, born, (0, 0, 0.03), p.getQuaternionFromEuler((0, -math.pi / 2, math.pi)))
        R = np.array(p.getMatrixFromQuaternion(corn)).reshape(3, 3)
        fwd, up = R @ np.array([1.0, 0, 0]), R @ np.array([0, 0, 1.0])
        view = p.computeViewMatrix(cpos, np.array(cpos) + 0.065 * fwd, up)                     # :211-224
        proj = p.computeProjectionMatrixFOV(60, 1.0, 0.01, 1.0)
        img = p.getCameraImage(size, size, view, proj, renderer=p.ER_BULLET_HARDWARE_OPENGL, flags=p.ER_SEGMENTATION_MASK_OBJECT_AND_LINKINDEX)
        return np.reshape(img[3], (size, size)).astype(np.float32)


PUSH_WORKFRAME = ([0.55, -0.20, 0.04], [-math.pi, 0.0, math.pi / 2])            # object_push_env.py:87-90 (ur5): well_designed_pos, rpy
PUSH_REST = [-0.29446578243858357, -2.1633703222876646, -1.7712875440608364, -0.7758826291678864, 1.569501010720629, -1.8628739133606422]
PUSH_VEL = [0.01, 0.0, 0.0, 0.0, 0.0, 2.0 * math.pi / 180]                      # work-frame twist of the push: +x (world +y, into the cube) at 1 cm/s, a slow yaw


class OraclePush:
    """The oracle's object_push env (oracle/ref_env.py) driven tick by tick."""
    name = "oracle"

    def __init__(self, assets=None, narrowphase="closed_form"):
        from oracle.ref_env import OracleObjectPushEnv
        modes = dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False, rand_obj_mass=False, traj_type="straight",
                     observation_mode="tactile_and_feature", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
        self.env = OracleObjectPushEnv(seed=1, env_modes=modes, narrowphase=narrowphase)
        self.env.reset()

    def control(self, twist):
        self.env._tcp_velocity_control(np.array(twist, dtype=np.float64))

    def tick(self):
        self.env._step_sim()

    def record(self):
        e = self.env
        pos, R = e.cube_pose()
        ids = [int(e.scene.contact_ids[k]) for k in range(8)][: int(e.scene.n_contacts)]
        tip = any(i >= 8 for i in ids)
        return dict(cube_pos=pos, cube_rot=R.reshape(9), cube_linvel=np.array(e.cube.linvel[:]), cube_angvel=np.array(e.cube.angvel[:]),
                    n_table=np.array(sum(1 for i in ids if i < 8)), n_tip=np.array(sum(1 for i in ids if i >= 8)), tip_contact=np.array(int(tip)),
                    tip_normal=np.array(e.scene.tip_normal[:]) if tip else np.zeros(3), tip_distance=np.array(float(e.scene.tip_depth) if tip else 0.0),
                    q=e.arm.q)


class PyBulletPush(PyBulletBackend):
    """object_push-v0's world in raw pybullet: UR5 + right-angle TacTip with the tip core on, the cube with the env's contact parameters."""

    def __init__(self, assets):
        import pybullet as p
        self.p = p
        if not assets or not os.path.isdir(assets):
            raise SystemExit("--assets must point at the reference's `tactile_gym/assets` directory")
        self.assets = assets
        p.connect(p.DIRECT)
        p.setGravity(0, 0, -9.81)
        p.setPhysicsEngineParameter(fixedTimeStep=SIM_DT, numSolverIterations=SOLVER_ITERS, enableConeFriction=1, contactBreakingThreshold=0.0001)
        p.loadURDF(os.path.join(assets, "shared_assets/environment_objects/plane/plane.urdf"), [0, 0, -0.625])
        self.table = p.loadURDF(os.path.join(assets, "shared_assets/environment_objects/table/table.urdf"), [0.50, 0.00, -0.625], [0.0, 0.0, 0.0, 1.0])
        self.robot = p.loadURDF(os.path.join(assets, "robot_assets/ur5/tactip/ur5_with_right_angle_tactip.urdf"), [0, 0, 0], [0, 0, 0, 1], useFixedBase=True)
        self.n_all = p.getNumJoints(self.robot)
        info = [p.getJointInfo(self.robot, i) for i in range(self.n_all)]
        self.link = {inf[12].decode(): i for i, inf in enumerate(info)}
        self.ctrl = [i for i, inf in enumerate(info) if inf[2] != p.JOINT_FIXED]
        self.tcp, self.body, self.tip = self.link["tcp_link"], self.link["tactip_body_link"], self.link["tactip_tip_link"]
        for i in range(self.n_all):
            p.changeDynamics(self.robot, i, linearDamping=0.04, angularDamping=0.04)
            p.changeDynamics(self.robot, i, jointDamping=0.01)
        p.setCollisionFilterGroupMask(self.robot, self.body, 0, 0)                             # tactile_sensor.py:46-57 (core "fixed": the tip stays on)
        p.setCollisionFilterGroupMask(self.robot, self.link["tactip_adapter_link"], 0, 0)
        p.changeDynamics(self.robot, self.tip, contactDamping=100, contactStiffness=50)        # reset_tip :320-332, t_s_dynamics of object_push_env.py:50-52
        p.changeDynamics(self.robot, self.tip, lateralFriction=10.0)
        pos = [PUSH_WORKFRAME[0][0], PUSH_WORKFRAME[0][1] + 0.04, 0.04]                          # setup_object :158-160
        self.cube = p.loadURDF(os.path.join(assets, "rl_env_assets/nonprehensile_manipulation/object_push/cube/cube.urdf"), pos,
                               p.getQuaternionFromEuler([-math.pi, 0.0, math.pi / 2]))         # base_object_env.py:70
        p.changeDynamics(self.cube, -1, lateralFriction=0.065, spinningFriction=0.0, rollingFriction=0.0, restitution=0.0, frictionAnchor=1,
                         collisionMargin=0.0001)                                                 # reset_object :216-225
        self.edge = None
        # Robot.reset to the work-frame origin, rpy 0 (base_object_env.py:96-103, robot.py:114-125)
        self.reset_joints(PUSH_REST)
        wq = _quat_from_euler(*PUSH_WORKFRAME[1])
        tpos, tq = np.array(PUSH_WORKFRAME[0]), _quat_mul(wq, _quat_from_euler(0.0, 0.0, 0.0))
        targ = np.array(p.calculateInverseKinematics(self.robot, self.tcp, list(tpos), list(tq), restPoses=list(PUSH_REST), maxNumIterations=100,
                                                     residualThreshold=1e-8))[: len(self.ctrl)]
        self.motors_position(targ, MAX_FORCE)
        _blocking_move(self, tpos, tq, targ)
        p.resetBasePositionAndOrientation(self.cube, pos, p.getQuaternionFromEuler([-math.pi, 0.0, math.pi / 2]))   # reset_object after the robot

    def control(self, twist):
        p = self.p
        wq = _quat_from_euler(*PUSH_WORKFRAME[1])
        tw = np.concatenate([_rotate(wq, twist[:3]), _rotate(wq, twist[3:])])                   # workvel_to_worldvel; the limits do not bind here
        q, _ = self.joints()
        req = np.linalg.inv(self.jacobian_tcp(q)) @ tw                                         # base_robot_arm.py:300-322
        self.motors_velocity(req)

    def record(self):
        p = self.p
        pos, orn = p.getBasePositionAndOrientation(self.cube)
        lv, av = p.getBaseVelocity(self.cube)
        R = np.array(p.getMatrixFromQuaternion(orn))
        table = p.getContactPoints(self.cube, self.table)
        tip = [c for c in p.getContactPoints(self.robot, self.cube) if c[3] == self.tip]
        deep = min(tip, key=lambda c: c[8]) if tip else None
        return dict(cube_pos=np.array(pos), cube_rot=R, cube_linvel=np.array(lv), cube_angvel=np.array(av), n_table=np.array(len(table)),
                    n_tip=np.array(len(tip)), tip_contact=np.array(int(bool(tip))), tip_normal=np.array(deep[7]) if deep else np.zeros(3),
                    tip_distance=np.array(deep[8] if deep else 0.0), q=self.joints()[0])


BAL_WORKFRAME = ([0.55, 0.0, 0.35], [0.0, 0.0, 0.0])                            # object_balance_env.py:71-73
BAL_REST = [0.19826, -2.01062, -1.96602, -0.73808, 4.71286, -3.34064]           # object_balance/rest_poses.py (ur5, standard)
BAL_GRAVITY, BAL_EMBED = -0.5, 0.0045                                           # inside reset_task's ranges (:301-316), fixed here
BAL_PUSH_AT = (0.3, -0.2)                                                       # the push point's offset from the pole's base centre, in half base widths (:364-370)
BAL_BASE_W, BAL_BASE_H = 0.1, 0.0025                                            # setup_object :176-178
BAL_VEL = [0.004, -0.003, 0.0, 2.0 * math.pi / 180, 0.0, 0.0]                   # work-frame twist: x, y, a tilt about x (inside the action ranges :123-131)
BALL_TORQUE = (0.6, -0.4)                                                       # apply_random_torque_ball's two draws in [-1, 1] (:393-401), fixed here


class OracleBalance:
    """The oracle's object_balance env (pole) driven tick by tick, with reset_task's draws replaced by the fixed values above."""
    name = "oracle"

    def __init__(self, assets=None, ball=False):
        from oracle.ref_env import OracleObjectBalanceEnv
        modes = dict(movement_mode="xyRxRy", control_mode="TCP_velocity_control", object_mode="ball_on_plate" if ball else "pole", rand_gravity=False,
                     rand_embed_dist=False, observation_mode="tactile", reward_mode="dense", arm_type="ur5", tactile_sensor_name="tactip")
        e = OracleObjectBalanceEnv(seed=1, env_modes=modes)
        self.ball = ball
        e.step_counter = 0
        e.gravity = BAL_GRAVITY                                                          # reset_task (:295-321) with fixed draws
        e.arm.set_gravity([0.0, 0.0, BAL_GRAVITY])
        e.embed_dist = BAL_EMBED
        e._set_init_obj_pos(buffer_height=0.0)
        e._update_constraint()
        e._reset_robot(np.zeros(3), np.zeros(3))                                         # base_object_env.py:171-172
        e._teleport_body(e.init_obj_pos, e.init_obj_rot)                                 # reset_object :335
        if ball:                                                                         # reset_ball + apply_random_torque_ball(0.001) :350-353, 393-401
            e._teleport_ball()
            for k in range(3):
                e.ball.ext_torque[k] = [BALL_TORQUE[0] * 0.001, BALL_TORQUE[1] * 0.001, 0.0][k]
            e.ball.ext_pending = 1
        else:
            fpos = e.init_obj_pos + np.array([BAL_PUSH_AT[0] * BAL_BASE_W / 2, BAL_PUSH_AT[1] * BAL_BASE_W / 2, 0.0])
            for k in range(3):                                                           # apply_random_force_base(0.1) :360-378
                e.body.ext_force[k] = [0.0, 0.0, -0.1][k]
                e.body.ext_pos[k] = float(fpos[k])
            e.body.ext_pending = 1
        self.env = e

    def control(self, twist):
        self.env._tcp_velocity_control(np.array(twist, dtype=np.float64))

    def tick(self):
        self.env._step_sim()

    def record(self):
        e = self.env
        pos, R = e.body_pose()
        tcp_pos = e._tcp_world()[0]
        pivot_b = pos + R @ np.array([0.0, 0.0, -BAL_BASE_H / 2 + BAL_EMBED])           # the child pivot is given in the base's INERTIAL frame
        out = dict(pole_pos=pos, pole_rot=R.reshape(9), pole_linvel=np.array(e.body.linvel[:]), pole_angvel=np.array(e.body.angvel[:]),
                   q=np.array(e.arm.q), gap=np.asarray(tcp_pos) - pivot_b)
        if self.ball:
            out.update(ball_pos=np.array(e.ball.pos[:]), ball_linvel=np.array(e.ball.linvel[:]), ball_angvel=np.array(e.ball.angvel[:]))
        return out


class PyBulletBalance(PyBulletBackend):
    """object_balance-v0's world (pole) in raw pybullet: the base world of PyBulletBackend (UR5 + standard TacTip), tip collisions off, the
    pole on a JOINT_POINT2POINT constraint to the TCP link."""

    def __init__(self, assets, ball=False):
        super().__init__(assets)
        p = self.p
        self.ball = None
        self.tip = self.link["tactip_tip_link"]
        p.setCollisionFilterGroupMask(self.robot, self.body, 0, 0)                              # tactile_sensor.py:46-57, t_s_core "no_core"
        p.setCollisionFilterGroupMask(self.robot, self.tip, 0, 0)
        init_pos = [BAL_WORKFRAME[0][0], BAL_WORKFRAME[0][1], BAL_WORKFRAME[0][2] + BAL_BASE_H / 2 - BAL_EMBED]    # :185-189, :317-321
        init_orn = p.getQuaternionFromEuler([0.0, 0.0, -math.pi / 2])                           # :190-191
        obj = "round_plate/round_plate.urdf" if ball else "pole/pole.urdf"                       # setup_object :173-199
        self.pole = p.loadURDF(os.path.join(assets, "rl_env_assets/nonprehensile_manipulation/object_balance", obj), init_pos, init_orn)   # base_object_env.py:70
        if ball:                                                                                # load_ball :241-259
            self.ball_pos = [BAL_WORKFRAME[0][0], BAL_WORKFRAME[0][1], BAL_WORKFRAME[0][2] + 0.0025 * 7.5]
            self.ball = p.loadURDF(os.path.join(assets, "rl_env_assets/nonprehensile_manipulation/object_balance/sphere/sphere.urdf"), self.ball_pos,
                                   [0, 0, 0, 1], globalScaling=7.5)
            p.changeDynamics(self.ball, -1, lateralFriction=10.0)
        self.cons = p.createConstraint(self.robot, self.tcp, self.pole, -1, p.JOINT_POINT2POINT, jointAxis=[0, 0, 1], parentFramePosition=[0, 0, 0],
                                       childFramePosition=[0, 0, -BAL_BASE_H / 2 + 0.0035], parentFrameOrientation=p.getQuaternionFromEuler([0, 0, 0]),
                                       childFrameOrientation=p.getQuaternionFromEuler([0, 0, 0]))   # apply_constraints :261-283 (embed_dist 0.0035 at that time)
        # reset(): reset_task with the fixed draws, Robot.reset, reset_object
        p.setGravity(0, 0, BAL_GRAVITY)
        p.changeConstraint(self.cons, jointChildPivot=[0, 0, -BAL_BASE_H / 2 + BAL_EMBED])      # update_constraints :285-294
        self.reset_joints(BAL_REST)
        wq = _quat_from_euler(*BAL_WORKFRAME[1])
        tpos, tq = np.array(BAL_WORKFRAME[0]), _quat_mul(wq, _quat_from_euler(0.0, 0.0, 0.0))
        targ = np.array(p.calculateInverseKinematics(self.robot, self.tcp, list(tpos), list(tq), restPoses=list(BAL_REST), maxNumIterations=100,
                                                     residualThreshold=1e-8))[: len(self.ctrl)]
        self.motors_position(targ, MAX_FORCE)
        _blocking_move(self, tpos, tq, targ)
        p.resetBasePositionAndOrientation(self.pole, init_pos, init_orn)                        # reset_object :335
        for link_id in range(-1, p.getNumJoints(self.pole)):
            p.changeDynamics(self.pole, link_id, linearDamping=0.0, angularDamping=0.0)         # :338-345
        if ball:                                                                                # :350-353
            p.resetBasePositionAndOrientation(self.ball, self.ball_pos, [0, 0, 0, 1])           # reset_ball :325-326
            p.applyExternalTorque(self.ball, -1, [BALL_TORQUE[0] * 0.001, BALL_TORQUE[1] * 0.001, 0.0], flags=p.LINK_FRAME)   # :393-401
        else:
            fpos = np.array(init_pos) + np.array([BAL_PUSH_AT[0] * BAL_BASE_W / 2, BAL_PUSH_AT[1] * BAL_BASE_W / 2, 0.0])
            p.applyExternalForce(self.pole, -1, [0.0, 0.0, -0.1], list(fpos), flags=p.WORLD_FRAME)  # :360-378: consumed by the next stepSimulation

    def control(self, twist):
        wq = _quat_from_euler(*BAL_WORKFRAME[1])
        tw = np.concatenate([_rotate(wq, twist[:3]), _rotate(wq, twist[3:])])                   # workvel_to_worldvel; the limits do not bind here
        q, _ = self.joints()
        self.motors_velocity(np.linalg.inv(self.jacobian_tcp(q)) @ tw)                          # base_robot_arm.py:300-322

    def record(self):
        p = self.p
        pos, orn = p.getBasePositionAndOrientation(self.pole)                                   # the root link's INERTIAL frame (A21)
        lv, av = p.getBaseVelocity(self.pole)
        R = np.array(p.getMatrixFromQuaternion(orn)).reshape(3, 3)
        pivot_b = np.array(pos) + R @ np.array([0.0, 0.0, -BAL_BASE_H / 2 + BAL_EMBED])         # childFramePosition: in the child's inertial frame
        tcp = np.array(p.getLinkState(self.robot, self.tcp, computeForwardKinematics=True)[0])  # the TCP link's inertial frame = parentFramePosition's origin
        out = dict(pole_pos=np.array(pos), pole_rot=R.reshape(9), pole_linvel=np.array(lv), pole_angvel=np.array(av), q=self.joints()[0],
                   gap=tcp - pivot_b)
        if self.ball is not None:
            bp, _ = p.getBasePositionAndOrientation(self.ball)
            blv, bav = p.getBaseVelocity(self.ball)
            out.update(ball_pos=np.array(bp), ball_linvel=np.array(blv), ball_angvel=np.array(bav))
        return out


ROLL_REST = [0.16682, -2.23156, -1.66642, -0.81399, 1.57315, 1.74001]           # object_roll/rest_poses.py (ur5, flat)
ROLL_R, ROLL_EMBED = 0.0025, 0.0025                                             # object_roll_env.py:161; embed inside reset_task's range (:188-189) - at the
                                                                                # default 1.5 mm the tip's collision core (its lower face 1.75 mm above the TCP) stays 0.23 mm clear of this marble
ROLL_WORKFRAME = ([0.65, 0.0, 2 * ROLL_R - ROLL_EMBED], [-math.pi, 0.0, math.pi / 2])   # :70-71
ROLL_VEL = [0.004, 0.002, 0.0, 0.0, 0.0, 0.0]                                   # work-frame twist (movement_mode "xy", inside the action range :113-135)


class OracleRoll:
    """The oracle's object_roll env driven tick by tick (every randomisation off: its reset is deterministic up to the goal, which does not
    enter the dynamics)."""
    name = "oracle"

    def __init__(self, assets=None):
        from oracle.ref_env import OracleObjectRollEnv
        self.env = OracleObjectRollEnv(seed=1, env_modes=dict(rand_init_obj_pos=False, rand_obj_size=False, rand_embed_dist=False))
        self.env.embed_dist = ROLL_EMBED
        self.env.reset()

    def control(self, twist):
        self.env._tcp_velocity_control(np.array(twist, dtype=np.float64))

    def tick(self):
        self.env._step_sim()

    def record(self):
        e = self.env
        ids = [int(e.scene.contact_ids[k]) for k in range(8)][: int(e.scene.n_contacts)]
        tip = any(i >= 8 for i in ids)
        return dict(ball_pos=np.array(e.ball.pos[:]), ball_linvel=np.array(e.ball.linvel[:]), ball_angvel=np.array(e.ball.angvel[:]),
                    table_contact=np.array(int(any(i < 8 for i in ids))), tip_contact=np.array(int(tip)),
                    tip_distance=np.array(float(e.scene.tip_depth) if tip else 0.0), q=np.array(e.arm.q))


class PyBulletRoll(PyBulletBackend):
    """object_roll-v0's world in raw pybullet: UR5 + flat TacTip with its tip core on (soft contact 10 / 100, friction 10), the marble on the
    table with the env's changeDynamics."""

    def __init__(self, assets):
        import pybullet as p
        self.p = p
        if not assets or not os.path.isdir(assets):
            raise SystemExit("--assets must point at the reference's `tactile_gym/assets` directory")
        self.assets = assets
        p.connect(p.DIRECT)
        p.setGravity(0, 0, -9.81)
        p.setPhysicsEngineParameter(fixedTimeStep=SIM_DT, numSolverIterations=SOLVER_ITERS, enableConeFriction=1, contactBreakingThreshold=0.0001)
        self.plane = p.loadURDF(os.path.join(assets, "shared_assets/environment_objects/plane/plane.urdf"), [0, 0, -0.625])
        self.table = p.loadURDF(os.path.join(assets, "shared_assets/environment_objects/table/table.urdf"), [0.50, 0.00, -0.625], [0.0, 0.0, 0.0, 1.0])
        self.robot = p.loadURDF(os.path.join(assets, "robot_assets/ur5/tactip/ur5_with_flat_tactip.urdf"), [0, 0, 0], [0, 0, 0, 1], useFixedBase=True)
        self.n_all = p.getNumJoints(self.robot)
        info = [p.getJointInfo(self.robot, i) for i in range(self.n_all)]
        self.link = {inf[12].decode(): i for i, inf in enumerate(info)}
        self.ctrl = [i for i, inf in enumerate(info) if inf[2] != p.JOINT_FIXED]
        self.tcp, self.body, self.tip = self.link["tcp_link"], self.link["tactip_body_link"], self.link["tactip_tip_link"]
        for i in range(self.n_all):
            p.changeDynamics(self.robot, i, linearDamping=0.04, angularDamping=0.04)
            p.changeDynamics(self.robot, i, jointDamping=0.01)
        p.setCollisionFilterGroupMask(self.robot, self.body, 0, 0)                              # tactile_sensor.py:46-57 (core "fixed": the tip stays on)
        p.changeDynamics(self.robot, self.tip, contactDamping=100, contactStiffness=10.0)       # reset_tip :320-332, t_s_dynamics of object_roll_env.py:57
        p.changeDynamics(self.robot, self.tip, lateralFriction=10.0)
        pos = [0.65, 0.0, ROLL_R]                                                               # setup_object :161-166
        self.marble = p.loadURDF(os.path.join(assets, "rl_env_assets/nonprehensile_manipulation/object_roll/sphere/sphere.urdf"), pos, [0, 0, 0, 1])
        self.edge = None
        self.reset_joints(ROLL_REST)                                                            # Robot.reset to the work-frame origin (robot.py:114-125)
        wq = _quat_from_euler(*ROLL_WORKFRAME[1])
        tpos, tq = np.array(ROLL_WORKFRAME[0]), _quat_mul(wq, _quat_from_euler(0.0, 0.0, 0.0))
        targ = np.array(p.calculateInverseKinematics(self.robot, self.tcp, list(tpos), list(tq), restPoses=list(ROLL_REST), maxNumIterations=100,
                                                     residualThreshold=1e-8))[: len(self.ctrl)]
        self.motors_position(targ, MAX_FORCE)
        _blocking_move(self, tpos, tq, targ)
        p.resetBasePositionAndOrientation(self.marble, pos, [0, 0, 0, 1])                       # reset_object :219
        p.changeDynamics(self.marble, -1, lateralFriction=10.0, spinningFriction=0.0, rollingFriction=0.0, restitution=0.0, frictionAnchor=0,
                         collisionMargin=0.000001)                                              # :228-237

    def control(self, twist):
        wq = _quat_from_euler(*ROLL_WORKFRAME[1])
        tw = np.concatenate([_rotate(wq, twist[:3]), _rotate(wq, twist[3:])])
        q, _ = self.joints()
        self.motors_velocity(np.linalg.inv(self.jacobian_tcp(q)) @ tw)

    def record(self):
        p = self.p
        pos, _ = p.getBasePositionAndOrientation(self.marble)
        lv, av = p.getBaseVelocity(self.marble)
        table = p.getContactPoints(self.marble, self.table)
        tip = [c for c in p.getContactPoints(self.robot, self.marble) if c[3] == self.tip]
        deep = min(tip, key=lambda c: c[8]) if tip else None
        return dict(ball_pos=np.array(pos), ball_linvel=np.array(lv), ball_angvel=np.array(av), table_contact=np.array(int(bool(table))),
                    tip_contact=np.array(int(bool(tip))), tip_distance=np.array(deep[8] if deep else 0.0), q=self.joints()[0])


def _blocking_move(b, tpos, tq, targ_j, max_steps=1000):
    """Robot.blocking_move(max_steps=1000, constant_vel=0.001) (robot.py:188-260) on a backend; returns the ticks used and the joint path."""
    cv, used, qs = 0.001, 0, []
    for _ in range(max_steps):
        tcp = b.tcp_state()
        cur_j, cur_jv = b.joints()
        diff = targ_j - cur_j
        nrm = np.linalg.norm(diff)
        step_j = cur_j + (diff / nrm if nrm > 0 else np.zeros_like(cur_j)) * cv
        if np.all(np.abs(diff) < cv):
            cv /= 2
        b.motors_position(step_j, None if b.name == "pybullet" else 100000.0)                   # no `forces`: PyBullet's default [A11]
        b.tick()
        used += 1
        qs.append(b.joints()[0])
        pos_err = np.sum(np.abs(tpos - tcp[:3]))
        orn_err = math.acos(float(np.clip(2 * np.inner(tq, tcp[3:7]) ** 2 - 1, -1, 1)))
        if pos_err < 2e-4 and orn_err < 1e-3 and np.sum(np.abs(cur_jv)) < 0.1:
            break
    return used, qs


# ----------------------------------------------------------------------------------------------------------------- scenarios
def _quat_from_euler(r, p_, y):      # PyBullet's getQuaternionFromEuler (x, y, z, w)
    cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p_ / 2), math.sin(p_ / 2), math.cos(y / 2), math.sin(y / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def _rotate(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ np.asarray(v, dtype=np.float64)


def scenario_arm_statics(b):
    out = {"poses": np.array(POSES)}
    out["gravity_torque"] = np.array([b.inverse_dynamics(q) for q in POSES])
    out["mass_matrix"] = np.array([b.mass_matrix(q) for q in POSES])
    out["jacobian_tcp"] = np.array([b.jacobian_tcp(q) for q in POSES])
    return out


def scenario_arm_velocity(b, ticks=48):
    b.reset_joints(UR5_REST)
    b.motors_velocity(JOINT_VEL)
    qs, qds, tcps = [], [], []
    for _ in range(ticks):
        b.tick()
        q, qd = b.joints()
        qs.append(q), qds.append(qd), tcps.append(b.tcp_state())
    return {"rest": np.array(UR5_REST), "qd_des": np.array(JOINT_VEL), "q": np.array(qs), "qd": np.array(qds), "tcp": np.array(tcps)}


def scenario_reset_move(b, max_steps=1000):
    """Robot.reset to edge_follow's start pose: work-frame (0, 0, embed), rpy 0 (edge_follow_env.py:301-309)."""
    b.reset_joints(UR5_REST)
    wq = _quat_from_euler(*WORKFRAME[1])
    tpos = np.array(WORKFRAME[0]) + _rotate(wq, [0.0, 0.0, EMBED])                              # workframe_to_worldframe
    tq = _quat_mul(wq, _quat_from_euler(0.0, 0.0, 0.0))
    targ_j = b.ik_tcp(tpos, tq)
    b.motors_position(targ_j, MAX_FORCE)                                                        # tcp_direct_workframe_move :211-220
    used, qs = _blocking_move(b, tpos, tq, targ_j, max_steps)
    return {"target_pos": tpos, "target_quat": tq, "ik": targ_j, "ticks": np.array(used), "q_final": b.joints()[0], "q_path": np.array(qs)}


def scenario_tactile_depth(b, size=128):
    out = scenario_reset_move(b)                                                                # the sensor pressed EMBED into the edge's top face
    return {"q": out["q_final"], "edge_pos": np.array(EDGE_POS), "edge_ang": np.array(EDGE_ANG), "depth": b.depth_of_edge(size)}


def scenario_push_contacts(b, steps=10):
    keys = ("cube_pos", "cube_rot", "cube_linvel", "cube_angvel", "n_table", "n_tip", "tip_contact", "tip_normal", "tip_distance", "q")
    rec = {k: [] for k in keys}
    for _ in range(steps):
        b.control(PUSH_VEL)
        for _ in range(24):
            b.tick()
            r = b.record()
            for k in keys:
                rec[k].append(r[k])
    return dict({k: np.array(v) for k, v in rec.items()}, twist=np.array(PUSH_VEL), workframe_pos=np.array(PUSH_WORKFRAME[0]))


def scenario_balance_constraint(b, steps=10):
    keys = ("pole_pos", "pole_rot", "pole_linvel", "pole_angvel", "q", "gap")
    rec = {k: [] for k in keys}
    for _ in range(steps):
        b.control(BAL_VEL)
        for _ in range(12):                                                                     # _velocity_action_repeat = floor((1/20) / (1/240)), :33-35
            b.tick()
            r = b.record()
            for k in keys:
                rec[k].append(r[k])
    return dict({k: np.array(v) for k, v in rec.items()}, twist=np.array(BAL_VEL), gravity=np.array(BAL_GRAVITY), embed=np.array(BAL_EMBED))


def scenario_ball_on_plate(b, steps=10):      # (by 20 steps the plate - balanced on a point under a ball five times its mass - has tipped 20 degrees: past 10 the comparison would measure the instability)
    keys = ("pole_pos", "pole_rot", "q", "gap", "ball_pos", "ball_linvel", "ball_angvel")
    rec = {k: [] for k in keys}
    for _ in range(steps):
        b.control(BAL_VEL)
        for _ in range(12):
            b.tick()
            r = b.record()
            for k in keys:
                rec[k].append(r[k])
    return dict({k: np.array(v) for k, v in rec.items()}, twist=np.array(BAL_VEL), gravity=np.array(BAL_GRAVITY), embed=np.array(BAL_EMBED),
                ball_torque=np.array(BALL_TORQUE))


def scenario_roll_contacts(b, steps=5):
    keys = ("ball_pos", "ball_linvel", "ball_angvel", "table_contact", "tip_contact", "tip_distance", "q")
    rec = {k: [] for k in keys}
    for _ in range(steps):
        b.control(ROLL_VEL)
        for _ in range(24):                                                                     # floor((1/10) / (1/240)), object_roll_env.py:34-36
            b.tick()
            r = b.record()
            for k in keys:
                rec[k].append(r[k])
    return dict({k: np.array(v) for k, v in rec.items()}, twist=np.array(ROLL_VEL), radius=np.array(ROLL_R), embed=np.array(ROLL_EMBED))


SCENARIOS = {"arm_statics": scenario_arm_statics, "arm_velocity": scenario_arm_velocity, "reset_move": scenario_reset_move,
             "tactile_depth": scenario_tactile_depth, "push_contacts": scenario_push_contacts, "balance_constraint": scenario_balance_constraint,
             "ball_on_plate": scenario_ball_on_plate, "push_manifold": scenario_push_contacts, "roll_contacts": scenario_roll_contacts}
WORLDS = {"push_contacts": {"oracle": OraclePush, "pybullet": PyBulletPush},          # scenarios with a world of their own
          "push_manifold": {"oracle": lambda a: OraclePush(a, narrowphase="gjk_manifold"), "pybullet": PyBulletPush},
          "balance_constraint": {"oracle": OracleBalance, "pybullet": PyBulletBalance}, "roll_contacts": {"oracle": OracleRoll, "pybullet": PyBulletRoll},
          "ball_on_plate": {"oracle": lambda a: OracleBalance(a, ball=True), "pybullet": lambda a: PyBulletBalance(a, ball=True)}}


def engine_residual_threshold(backend, oracle_threshold):
    """btContactSolverInfo::m_leastSquaresResidualThreshold of the world the scenario ran in (PARITY_ASSUMPTIONS A7b).  PyBullet: what
    getPhysicsEngineParameters() reports - the reference never sets it (base_tactile_env.py:127-130), so this IS the answer to A7b; NaN when
    this build of PyBullet does not report the field.  Oracle backend: the value it was run with."""
    if backend != "pybullet":
        return float(oracle_threshold)
    try:
        import pybullet as p
        return float(p.getPhysicsEngineParameters().get("solverResidualThreshold", float("nan")))
    except Exception:  # noqa: BLE001
        return float("nan")


def run(backend, out_dir, assets=None, scenarios=None, residual_threshold=0.0):
    """residual_threshold (oracle backend only): the solver's exit threshold the oracle runs with - 0 (Bullet's library default: exit at a fixed
    point only) or e.g. 1e-7 (what PyBullet's server is believed to install); tests/test_pybullet_golden.py replays a PyBullet file under both
    and names the one that matches.  Every file records `solver_residual_threshold` (see engine_residual_threshold)."""
    os.makedirs(out_dir, exist_ok=True)
    written = []
    if backend == "oracle":
        from oracle import minibullet as mb
        mb.set_solver_residual_threshold(residual_threshold)
    try:
        for name in scenarios or SCENARIOS:
            b = WORLDS.get(name, {"oracle": OracleBackend, "pybullet": PyBulletBackend})[backend](assets)   # a fresh world per scenario
            data = SCENARIOS[name](b)
            path = os.path.join(out_dir, f"pybullet_{name}.npz")
            np.savez_compressed(path, backend=np.array(backend), solver_residual_threshold=np.array(engine_residual_threshold(backend, residual_threshold)),
                                **data)
            written.append(path)
    finally:
        if backend == "oracle":
            mb.set_solver_residual_threshold(0.0)
    return written


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--backend", choices=["pybullet", "oracle"], default="pybullet")
    ap.add_argument("--assets", default=os.environ.get("TG_PYBULLET_ASSETS"), help="the reference's tactile_gym/assets directory (pybullet backend)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--scenarios", nargs="*", choices=sorted(SCENARIOS))
    ap.add_argument("--residual-threshold", type=float, default=0.0, help="oracle backend: the solver's exit threshold (PARITY A7b); PyBullet's own is recorded, not set")
    a = ap.parse_args()
    for pth in run(a.backend, a.out, a.assets, a.scenarios, a.residual_threshold):
        print("wrote", pth)
    if a.backend == "pybullet":
        print("PyBullet reports solverResidualThreshold =", engine_residual_threshold("pybullet", 0.0), "(PARITY_ASSUMPTIONS A7b: the value tg_config.solver_residual_threshold should take)")


if __name__ == "__main__":
    main()
