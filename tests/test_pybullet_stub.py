"""The PyBullet backends of tools/pybullet_probe.py, executed against a STUB of the pybullet module (this container has no PyBullet).

The stub is not a physics engine.  It is the part of PyBullet's Python API the kit touches, with the parameter names of the PyBullet Quickstart
Guide as real Python signatures - a keyword the API does not have, a function that does not exist, a missing positional argument raise here
instead of on the first box that has PyBullet - and just enough state (joint angles that follow their motors, link names from the robots'
URDFs as extracted into tests/golden/urdf_facts.json, base poses that remember their last reset) for every scenario script to run to its end.
What it checks: all nine scenarios run through the PyBullet backend code path and write files with exactly the fields and shapes the oracle
backend writes, i.e. tests/test_pybullet_golden.py's comparison would run on them.  What it cannot check: anything PyBullet computes."""
import json
import math
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
FACTS = json.load(open(os.path.join(ROOT, "tests", "golden", "urdf_facts.json")))["robots"]
ROBOT_OF = {"ur5_with_standard_tactip.urdf": "ur5_standard_tactip", "ur5_with_right_angle_tactip.urdf": "ur5_right_angle_tactip",
            "ur5_with_flat_tactip.urdf": "ur5_flat_tactip"}


def make_stub():
    p = types.ModuleType("pybullet")
    p.DIRECT, p.GUI = 2, 1
    p.JOINT_REVOLUTE, p.JOINT_PRISMATIC, p.JOINT_FIXED, p.JOINT_POINT2POINT = 0, 1, 4, 5
    p.POSITION_CONTROL, p.VELOCITY_CONTROL, p.TORQUE_CONTROL = 2, 0, 1
    p.WORLD_FRAME, p.LINK_FRAME = 2, 1
    p.ER_BULLET_HARDWARE_OPENGL, p.ER_TINY_RENDERER, p.ER_SEGMENTATION_MASK_OBJECT_AND_LINKINDEX = 131072, 65536, 1
    S = {"bodies": [], "dt": 1.0 / 240.0, "ik_target": None, "calls": set()}
    p._state = S

    def body(uid):
        assert isinstance(uid, int) and 0 <= uid < len(S["bodies"]), f"unknown body id {uid!r}"
        return S["bodies"][uid]

    def api(fn):
        def wrapped(*a, **k):
            S["calls"].add(fn.__name__)
            return fn(*a, **k)
        wrapped.__name__ = fn.__name__
        setattr(p, fn.__name__, wrapped)
        return wrapped

    @api
    def connect(method, key=None, options=""):
        assert method in (p.DIRECT, p.GUI)
        return 0

    @api
    def setGravity(gravX, gravY, gravZ, physicsClientId=0):
        S["gravity"] = (gravX, gravY, gravZ)

    @api
    def setPhysicsEngineParameter(fixedTimeStep=None, numSolverIterations=None, useSplitImpulse=None, splitImpulsePenetrationThreshold=None,
                                  numSubSteps=None, collisionFilterMode=None, contactBreakingThreshold=None, maxNumCmdPer1ms=None, enableFileCaching=None,
                                  restitutionVelocityThreshold=None, erp=None, contactERP=None, frictionERP=None, enableConeFriction=None,
                                  deterministicOverlappingPairs=None, solverResidualThreshold=None, physicsClientId=0):
        if fixedTimeStep is not None:
            S["dt"] = fixedTimeStep
        if solverResidualThreshold is not None:
            S["solver_residual_threshold"] = solverResidualThreshold

    @api
    def getPhysicsEngineParameters(physicsClientId=0):
        # (the Quickstart Guide lists solverResidualThreshold among the returned fields; 1e-7 is the default it documents for the setter)
        return {"fixedTimeStep": S.get("dt", 1.0 / 240.0), "numSolverIterations": 50, "solverResidualThreshold": S.get("solver_residual_threshold", 1e-7)}

    @api
    def loadURDF(fileName, basePosition=(0, 0, 0), baseOrientation=(0, 0, 0, 1), useMaximalCoordinates=0, useFixedBase=0, flags=0, globalScaling=1.0,
                 physicsClientId=0):
        assert fileName.endswith(".urdf") and len(basePosition) == 3 and len(baseOrientation) == 4
        name = os.path.basename(fileName)
        joints = FACTS[ROBOT_OF[name]]["joints"] if name in ROBOT_OF else []
        n = len(joints)
        S["bodies"].append({"file": name, "joints": joints, "q": [0.0] * n, "qd": [0.0] * n, "mode": [None] * n, "target": [0.0] * n,
                            "pos": tuple(basePosition), "orn": tuple(baseOrientation), "v": (0.0, 0.0, 0.0), "w": (0.0, 0.0, 0.0)})
        return len(S["bodies"]) - 1

    @api
    def getNumJoints(bodyUniqueId, physicsClientId=0):
        return len(body(bodyUniqueId)["joints"])

    @api
    def getJointInfo(bodyUniqueId, jointIndex, physicsClientId=0):
        j = body(bodyUniqueId)["joints"][jointIndex]
        jt = {"fixed": p.JOINT_FIXED, "revolute": p.JOINT_REVOLUTE, "prismatic": p.JOINT_PRISMATIC}[j["type"]]
        return (jointIndex, j["name"].encode(), jt, -1, -1, 0, 0.0, 0.0, -1.0, 1.0, 0.0, 0.0, j["child"].encode(), (0.0, 0.0, 1.0), (0.0, 0.0, 0.0),
                (0.0, 0.0, 0.0, 1.0), jointIndex - 1)

    @api
    def changeDynamics(bodyUniqueId, linkIndex, mass=None, lateralFriction=None, spinningFriction=None, rollingFriction=None, restitution=None,
                       linearDamping=None, angularDamping=None, contactStiffness=None, contactDamping=None, frictionAnchor=None, localInertiaDiagonal=None,
                       ccdSweptSphereRadius=None, contactProcessingThreshold=None, activationState=None, jointDamping=None, anisotropicFriction=None,
                       maxJointVelocity=None, collisionMargin=None, jointLowerLimit=None, jointUpperLimit=None, jointLimitForce=None, physicsClientId=0):
        assert -1 <= linkIndex < max(1, len(body(bodyUniqueId)["joints"]))

    @api
    def setCollisionFilterGroupMask(bodyUniqueId, linkIndexA, collisionFilterGroup, collisionFilterMask, physicsClientId=0):
        assert -1 <= linkIndexA < len(body(bodyUniqueId)["joints"])

    @api
    def resetJointState(bodyUniqueId, jointIndex, targetValue, targetVelocity=0.0, physicsClientId=0):
        b = body(bodyUniqueId)
        b["q"][jointIndex], b["qd"][jointIndex] = float(targetValue), float(targetVelocity)

    @api
    def setJointMotorControlArray(bodyUniqueId, jointIndices, controlMode, targetPositions=None, targetVelocities=None, forces=None, positionGains=None,
                                  velocityGains=None, physicsClientId=0):
        b = body(bodyUniqueId)
        n = len(jointIndices)
        for name, arr in (("targetPositions", targetPositions), ("targetVelocities", targetVelocities), ("forces", forces),
                          ("positionGains", positionGains), ("velocityGains", velocityGains)):
            assert arr is None or len(arr) == n, name
        for k, j in enumerate(jointIndices):
            if controlMode == p.POSITION_CONTROL:
                b["mode"][j], b["target"][j] = "pos", float(targetPositions[k])
            elif controlMode == p.VELOCITY_CONTROL:
                b["mode"][j], b["target"][j] = "vel", float(targetVelocities[k])
            else:
                assert controlMode == p.TORQUE_CONTROL and forces is not None

    @api
    def getJointStates(bodyUniqueId, jointIndices, physicsClientId=0):
        b = body(bodyUniqueId)
        return [(b["q"][j], b["qd"][j], (0.0,) * 6, 0.0) for j in jointIndices]

    @api
    def stepSimulation(physicsClientId=0):
        for b in S["bodies"]:
            for j in range(len(b["q"])):
                if b["mode"][j] == "pos":
                    b["qd"][j] = 0.0
                    b["q"][j] = b["target"][j]
                elif b["mode"][j] == "vel":
                    b["qd"][j] = b["target"][j]
                    b["q"][j] += S["dt"] * b["target"][j]

    def movable(b):
        return [i for i, j in enumerate(b["joints"]) if j["type"] != "fixed"]

    @api
    def calculateInverseDynamics(bodyUniqueId, objPositions, objVelocities, objAccelerations, physicsClientId=0):
        n = len(movable(body(bodyUniqueId)))
        assert len(objPositions) == len(objVelocities) == len(objAccelerations) == n
        return tuple(0.0 for _ in range(n))

    @api
    def calculateMassMatrix(bodyUniqueId, objPositions, physicsClientId=0):
        n = len(movable(body(bodyUniqueId)))
        assert len(objPositions) == n
        return tuple(tuple(1.0 if r == c else 0.0 for c in range(n)) for r in range(n))

    @api
    def calculateJacobian(bodyUniqueId, linkIndex, localPosition, objPositions, objVelocities, objAccelerations, physicsClientId=0):
        n = len(movable(body(bodyUniqueId)))
        assert len(localPosition) == 3 and len(objPositions) == len(objVelocities) == len(objAccelerations) == n == 6
        eye = [[1.0 if r == c else 0.0 for c in range(n)] for r in range(6)]
        return tuple(map(tuple, eye[:3])), tuple(map(tuple, eye[3:]))

    @api
    def calculateInverseKinematics(bodyUniqueId, endEffectorLinkIndex, targetPosition, targetOrientation=None, lowerLimits=None, upperLimits=None,
                                   jointRanges=None, restPoses=None, jointDamping=None, solver=0, currentPosition=None, maxNumIterations=20,
                                   residualThreshold=1e-4, physicsClientId=0):
        b = body(bodyUniqueId)
        S["ik_target"] = (tuple(targetPosition), tuple(targetOrientation))
        return tuple(restPoses) if restPoses is not None else tuple(b["q"][j] for j in movable(b))

    @api
    def getLinkState(bodyUniqueId, linkIndex, computeLinkVelocity=0, computeForwardKinematics=0, physicsClientId=0):
        assert 0 <= linkIndex < len(body(bodyUniqueId)["joints"])
        pos, orn = S["ik_target"] if S["ik_target"] else ((0.6, 0.0, 0.1), (0.0, 0.0, 0.0, 1.0))
        out = (pos, orn, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), pos, orn)
        return out + ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)) if computeLinkVelocity else out

    @api
    def getQuaternionFromEuler(eulerAngles, physicsClientId=0):
        r, pt, y = eulerAngles
        cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(pt / 2), math.sin(pt / 2), math.cos(y / 2), math.sin(y / 2)
        return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)

    @api
    def getMatrixFromQuaternion(quaternion, physicsClientId=0):
        x, y, z, w = quaternion
        return (1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y))

    @api
    def multiplyTransforms(positionA, orientationA, positionB, orientationB, physicsClientId=0):
        R = np.array(getMatrixFromQuaternion(orientationA)).reshape(3, 3)
        return tuple(np.array(positionA) + R @ np.array(positionB)), tuple(orientationA)

    @api
    def computeViewMatrix(cameraEyePosition, cameraTargetPosition, cameraUpVector, physicsClientId=0):
        assert len(cameraEyePosition) == len(cameraTargetPosition) == len(cameraUpVector) == 3
        return tuple(float(i % 5 == 0) for i in range(16))

    @api
    def computeProjectionMatrixFOV(fov, aspect, nearVal, farVal, physicsClientId=0):
        return tuple(float(i % 5 == 0) for i in range(16))

    @api
    def getCameraImage(width, height, viewMatrix=None, projectionMatrix=None, lightDirection=None, lightColor=None, lightDistance=None, shadow=None,
                       lightAmbientCoeff=None, lightDiffuseCoeff=None, lightSpecularCoeff=None, renderer=None, flags=None, physicsClientId=0):
        assert len(viewMatrix) == 16 and len(projectionMatrix) == 16
        return (width, height, np.zeros((height, width, 4), np.uint8), np.full(width * height, 0.5, np.float32), np.zeros(width * height, np.int32))

    @api
    def getBasePositionAndOrientation(bodyUniqueId, physicsClientId=0):
        b = body(bodyUniqueId)
        return b["pos"], b["orn"]

    @api
    def resetBasePositionAndOrientation(bodyUniqueId, posObj, ornObj, physicsClientId=0):
        assert len(posObj) == 3 and len(ornObj) == 4
        b = body(bodyUniqueId)
        b["pos"], b["orn"] = tuple(posObj), tuple(ornObj)

    @api
    def getBaseVelocity(bodyUniqueId, physicsClientId=0):
        b = body(bodyUniqueId)
        return b["v"], b["w"]

    @api
    def getContactPoints(bodyA=-1, bodyB=-1, linkIndexA=-2, linkIndexB=-2, physicsClientId=0):
        body(bodyA), body(bodyB)
        return ()

    @api
    def createConstraint(parentBodyUniqueId, parentLinkIndex, childBodyUniqueId, childLinkIndex, jointType, jointAxis, parentFramePosition, childFramePosition,
                         parentFrameOrientation=(0, 0, 0, 1), childFrameOrientation=(0, 0, 0, 1), physicsClientId=0):
        body(parentBodyUniqueId), body(childBodyUniqueId)
        assert jointType in (p.JOINT_POINT2POINT, p.JOINT_FIXED) and len(jointAxis) == len(parentFramePosition) == len(childFramePosition) == 3
        return 7

    @api
    def changeConstraint(userConstraintUniqueId, jointChildPivot=None, jointChildFrameOrientation=None, maxForce=None, gearRatio=None, gearAuxLink=None,
                         relativePositionTarget=None, erp=None, physicsClientId=0):
        assert userConstraintUniqueId == 7

    @api
    def applyExternalForce(objectUniqueId, linkIndex, forceObj, posObj, flags, physicsClientId=0):
        body(objectUniqueId)
        assert len(forceObj) == len(posObj) == 3 and flags in (p.WORLD_FRAME, p.LINK_FRAME)

    @api
    def applyExternalTorque(objectUniqueId, linkIndex, torqueObj, flags, physicsClientId=0):
        body(objectUniqueId)
        assert len(torqueObj) == 3 and flags in (p.WORLD_FRAME, p.LINK_FRAME)

    return p


@pytest.fixture()
def stub(monkeypatch):
    p = make_stub()
    monkeypatch.setitem(sys.modules, "pybullet", p)
    return p


def test_every_scenario_runs_through_the_pybullet_backends_and_writes_the_oracle_backends_format(stub, tmp_path):
    import pybullet_probe as probe
    assets = tmp_path / "assets"
    assets.mkdir()
    mine = probe.run("pybullet", str(tmp_path / "pb"), assets=str(assets))
    ref = probe.run("oracle", str(tmp_path / "oracle"))
    assert sorted(map(os.path.basename, mine)) == sorted(map(os.path.basename, ref)) == sorted(f"pybullet_{s}.npz" for s in probe.SCENARIOS)
    for a, b in zip(sorted(mine), sorted(ref)):
        da, db = np.load(a), np.load(b)
        assert str(da["backend"]) == "pybullet" and str(db["backend"]) == "oracle"
        assert sorted(da.files) == sorted(db.files), os.path.basename(a)
        for f in da.files:
            if f in ("backend", "ticks", "q_path"):            # the blocking move's tick count (and with it the path's length) is the engine's to say
                continue
            assert da[f].shape == db[f].shape, (os.path.basename(a), f, da[f].shape, db[f].shape)
            assert np.all(np.isfinite(np.asarray(da[f], dtype=np.float64))), (os.path.basename(a), f)
    used = stub._state["calls"]
    for name in ("createConstraint", "changeConstraint", "applyExternalForce", "applyExternalTorque", "getCameraImage", "calculateInverseKinematics",
                 "setCollisionFilterGroupMask", "getContactPoints", "calculateJacobian", "calculateMassMatrix", "calculateInverseDynamics"):
        assert name in used, f"{name} never called: a scenario lost the call the reference makes"


def test_the_stub_rejects_what_pybullet_would_reject(stub):
    """The point of the stub: a keyword PyBullet does not have, or a call it does not know, fails here."""
    import pybullet as p
    r = p.loadURDF("robot_assets/ur5/tactip/ur5_with_standard_tactip.urdf", [0, 0, 0], [0, 0, 0, 1], useFixedBase=True)
    with pytest.raises(TypeError):
        p.changeConstraint(7, jointChildPivotPosition=[0, 0, 0])
    with pytest.raises(TypeError):
        p.setJointMotorControlArray(r, [1], p.VELOCITY_CONTROL, targetVelocity=[0.0])
    with pytest.raises(AttributeError):
        p.getJointStateArray
    assert p.getJointInfo(r, 10)[12] == b"tcp_link" and p.getJointInfo(r, 1)[2] == p.JOINT_REVOLUTE


def test_bench_cpu_baseline_through_pybullet_runs_against_the_stub(stub, tmp_path, monkeypatch):
    """bench.py's `cpu_baseline.pybullet_reference` (the raw engine timed beside the oracle when `import pybullet` works and
    TG_PYBULLET_ASSETS is set) - the same code path, against the stub: it returns a rate, not the "probe failed" string."""
    sys.path.insert(0, ROOT)
    import bench
    assets = tmp_path / "assets"
    assets.mkdir()
    monkeypatch.delenv("TG_PYBULLET_ASSETS", raising=False)
    assert "TG_PYBULLET_ASSETS" in bench.pybullet_reference(0.05)
    monkeypatch.setenv("TG_PYBULLET_ASSETS", str(assets))
    out = bench.pybullet_reference(0.05)
    assert isinstance(out, dict) and out["value"] > 0 and out["unit"] == "env-steps/s" and out["cores"] == 1, out
