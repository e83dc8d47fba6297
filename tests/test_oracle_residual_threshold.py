"""CPU tests of the oracle's solver-residual-threshold mode (PARITY_ASSUMPTIONS A7b / A7c): Bullet's exit rule for the Gauss-Seidel loop of
stepSimulation - leave after the sweep whose largest SQUARED row velocity change (deltaImpulse / jacDiagABInv) is <= the threshold, never
before the first sweep, at the latest after numSolverIterations - which the reference never overrides (base_tactile_env.py:127-130 passes four
engine parameters, not solverResidualThreshold).  Known answers first (cases whose sweep count follows by hand), then the C loops against an
independent numpy restatement of the rule, then what the mode does to an episode.  No HIP here."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import minibullet as mb   # noqa: E402
from oracle import ref_env            # noqa: E402

DT = 1.0 / 240.0


@pytest.fixture(autouse=True)
def _restore_threshold():
    yield
    mb.set_solver_residual_threshold(0.0)


def _ur5():
    arm = mb.Arm(ref_env.load_tg("ur5_standard_tactip"))
    arm.reset_joint_states([0.3, -1.9, -1.7, -1.1, 1.5, 1.7])
    return arm


def _gravity_compensated_step(arm, iters=150):
    arm.apply_torques(arm.inverse_dynamics(arm.q, arm.qd, np.zeros(arm.n)))
    arm.step_simulation(DT, iters)
    return mb.last_sweeps()


def test_nothing_to_do_leaves_after_exactly_one_sweep():
    """Hand-checked: an arm at rest under gravity compensation whose velocity motors ask for zero velocity.  The unconstrained velocity is 0,
    every row's right-hand side is 0, every delta of the first sweep is 0: residual 0 <= threshold.  Bullet tests the residual AFTER a sweep,
    so the loop runs one sweep - not zero - whatever the threshold (0 included)."""
    for thr in (0.0, 1e-7, 1.0):
        arm = _ur5()
        arm.set_motors_velocity(np.zeros(6), 1.0, 1000.0)
        mb.set_solver_residual_threshold(thr)
        assert _gravity_compensated_step(arm) == 1
        assert np.abs(arm.qd).max() < 1e-12


def test_a_request_below_the_threshold_is_one_sweep_and_above_it_more():
    """Hand-checked on the first row visited.  Sweep 1 runs in reverse order from lambda = 0, so its first update (the last joint) is
    delta = (des - v) / Minv_ii with velocity change deltaVel = des - v exactly.  Ask ONLY the last joint for 2e-4 rad/s: that row changes its
    velocity by 2e-4, the rows visited after it by the coupling Minv_ji / Minv_jj x 2e-4 (smaller); residual <= (2e-4)^2 = 4e-8 <= 1e-7: ONE sweep.
    Ask for 1e-3: (1e-3)^2 = 1e-6 > 1e-7, the loop must go on; the UR5's iteration contracts by about a half per sweep, so it is done within a
    handful of sweeps - far from the 50+ the zero threshold needs for the same request."""
    sweeps = {}
    for req in (2e-4, 1e-3):
        for thr in (1e-7, 0.0):
            arm = _ur5()
            des = np.zeros(6)
            des[5] = req
            arm.set_motors_velocity(des, 1.0, 1000.0)
            mb.set_solver_residual_threshold(thr)
            sweeps[(req, thr)] = _gravity_compensated_step(arm)
    assert sweeps[(2e-4, 1e-7)] == 1
    assert 2 <= sweeps[(1e-3, 1e-7)] <= 8
    assert sweeps[(2e-4, 0.0)] > 20 and sweeps[(1e-3, 0.0)] > 20


def test_one_sweep_leaves_the_velocity_error_the_rule_allows():
    """What the mode means physically (the caveat PARITY A7b has carried since round 2): after ONE sweep from lambda = 0 the rows visited first
    have been disturbed by the rows visited after them, so the joint velocities are NOT the motors' targets.  The rule bounds each row's own
    CHANGE by sqrt(threshold) = 3.2e-4 rad/s, not the error left behind: a wrist row's update moves the neighbouring wrist joint by the coupling
    ratio Minv_ji / Minv_jj, which exceeds 1 on the UR5 - measured 7.8e-4 rad/s here for a request of 2.5e-4 on every joint - while the zero
    threshold lands on the targets to 1e-12."""
    des = np.full(6, 2.5e-4)
    err = {}
    for thr in (1e-7, 0.0):
        arm = _ur5()
        arm.set_motors_velocity(des, 1.0, 1000.0)
        mb.set_solver_residual_threshold(thr)
        n = _gravity_compensated_step(arm)
        err[thr] = (n, np.abs(arm.qd - des).max())
    assert err[1e-7][0] == 1 and 1e-5 < err[1e-7][1] < 3e-3
    assert err[0.0][0] > 20 and err[0.0][1] < 1e-12


def _numpy_pgs(Minv, rhs_vel, maximp, iters, thr):
    """Independent restatement of the motor solve with Bullet's exit rule: rows J = e_i, A_ii = Minv_ii, delta in impulse, residual in
    velocity (delta * A_ii), reverse order on even sweeps."""
    n = len(rhs_vel)
    lam, dv = np.zeros(n), np.zeros(n)
    for it in range(iters):
        res = 0.0
        for jj in range(n):
            i = jj if (it & 1) else n - 1 - jj
            delta = (rhs_vel[i] - dv[i]) / Minv[i, i]
            s = min(max(lam[i] + delta, -maximp), maximp)
            delta = s - lam[i]
            lam[i] = s
            dv += Minv[:, i] * delta
            res = max(res, (delta * Minv[i, i]) ** 2)
        if res <= thr:
            return dv, it + 1
    return dv, iters


@pytest.mark.parametrize("thr", [1e-7, 1e-9, 1e-5])
def test_motor_solve_equals_a_numpy_restatement_of_the_rule(thr):
    """mb_step against the rule written down again in numpy on the mass matrix the oracle reports: same sweep count on every case, same
    post-step velocities to rounding - UR5 (contraction ~0.5 per sweep) and MG400 (~0.94, the arm the rule changes most)."""
    rng = np.random.default_rng(5)
    for name, q0 in (("ur5_standard_tactip", [0.3, -1.9, -1.7, -1.1, 1.5, 1.7]), ("mg400_standard_tactip", None)):
        tg = ref_env.load_tg(name)
        for case in range(12):
            arm = mb.Arm(tg)
            q = np.array(q0) if q0 is not None else np.array([0.2, 0.5, 0.4, -0.9, 0.1, 0.5, -0.5, 0.9])[: tg.ndof]
            arm.reset_joint_states(q + rng.uniform(-0.05, 0.05, size=arm.n))
            des = rng.uniform(-0.3, 0.3, size=arm.n) * 10.0 ** rng.uniform(-3, 0)
            arm.set_motors_velocity(des, 1.0, 1000.0)
            Minv = np.linalg.inv(arm.mass_matrix(arm.q))
            mb.set_solver_residual_threshold(thr)
            n_c = _gravity_compensated_step(arm)
            # at rest with gravity compensated the unconstrained velocity is 0, so the rows ask for des itself
            dv, n_np = _numpy_pgs(Minv, des, 1000.0 * DT, 150, thr)
            assert n_c == n_np, (name, case, n_c, n_np)
            assert np.abs(arm.qd - dv).max() < 1e-12 * max(1.0, np.abs(des).max()), (name, case)
    mb.set_solver_residual_threshold(0.0)


def test_threshold_zero_is_the_old_behaviour_bit_for_bit():
    """0 must stay what it was: the loop leaves only at an exact floating-point fixed point."""
    a = ref_env.OracleEdgeFollowEnv(seed=11)
    b = ref_env.OracleEdgeFollowEnv(seed=11)
    b.solver_residual_threshold = 0.0
    a.reset(), b.reset()
    rng = np.random.default_rng(0)
    for _ in range(3):
        act = rng.uniform(-0.25, 0.25, size=2)
        a.step(act), b.step(act)
    assert np.array_equal(a.arm.q, b.arm.q) and a.sweeps_total == b.sweeps_total
    assert a.sweeps_total / a.ticks > 40          # dozens of sweeps per tick


@pytest.mark.parametrize("cls,kw,lo,hi", [
    ("OracleEdgeFollowEnv", {}, 1.0, 4.0),                                                   # UR5: a couple of sweeps per tick
    ("OracleObjectBalanceEnv", {}, 1.0, 6.0),                                                # + the three P2P rows
    ("OracleObjectPushEnv", dict(env_modes=dict(movement_mode="TyRz", control_mode="TCP_velocity_control", rand_init_orn=False,
                                                rand_obj_mass=False, traj_type="simplex", observation_mode="tactile_and_feature",
                                                reward_mode="dense", arm_type="mg400", tactile_sensor_name="digitac")), 20.0, 120.0),
])
def test_sweeps_per_tick_with_the_threshold(cls, kw, lo, hi):
    """The figure that decides what config 4 costs: with 1e-7 the contact problems leave after tens of sweeps, the arm-only ones after a few;
    with 0 they run (nearly) all 150.  Ranges, not exact values: the exact counts are compared HIP against oracle on the device suite."""
    rng = np.random.default_rng(1)
    out = {}
    for thr in (1e-7, 0.0):
        e = getattr(ref_env, cls)(seed=4, **kw)
        e.solver_residual_threshold = thr
        e.reset()
        t0, s0 = e.ticks, e.sweeps_total
        n_act = {"OracleEdgeFollowEnv": 2, "OracleObjectBalanceEnv": 2, "OracleObjectPushEnv": 2}[cls]
        for _ in range(4):
            e.step(rng.uniform(-0.25, 0.25, size=n_act))
        out[thr] = (e.sweeps_total - s0) / (e.ticks - t0)
    assert lo <= out[1e-7] <= hi, out
    assert out[0.0] > 50.0 and out[0.0] > 2.0 * out[1e-7], out


def test_the_two_modes_differ_where_a7b_says_they_do():
    """Same seed, same actions: the joint angles after a step differ between the two readings of PyBullet's default by micro-radians (a
    velocity error of up to 3e-4 rad/s for a tick or two after every change of the request) - far above the 1e-9 rad the HIP path is held to
    against either oracle mode, far below anything a tactile image shows."""
    qs = {}
    for thr in (0.0, 1e-7):
        e = ref_env.OracleEdgeFollowEnv(seed=21)
        e.solver_residual_threshold = thr
        e.reset()
        rng = np.random.default_rng(3)
        for _ in range(5):
            e.step(rng.uniform(-0.25, 0.25, size=2))
        qs[thr] = e.arm.q.copy()
    d = np.abs(qs[0.0] - qs[1e-7]).max()
    assert 1e-8 < d < 1e-4, d
    assert math.isfinite(d)
