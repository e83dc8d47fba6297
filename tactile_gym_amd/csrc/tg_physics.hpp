// tg_physics.hpp — per-env articulated-body dynamics and control, one lane per env (gfx950).
//
// Replaces what the reference delegates to PyBullet on every sim tick (tactile_gym/robots/arms/robot.py:131-141):
//   calculateInverseDynamics (base_robot_arm.py:176-178), the TORQUE_CONTROL feed-forward (:184-189) and
//   stepSimulation() (robot.py:141) with its joint-motor constraint rows (base_robot_arm.py:211-220, 325-332),
// plus, once per env step, calculateJacobian + the Jacobian inverse of tcp_velocity_control (:281-332) and, on reset,
// calculateInverseKinematics (:201-209).
//
// Formulation (deliberately different from the CPU oracle, which sums body by body):
//   * bodies welded to the same moving link are merged on the host into one (m, com, I) per link;
//   * one root->leaf sweep gives world-frame link origins, joint axes, velocities and the velocity-product
//     accelerations; one leaf->root sweep accumulates Newton-Euler bias wrenches and composite inertias about each
//     joint origin; M(q) follows as  M_ij = a_i . ( Io_j a_j + (o_j - o_i) x (a_j x h_j) )  for i above j;
//   * M is Cholesky-factored and explicitly inverted (n <= 8) because the motor rows of the projected Gauss-Seidel
//     solver need columns of M^-1 (Bullet gets the same columns from calcAccelerationDeltasMultiDof);
//   * everything is fully unrolled over compile-time link indices so per-link quantities live in VGPRs: a lane never
//     indexes a private array dynamically, and the robot constants are wave-uniform scalar loads.
//
// Why one lane per env and not one wavefront per env: the constraint system has ndof <= 8 rows and Gauss-Seidel is
// a serial chain over rows and sweeps; 64 lanes cannot shorten that chain, they can only idle.  One lane per env
// keeps all 64 lanes busy with 64 independent chains and makes every global access (SoA, env-minor) a perfectly
// coalesced 512-byte (f64) wavefront transaction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace tg {

constexpr int kMaxDof = 8;
constexpr int kMaxBodiesPerLink = 4;

// ------------------------------------------------------------------------------------------------ topology
template <int TOPO> struct Topo;
template <> struct Topo<0> {  // serial 6-DoF chain (UR5)
    static constexpr int N = 6;
    static constexpr int NP = 6;   // joints that can lie on the path from the base to the tool
    __host__ __device__ static constexpr int parent(int i) { return i - 1; }
};
template <> struct Topo<1> {  // MG400: j1 -> {j2_1 -> j3_1 -> j4_1 -> j5, j2_2 -> j3_2 -> j4_2}
    static constexpr int N = 8;
    static constexpr int NP = 5;   // j1, j2_1, j3_1, j4_1, j5; the j*_2 branch only carries the parallel linkage
    __host__ __device__ static constexpr int parent(int i) { return i == 0 ? -1 : (i == 5 ? 0 : i - 1); }
};
template <int TOPO> __host__ __device__ constexpr bool is_ancestor_or_self(int anc, int i) {
    while (i >= 0) { if (i == anc) return true; i = Topo<TOPO>::parent(i); }
    return false;
}

// ------------------------------------------------------------------------------------------------ device constants
template <typename T> struct DevRobot {
    T jpos[kMaxDof][3], jrot[kMaxDof][9], jaxis[kMaxDof][3];
    // Rj Rot(axis, q) = fkA + cos(q) fkB + sin(q) fkC  with  fkA = Rj a a^T, fkB = Rj (I - a a^T), fkC = Rj [a]x
    T fkA[kMaxDof][9], fkB[kMaxDof][9], fkC[kMaxDof][9];
    T lang[kMaxDof][6];                                          // sum_b R_b diag(I_b) R_b^T per link (angular damping), link frame
    T lmass[kMaxDof], lcom[kMaxDof][3], linert[kMaxDof][6];       // merged per-link inertia about lcom, link frame (xx,xy,xz,yy,yz,zz)
    T bmass[kMaxDof][kMaxBodiesPerLink], bcom[kMaxDof][kMaxBodiesPerLink][3], brot[kMaxDof][kMaxBodiesPerLink][9],
        binert[kMaxDof][kMaxBodiesPerLink][3];                     // individual bodies (velocity damping is per body)
    int tcp_link; T tcp_pos[3], tcp_rot[9];
    int sensor_link; T sensor_pos[3], sensor_rot[9];
    T gravity[3], lin_damp, ang_damp, joint_damp, max_force, pos_gain, vel_gain;
    T rest_q[kMaxDof];
    T trace_bound;   // >= trace(M(q)) for every q (host, build_dev_robot)
    T diag_sqrt[kMaxDof], diag_sqrt_max;   // sqrt of per-joint bounds d_i >= M_ii(q) for every q, and their maximum
    // tg_config.solver_residual_threshold (btContactSolverInfo::m_leastSquaresResidualThreshold, PARITY_ASSUMPTIONS A7b / A7c).  0: the solver
    // loops exit at last-bit convergence only and the licensed analytic fixed point may stand in for a converged solve (every kernel's default
    // path).  > 0 ("threshold mode"): every tick is a full tick and its Gauss-Seidel loop leaves after the sweep whose largest squared row
    // velocity change deltaImpulse / jacDiagABInv (a cone-friction pair counts once, with the sum of its two) is <= this - Bullet's rule,
    // evaluated per env, sweep by sweep, in the oracle's order; no licence, no composed sweeps.  Lives here because every sim_tick* takes `m`.
    T res_thr;
};

// ------------------------------------------------------------------------------------------------ small vector algebra
template <typename T> struct V3 { T x, y, z; };
template <typename T> __device__ __forceinline__ V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> __device__ __forceinline__ V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ T tsqrt(T x);
template <> __device__ __forceinline__ double tsqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float tsqrt<float>(float x) { return sqrtf(x); }
template <typename T> __device__ __forceinline__ void tsincos(T x, T* s, T* c);
template <> __device__ __forceinline__ void tsincos<double>(double x, double* s, double* c) { sincos(x, s, c); }
template <> __device__ __forceinline__ void tsincos<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <typename T> __device__ __forceinline__ T tatan2(T y, T x);
template <> __device__ __forceinline__ double tatan2<double>(double y, double x) { return atan2(y, x); }
template <> __device__ __forceinline__ float tatan2<float>(float y, float x) { return atan2f(y, x); }
template <typename T> __device__ __forceinline__ T tasin(T x);
template <> __device__ __forceinline__ double tasin<double>(double x) { return asin(x); }
template <> __device__ __forceinline__ float tasin<float>(float x) { return asinf(x); }
template <typename T> __device__ __forceinline__ T tacos(T x);
template <> __device__ __forceinline__ double tacos<double>(double x) { return acos(x); }
template <> __device__ __forceinline__ float tacos<float>(float x) { return acosf(x); }
__device__ __forceinline__ double tabs(double x) { return __builtin_fabs(x); }     // source modifier |x|, no compare + select
__device__ __forceinline__ float tabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double tmax(double a, double b) { return __builtin_fmax(a, b); }   // v_max_f64 / v_min_f64
__device__ __forceinline__ float tmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double tmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float tmin(float a, float b) { return __builtin_fminf(a, b); }
// Hardware seeds + Newton steps instead of the library's range-scaled, special-cased sqrt / division (about 8 instructions instead
// of 20-25); operands here are well-scaled physical quantities.  Results agree with the IEEE ones to 1-2 ulp.
__device__ __forceinline__ double trsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - (0.5 * x) * y * y);
    y = y * (1.5 - (0.5 * x) * y * y);
    return y;
}
__device__ __forceinline__ float trsqrt(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    y = y * (1.5f - (0.5f * x) * y * y);
    return y;
}
__device__ __forceinline__ double trcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}
__device__ __forceinline__ float trcp(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    y = y * (2.0f - x * y);
    return y;
}
template <typename T> __device__ __forceinline__ T tsqrt_fast(T x) { return x * trsqrt(tmax(x, T(1e-30))); }   // sqrt(0) = 0
template <typename T> __device__ __forceinline__ T norm(V3<T> a) { return tsqrt(dot(a, a)); }

template <typename T> struct M3 { T m[9]; };  // row major
template <typename T> __device__ __forceinline__ V3<T> mul(const M3<T>& A, V3<T> v) {
    return {A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z, A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
template <typename T> __device__ __forceinline__ V3<T> mulT(const M3<T>& A, V3<T> v) {
    return {A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z, A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z};
}
template <typename T> __device__ __forceinline__ M3<T> mul(const M3<T>& A, const M3<T>& B) {
    M3<T> C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
template <typename T> __device__ __forceinline__ M3<T> load_m3(const T* p) {
    M3<T> A;
#pragma unroll
    for (int k = 0; k < 9; ++k) A.m[k] = p[k];
    return A;
}
template <typename T> __device__ __forceinline__ V3<T> load_v3(const T* p) { return {p[0], p[1], p[2]}; }
template <typename T> __device__ __forceinline__ M3<T> axis_angle(V3<T> a, T q) {
    T s, c;
    tsincos(q, &s, &c);
    const T v = T(1) - c;
    M3<T> R;
    R.m[0] = c + a.x * a.x * v;       R.m[1] = a.x * a.y * v - a.z * s; R.m[2] = a.x * a.z * v + a.y * s;
    R.m[3] = a.y * a.x * v + a.z * s; R.m[4] = c + a.y * a.y * v;       R.m[5] = a.y * a.z * v - a.x * s;
    R.m[6] = a.z * a.x * v - a.y * s; R.m[7] = a.z * a.y * v + a.x * s; R.m[8] = c + a.z * a.z * v;
    return R;
}

// symmetric 3x3 (xx, xy, xz, yy, yz, zz)
template <typename T> struct S3 { T xx, xy, xz, yy, yz, zz; };
template <typename T> __device__ __forceinline__ V3<T> mul(const S3<T>& S, V3<T> v) {
    return {S.xx * v.x + S.xy * v.y + S.xz * v.z, S.xy * v.x + S.yy * v.y + S.yz * v.z, S.xz * v.x + S.yz * v.y + S.zz * v.z};
}
template <typename T> __device__ __forceinline__ S3<T> operator+(const S3<T>& a, const S3<T>& b) {
    return {a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz};
}
// R S R^T for symmetric S
template <typename T> __device__ __forceinline__ S3<T> rotate(const M3<T>& R, const S3<T>& S) {
    // columns of S R^T = S * (rows of R)
    const V3<T> r0{R.m[0], R.m[1], R.m[2]}, r1{R.m[3], R.m[4], R.m[5]}, r2{R.m[6], R.m[7], R.m[8]};
    const V3<T> s0 = mul(S, r0), s1 = mul(S, r1), s2 = mul(S, r2);
    return {dot(r0, s0), dot(r0, s1), dot(r0, s2), dot(r1, s1), dot(r1, s2), dot(r2, s2)};
}
// m [ (r.r) E - r r^T ]
template <typename T> __device__ __forceinline__ S3<T> point_inertia(T m, V3<T> r) {
    const T rr = dot(r, r);
    return {m * (rr - r.x * r.x), -m * r.x * r.y, -m * r.x * r.z, m * (rr - r.y * r.y), -m * r.y * r.z, m * (rr - r.z * r.z)};
}
// 2 (h.r) E - h r^T - r h^T
template <typename T> __device__ __forceinline__ S3<T> cross_inertia(V3<T> h, V3<T> r) {
    const T hr2 = T(2) * dot(h, r);
    return {hr2 - T(2) * h.x * r.x, -(h.x * r.y + r.x * h.y), -(h.x * r.z + r.x * h.z), hr2 - T(2) * h.y * r.y, -(h.y * r.z + r.y * h.z),
            hr2 - T(2) * h.z * r.z};
}

// ------------------------------------------------------------------------------------------------ kinematics
template <typename T, int TOPO> struct Kin {
    static constexpr int N = Topo<TOPO>::N;
    M3<T> R[N];
    V3<T> o[N], a[N];
};

// sin / cos of the joint angles, carried across the ticks of one env step: after q += dq the pair is advanced by the angle-addition
// formulas with a degree-7/6 Taylor pair for (sin dq, cos dq) (|dq| <= 0.02: truncation < 1e-20), ~12 instructions per joint instead
// of a full double-precision sincos (~90).  Re-anchored with the exact sincos at every env step (24 ticks: drift < 1e-15).
template <typename T, int N> struct JointTrig { T s[N], c[N]; };
template <typename T, int N> __device__ __forceinline__ void trig_init(const T (&q)[N], JointTrig<T, N>& t) {
#pragma unroll
    for (int i = 0; i < N; ++i) tsincos(q[i], &t.s[i], &t.c[i]);
}
template <typename T, int N> __device__ __forceinline__ void trig_advance(const T (&q)[N], const T (&dq)[N], JointTrig<T, N>& t) {
    T big = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) big = tmax(big, tabs(dq[i]));
    if (__any(big > T(0.02))) { trig_init<T, N>(q, t); return; }   // never in the reference's velocity range; exact fallback
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const T d = dq[i], d2 = d * d;
        const T cd = T(1) + d2 * (T(-0.5) + d2 * (T(1.0 / 24.0) + d2 * T(-1.0 / 720.0)));
        const T sd = d * (T(1) + d2 * (T(-1.0 / 6.0) + d2 * (T(1.0 / 120.0) + d2 * T(-1.0 / 5040.0))));
        const T s0 = t.s[i], c0 = t.c[i];
        t.s[i] = s0 * cd + c0 * sd;
        t.c[i] = c0 * cd - s0 * sd;
    }
}

template <typename T, int TOPO, bool TRIG = false>
__device__ __forceinline__ void forward_kinematics(const DevRobot<T>& m, const T (&q)[Topo<TOPO>::N], Kin<T, TOPO>& k,
                                                   const JointTrig<T, Topo<TOPO>::N>* trig = nullptr) {
    constexpr int N = Topo<TOPO>::N;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int p = Topo<TOPO>::parent(i);
        const V3<T> ax = load_v3(m.jaxis[i]);
        T sq, cq;
        if (TRIG) { sq = trig->s[i]; cq = trig->c[i]; }
        else tsincos(q[i], &sq, &cq);
        M3<T> Rl;   // Rj * Rot(axis, q)
#pragma unroll
        for (int e = 0; e < 9; ++e) Rl.m[e] = m.fkA[i][e] + cq * m.fkB[i][e] + sq * m.fkC[i][e];
        if (p < 0) {
            k.R[i] = Rl;
            k.o[i] = load_v3(m.jpos[i]);
        } else {
            k.R[i] = mul(k.R[p], Rl);
            k.o[i] = k.o[p] + mul(k.R[p], load_v3(m.jpos[i]));
        }
        k.a[i] = mul(k.R[i], ax);
    }
}

// World pose of a frame welded to moving link `link` (link index is wave-uniform; resolved by an unrolled select).
template <typename T, int TOPO> __device__ __forceinline__ void link_frame(const Kin<T, TOPO>& k, int link, const T* fpos, const T* frot,
                                                                             V3<T>& pos, M3<T>& rot) {
    constexpr int N = Topo<TOPO>::N;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i == link) {
            pos = k.o[i] + mul(k.R[i], load_v3(fpos));
            rot = mul(k.R[i], load_m3(frot));
        }
    }
}

// ------------------------------------------------------------------------------------------------ dynamics
// Outputs of one dynamics evaluation at (q, qd):
//   hbias  = ID(q, qd, 0)      (gravity + velocity-product generalised forces: the gravity-compensation torque)
//   qdamp  = generalised force of Bullet's per-body linear/angular velocity damping
//   Minv   = inverse joint-space inertia (full symmetric storage)
// BIAS = false skips hbias (left 0): inside a sim tick with the reference's gravity compensation the applied torque ID(q, qd, 0)
// and the forward dynamics' bias force are the same vector and cancel, (hbias - d) - hbias = -d, so the whole velocity-product /
// gravity recursion (wd, ao, link wrenches and their leaf-to-root accumulation) is dead weight there.
template <typename T, int TOPO, bool BIAS = true, bool TRIG = false>
__device__ __forceinline__ void dynamics_terms(const DevRobot<T>& m, const T (&q)[Topo<TOPO>::N], const T (&qd)[Topo<TOPO>::N],
                                               T (&hbias)[Topo<TOPO>::N], T (&qdamp)[Topo<TOPO>::N],
                                               T (&Minv)[Topo<TOPO>::N][Topo<TOPO>::N], T& traceM, const V3<T> g, Kin<T, TOPO>& k,
                                               const JointTrig<T, Topo<TOPO>::N>* trig = nullptr) {
    constexpr int N = Topo<TOPO>::N;
    forward_kinematics<T, TOPO, TRIG>(m, q, k, trig);
    V3<T> w[N], wd[N], vo[N], ao[N];
    V3<T> WF[N], WN[N], DF[N], DN[N], hc[N];     // bias wrench, damping wrench (about o_i), composite first moment
    S3<T> Io[N];
    T mc[N];
    // root -> leaf: velocities, velocity-product accelerations (qdd = 0), per-link wrenches about the joint origin
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int p = Topo<TOPO>::parent(i);
        const V3<T> aq = qd[i] * k.a[i];
        if (p < 0) {
            w[i] = aq;
            vo[i] = mk<T>(0, 0, 0);
            if (BIAS) {
                wd[i] = mk<T>(0, 0, 0);
                ao[i] = mk<T>(0, 0, 0) - g;         // fictitious base acceleration = -gravity
            }
        } else {
            const V3<T> r = k.o[i] - k.o[p];
            w[i] = w[p] + aq;
            vo[i] = vo[p] + cross(w[p], r);
            if (BIAS) {
                wd[i] = wd[p] + cross(w[p], aq);
                ao[i] = ao[p] + cross(wd[p], r) + cross(w[p], cross(w[p], r));
            }
        }
        // merged link body
        const V3<T> rc = mul(k.R[i], load_v3(m.lcom[i]));
        const S3<T> Il{m.linert[i][0], m.linert[i][1], m.linert[i][2], m.linert[i][3], m.linert[i][4], m.linert[i][5]};
        const S3<T> Iw = rotate(k.R[i], Il);
        if (BIAS) {
            const V3<T> ac = ao[i] + cross(wd[i], rc) + cross(w[i], cross(w[i], rc));
            const V3<T> F = m.lmass[i] * ac;
            const V3<T> Nc = mul(Iw, wd[i]) + cross(w[i], mul(Iw, w[i]));
            WF[i] = F;
            WN[i] = Nc + cross(rc, F);
        }
        mc[i] = m.lmass[i];
        hc[i] = m.lmass[i] * rc;
        Io[i] = Iw + point_inertia(m.lmass[i], rc);
        // per-body velocity damping  F = -m v (K + K|v|),  N = -(I w)(K + K|w|).  |w| is common to the bodies welded to a
        // link, so their angular parts are merged exactly through lang = sum_b R_b I_b R_b^T; |v| differs per body.
        const T sw = m.ang_damp + m.ang_damp * tsqrt_fast(dot(w[i], w[i]));
        const S3<T> Ia{m.lang[i][0], m.lang[i][1], m.lang[i][2], m.lang[i][3], m.lang[i][4], m.lang[i][5]};
        V3<T> dF = mk<T>(0, 0, 0);
        V3<T> dN = (-sw) * mul(k.R[i], mul(Ia, mulT(k.R[i], w[i])));
#pragma unroll
        for (int b = 0; b < kMaxBodiesPerLink; ++b) {
            if (m.bmass[i][b] > T(0)) {  // wave-uniform
                const V3<T> rb = mul(k.R[i], load_v3(m.bcom[i][b]));
                const V3<T> vb = vo[i] + cross(w[i], rb);
                const T sv = m.lin_damp + m.lin_damp * tsqrt_fast(dot(vb, vb));
                const V3<T> Fb = (-m.bmass[i][b] * sv) * vb;
                dF = dF + Fb;
                dN = dN + cross(rb, Fb);
            }
        }
        DF[i] = dF;
        DN[i] = dN;
    }
    // leaf -> root: accumulate subtree wrenches and composite inertias about each joint origin
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        hbias[i] = BIAS ? dot(k.a[i], WN[i]) : T(0);
        qdamp[i] = dot(k.a[i], DN[i]);
        const int p = Topo<TOPO>::parent(i);
        if (p >= 0) {
            const V3<T> r = k.o[i] - k.o[p];
            if (BIAS) {
                WF[p] = WF[p] + WF[i];
                WN[p] = WN[p] + WN[i] + cross(r, WF[i]);
            }
            DF[p] = DF[p] + DF[i];
            DN[p] = DN[p] + DN[i] + cross(r, DF[i]);
            Io[p] = Io[p] + Io[i] + point_inertia(mc[i], r) + cross_inertia(hc[i], r);
            hc[p] = hc[p] + hc[i] + mc[i] * r;
            mc[p] = mc[p] + mc[i];
        }
    }
    // joint-space inertia, lower triangle
    T L[N][N];
    traceM = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const V3<T> F = cross(k.a[j], hc[j]);
        const V3<T> Nj = mul(Io[j], k.a[j]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i == j) { L[j][j] = dot(k.a[j], Nj); traceM += L[j][j]; }
            else if (i < j) L[j][i] = is_ancestor_or_self<TOPO>(i, j) ? dot(k.a[i], Nj + cross(k.o[j] - k.o[i], F)) : T(0);
        }
    }
    // in-place Cholesky  M = L L^T  (Li = L^-1 shares the pivots' reciprocals)
    T Li[N][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T d = L[j][j];
#pragma unroll
        for (int c = 0; c < j; ++c) d -= L[j][c] * L[j][c];
        const T inv = trsqrt(d);
        L[j][j] = d * inv;
        Li[j][j] = inv;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            T s = L[i][j];
#pragma unroll
            for (int c = 0; c < j; ++c) s -= L[i][c] * L[j][c];
            L[i][j] = s * inv;
        }
    }
    // Linv (lower), then Minv = Linv^T Linv
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            T s = T(0);
#pragma unroll
            for (int c = j; c < i; ++c) s -= L[i][c] * Li[c][j];
            Li[i][j] = s * Li[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            T s = T(0);
#pragma unroll
            for (int c = i; c < N; ++c) s += Li[c][i] * Li[c][j];
            Minv[i][j] = s;
            Minv[j][i] = s;
        }
}

// Projected Gauss-Seidel over the joint-motor rows (btMultiBodyJointMotor rows solved by
// btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric [PARITY_ASSUMPTIONS A7]):
//   delta = rhs_i - (J_i . dv) jacDiagABInv_i ;  sum = lambda_i + delta ; clamp sum to +-maxImpulse ;  dv += Minv[:, i] delta
// `iters` sweeps, reverse row order on even sweeps, forward on odd.
//
// Bullet leaves the loop after a sweep whose largest squared delta is <= leastSquaresResidualThreshold (0 here, A7b).  In the
// oracle's dv-form that happens at the floating-point fixed point (sweep 50-60 in edge_follow) or never, when the iteration
// settles into a 1-ulp limit cycle; either way the state stops moving beyond its last bit.  pgs_unclamped below detects the same
// point in its residual form and leaves the loop (see there); pgs_clamped runs all sweeps.
//
// pgs_unclamped is used when no row can reach its impulse limit.  Each row update minimises the energy
// E = 1/2 (l - l*)^T A (l - l*), A = J Minv J^T = Minv, exactly along one coordinate, so E never grows from l = 0:
//   |l_i| <= ||l||_2 <= 2 ||l*||_A / sqrt(sigma_min(A)) <= 2 sigma_max(M) ||dv*||_2 <= 2 trace(M) ||dv*||_2
// (dv* = requested velocity change).  The caller checks 2 trace(M) ||dv*||_2 < maxImpulse / 2 per lane; otherwise the
// wavefront takes pgs_clamped, which evaluates the identical expressions plus the clamp as branch-free selects.
// Unclamped sweeps are carried in residual form: r_j = rhs_j - dv_j jdi_j is updated directly,
//   row i:  t = r_i ;  r_j -= (Minv[j][i] jdi_j) t  for all j      (6 FMAs per row, one-FMA serial chain)
// which is the same Gauss-Seidel recurrence as above with dv eliminated; dv is recovered once at the end.
template <typename T, int N, bool FWD>
__device__ __forceinline__ void pgs_sweep_unclamped(const T (&G)[N][N], T (&r)[N]) {
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
        const int i = FWD ? jj : N - 1 - jj;
        const T t = r[i];
#pragma unroll
        for (int j = 0; j < N; ++j) r[j] -= G[j][i] * t;
    }
}
// Convergence exit: in this form r_j -> 0 geometrically and dv_j = (rimp_j - r_j) Minv_jj, so once every |r_j| is below
// 2^-56 max|rimp| (2^-27 in f32) further sweeps cannot move any dv_j by more than a fraction of its last bit.  That is Bullet's
// own exit (leastSquaresResidual <= threshold, threshold 0 [PARITY_ASSUMPTIONS A7b]) taken where the oracle's dv-form reaches its
// floating-point fixed point or 1-ulp limit cycle (sweep 50-60 of 150 in edge_follow).  Checked after every block of 4 sweeps, wave-uniform
// (__all).  iters < 0 runs exactly |iters| sweeps (tg_config.pgs_full_sweeps).
// Returns the number of sweeps after which the exit fired, or -1 if it did not.
template <typename T, int N>
__device__ __forceinline__ int pgs_unclamped(const T (&Minv)[N][N], const T (&rimp)[N], const T (&jdi)[N], int iters, T (&dv)[N]) {
    T G[N][N], r[N];
    T thr = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        r[j] = rimp[j];
        thr = tmax(thr, tabs(rimp[j]));
#pragma unroll
        for (int i = 0; i < N; ++i) G[j][i] = Minv[j][i] * jdi[j];
    }
    const bool full = iters < 0;
    const int n_it = full ? -iters : iters;
    thr = full ? T(-1) : thr * (sizeof(T) == 8 ? T(1.3877787807814457e-17) : T(7.450580596923828e-09));
    int it = 0;
    bool converged = false;
    for (; it + 3 < n_it; it += 4) {          // blocks of four sweeps (reverse, forward, reverse, forward), test after each block
        pgs_sweep_unclamped<T, N, false>(G, r);
        pgs_sweep_unclamped<T, N, true>(G, r);
        pgs_sweep_unclamped<T, N, false>(G, r);
        pgs_sweep_unclamped<T, N, true>(G, r);
        T mx = T(0);
#pragma unroll
        for (int j = 0; j < N; ++j) mx = tmax(mx, tabs(r[j]));
        if (__all(mx <= thr)) { converged = true; break; }
    }
    if (!converged) {
        for (; it + 1 < n_it; it += 2) {
            pgs_sweep_unclamped<T, N, false>(G, r);
            pgs_sweep_unclamped<T, N, true>(G, r);
        }
        if (it < n_it) pgs_sweep_unclamped<T, N, false>(G, r);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) dv[j] = (rimp[j] - r[j]) * Minv[j][j];
    return converged ? it + 4 : -1;
}
// The same iteration with eight sweeps at a time, for arms whose motor solve does not converge inside the sweep budget (the MG400: the
// contraction per sweep is ~0.94, so all 150 sweeps run in every tick of every step and reset, and they are 9 600 of the tick's FMAs).
// An unclamped sweep is a LINEAR map of the residual vector, so a reverse + forward pair is one N x N matrix P - built column by column by
// sweeping the unit vectors - four pairs are P^4 (two squarings), and r after n sweeps is P^(n/2) r0 however the factors are grouped: the
// pairs that do not fill a block of four run first as plain sweeps, then P^4 is applied n/8 times (64 FMAs per 8 sweeps instead of 512).
// Same recurrence, same row order, same exit rule (tested after every block of eight sweeps here); the results differ from the
// sweep-by-sweep evaluation by rounding only (the tests that compare with the oracle's literal 150 sweeps hold their 1e-9 rad).
// n_it even.  `pgs_full_sweeps` keeps the literal loop above.
template <typename T, int N>
__device__ __forceinline__ int pgs_unclamped_blocks(const T (&Minv)[N][N], const T (&rimp)[N], const T (&jdi)[N], int n_it, T (&dv)[N]) {
    T r[N], B[N][N];
    T thr = T(0);
    const int pairs = n_it / 2, lead = pairs & 3, blocks = pairs >> 2;
    {
        T G[N][N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            r[j] = rimp[j];
            thr = tmax(thr, tabs(rimp[j]));
#pragma unroll
            for (int i = 0; i < N; ++i) G[j][i] = Minv[j][i] * jdi[j];
        }
        for (int k = 0; k < lead; ++k) {
            pgs_sweep_unclamped<T, N, false>(G, r);
            pgs_sweep_unclamped<T, N, true>(G, r);
        }
        T P[N][N];
#pragma unroll
        for (int c = 0; c < N; ++c) {            // column c of P: the pair applied to e_c
            T t[N];
#pragma unroll
            for (int j = 0; j < N; ++j) t[j] = j == c ? T(1) : T(0);
            pgs_sweep_unclamped<T, N, false>(G, t);
            pgs_sweep_unclamped<T, N, true>(G, t);
#pragma unroll
            for (int j = 0; j < N; ++j) P[j][c] = t[j];
        }
        T P2[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < N; ++k) acc += P[i][k] * P[k][j];
                P2[i][j] = acc;
            }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < N; ++k) acc += P2[i][k] * P2[k][j];
                B[i][j] = acc;
            }
    }
    thr = thr * (sizeof(T) == 8 ? T(1.3877787807814457e-17) : T(7.450580596923828e-09));
    int it = 2 * lead;
    bool converged = false;
    for (int b = 0; b < blocks; ++b) {
        T rn[N];
        T mx = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += B[i][j] * r[j];
            rn[i] = acc;
            mx = tmax(mx, tabs(acc));
        }
#pragma unroll
        for (int i = 0; i < N; ++i) r[i] = rn[i];
        it += 8;
        if (__all(mx <= thr)) { converged = true; break; }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) dv[j] = (rimp[j] - r[j]) * Minv[j][j];
    return converged ? it : -1;
}
template <typename T, int N, bool FWD>
__device__ __forceinline__ void pgs_sweep_clamped(const T (&Minv)[N][N], const T (&rimp)[N], const T (&jdi)[N], T maximp, T (&lam)[N],
                                                  T (&dv)[N]) {
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
        const int i = FWD ? jj : N - 1 - jj;
        const T t = rimp[i] - dv[i] * jdi[i];
        const T sum = lam[i] + t;
        const T lo = sum < -maximp ? -maximp : sum;
        const T sc = lo > maximp ? maximp : lo;
        const T delta = (sc == sum) ? t : sc - lam[i];   // branch-free select: the unclamped case keeps t exactly
        lam[i] = sc;
#pragma unroll
        for (int r = 0; r < N; ++r) dv[r] += Minv[r][i] * delta;
    }
}
template <typename T, int N>
__device__ __forceinline__ void pgs_clamped(const T (&Minv)[N][N], const T (&rimp)[N], const T (&jdi)[N], T maximp, int iters, T (&dv)[N]) {
    T lam[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { lam[i] = T(0); dv[i] = T(0); }
    iters = iters < 0 ? -iters : iters;
    int it = 0;
    for (; it + 1 < iters; it += 2) {
        pgs_sweep_clamped<T, N, false>(Minv, rimp, jdi, maximp, lam, dv);
        pgs_sweep_clamped<T, N, true>(Minv, rimp, jdi, maximp, lam, dv);
    }
    if (it < iters) pgs_sweep_clamped<T, N, false>(Minv, rimp, jdi, maximp, lam, dv);
}

// Threshold mode (DevRobot::res_thr > 0): the literal clamped sweep with Bullet's residual and a per-lane `live` flag - an env whose sweep met the
// threshold keeps its impulses while its 63 neighbours go on (Bullet leaves the loop per world; a lane is a world).  Returns the sweep's
// residual: the largest squared velocity change delta / jacDiagABInv = delta * Minv_ii of a row update (oracle/minibullet.c: mb_step).
template <typename T, int N, bool FWD>
__device__ __forceinline__ T pgs_sweep_residual(const T (&Minv)[N][N], const T (&rimp)[N], const T (&jdi)[N], T maximp, bool live, T (&lam)[N],
                                                T (&dv)[N]) {
    T res = T(0);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
        const int i = FWD ? jj : N - 1 - jj;
        const T t = rimp[i] - dv[i] * jdi[i];
        const T sum = lam[i] + t;
        const T lo = sum < -maximp ? -maximp : sum;
        const T sc = lo > maximp ? maximp : lo;
        const T delta = live ? ((sc == sum) ? t : sc - lam[i]) : T(0);
        lam[i] = live ? sc : lam[i];
#pragma unroll
        for (int r = 0; r < N; ++r) dv[r] += Minv[r][i] * delta;
        const T dvel = delta * Minv[i][i];
        res = tmax(res, dvel * dvel);
    }
    return res;
}
// ... and the loop: at most `iters` sweeps, reverse order first; returns the number of sweeps THIS lane's env ran.
template <typename T, int N>
__device__ __forceinline__ int pgs_threshold(const T (&Minv)[N][N], const T (&rimp)[N], const T (&jdi)[N], T maximp, int iters, T res_thr, T (&dv)[N]) {
    T lam[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { lam[i] = T(0); dv[i] = T(0); }
    iters = iters < 0 ? -iters : iters;
    bool live = iters > 0;
    int sweeps = 0;
    for (int it = 0; it < iters && __any(live); it += 2) {
        const T r0 = pgs_sweep_residual<T, N, false>(Minv, rimp, jdi, maximp, live, lam, dv);
        if (live) { ++sweeps; live = !(r0 <= res_thr); }
        if (it + 1 < iters) {
            const T r1 = pgs_sweep_residual<T, N, true>(Minv, rimp, jdi, maximp, live, lam, dv);
            if (live) { ++sweeps; live = !(r1 <= res_thr); }
        }
    }
    return sweeps;
}

enum { kMotorOff = 0, kMotorVelocity = 1, kMotorPosition = 2 };

// One stepSimulation() tick with the reference's per-tick gravity compensation (robot.py:131-141).
//   applied torque  = ID(q, qd, 0)                     (base_robot_arm.py:174-189, TORQUE_CONTROL feed-forward)
//                   - joint_damping * qd                (changeDynamics(jointDamping), base_robot_arm.py:25)
//   unconstrained   qdd = Minv (applied - hbias + qdamp);  v = qd + dt qdd
//   motors          projected Gauss-Seidel on the velocity-level rows  v_i -> target_i,  |impulse| <= max_force dt,
//                   `iters` sweeps alternating reverse/forward row order, early exit on an exactly-zero sweep
//   integration     q += dt qd
template <typename T, int TOPO, int MOTOR, bool GC = true, bool TRIG = false>
__device__ __forceinline__ void sim_tick(const DevRobot<T>& m, T (&q)[Topo<TOPO>::N], T (&qd)[Topo<TOPO>::N],
                                         const T (&q_des)[Topo<TOPO>::N], const T (&qd_des)[Topo<TOPO>::N], T kp, T kd, T max_force, T dt,
                                         int iters, JointTrig<T, Topo<TOPO>::N>* trig = nullptr, int* verified = nullptr,
                                         int* sweep_acc = nullptr /* threshold mode: += the sweeps this tick ran */) {
    constexpr int N = Topo<TOPO>::N;
    // Analytic fixed point.  Every joint carries a motor row (J = e_i), so the rows together prescribe the whole velocity: the unique
    // solution of the unclamped system A lambda = rhs, A = Minv, is dv = des - v, i.e. the post-solve velocity is `des` itself, and with
    // the reference's velocity gain 1 (ur5.py:19-21, mg400.py:27-29)  des = kp (q_des - q)/dt + qd_des  does not depend on the dynamics
    // at all.  Gauss-Seidel on an SPD system converges to that solution, but whether `iters` sweeps get there depends on the arm: the
    // UR5's iteration contracts by ~0.5 per sweep (the oracle's 150 sweeps land on the target to the last bit: |qd - target| = 0 after
    // every tick), the MG400's by ~0.94 (0.3 % of the initial error survives 150 sweeps, so there the dynamics do shape the result).
    // The shortcut is therefore licensed at run time: `*verified` counts the ticks for which it may be taken, and is (re)armed only when
    // the full solve below has just demonstrated last-bit convergence (1e-17 of the jump) within 80 % of the sweep budget in this
    // configuration, so that the remainder after all sweeps is < 1e-21 of the jump.  It is taken when, in addition, no row can reach its impulse limit - proved a
    // priori by the same energy bound as for pgs_unclamped, with the host-side bound trace_bound >= trace(M(q)) and the damping impulse
    // bounded through the same quantity (gravity is compensated).  Then the tick is  qd = des, q += dt des.  iters < 0
    // (pgs_full_sweeps) forces the literal path.
    if (MOTOR != kMotorOff && GC && iters >= 0 && kd == T(1) && verified != nullptr && *verified > 0 && !(m.res_thr > T(0))) {
        T des[N], dvw = T(0), v2 = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            des[i] = ((MOTOR == kMotorPosition) ? kp * (q_des[i] - q[i]) / dt : T(0)) + qd_des[i];
            dvw += m.diag_sqrt[i] * tabs(des[i] - qd[i]);
            v2 += qd[i] * qd[i];
        }
        // lambda* = M (des - v) = M (des - qd) - dt f  with f the damping force, ||f|| <= (joint_damp + 2 (K_lin + K_ang)(1 + |v|) trace(M)) ||qd||;
        // the constants below are generous (sqrt(N) <= 3, |v| < 1).  Energy argument: Gauss-Seidel on the SPD system A = Minv never
        // increases the A-norm of the error, so ||lambda^k||_A <= 2 ||lambda*||_A; a row obeys |lambda_i| <= sqrt(M_ii) ||lambda||_A
        // (Cauchy-Schwarz in the A inner product) and ||M dv||_A = ||dv||_M <= sum_j sqrt(M_jj) |dv_j| (triangle inequality), so with the
        // host's per-joint bounds d_j >= M_jj(q):  |lambda_i^k| <= 2 sqrt(d_max) sum_j sqrt(d_j) |dv_j|  (<= 2 trace ||dv||_2, the
        // earlier form, but far tighter when the jump sits in the light wrist joints).
        const T lam_star = m.diag_sqrt_max * dvw +
                           dt * (m.joint_damp + T(4) * (m.lin_damp + m.ang_damp) * m.trace_bound) * T(3) * tsqrt_fast(v2);
        if (__all(T(2.5) * lam_star < max_force * dt)) {   // energy bound: no iterate exceeds 2 ||lambda*||; 25 % margin on top
            T dq[N];
#pragma unroll
            for (int i = 0; i < N; ++i) { qd[i] = des[i]; dq[i] = dt * des[i]; q[i] += dq[i]; }
            if (TRIG) trig_advance<T, N>(q, dq, *trig);
            --*verified;
            return;
        }
    }
    // Compiler barrier: without it the ~250 scalar robot constants are hoisted out of the caller's tick loop, overflow the
    // 100 SGPRs and get spilled into VGPR lanes (v_readlane per use).  Re-issuing the s_loads every tick is cheaper.
    asm volatile("" ::: "memory");
    T hb[N], qdm[N], Minv[N][N], traceM;
    Kin<T, TOPO> kin;
    dynamics_terms<T, TOPO, !GC, TRIG>(m, q, qd, hb, qdm, Minv, traceM, load_v3(m.gravity), kin, trig);
    T rhs[N], v[N];
#pragma unroll
    for (int i = 0; i < N; ++i)   // GC: (hb - d qd) - hb + qdm with the two hb cancelled analytically
        rhs[i] = GC ? (qdm[i] - m.joint_damp * qd[i]) : ((T(0) - m.joint_damp * qd[i]) - hb[i]) + qdm[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < N; ++j) acc += Minv[i][j] * rhs[j];
        v[i] = qd[i] + dt * acc;
    }
    if (MOTOR != kMotorOff) {
        T rimp[N], jdi[N], dv[N];
        const T maximp = max_force * dt;
        T dv2 = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const T pos_term = (MOTOR == kMotorPosition) ? kp * (q_des[i] - q[i]) / dt : T(0);
            const T des = pos_term + v[i] + kd * (qd_des[i] - v[i]);
            jdi[i] = trcp(Minv[i][i]);
            rimp[i] = (des - v[i]) * jdi[i];
            dv2 += (des - v[i]) * (des - v[i]);
        }
        const bool no_clamp_possible = T(4) * traceM * tsqrt(dv2) < maximp;   // 2 trace(M) ||dv*|| < maxImpulse / 2
        int sweeps = -1;
        if (m.res_thr > T(0)) {              // threshold mode: Bullet's exit per env, sweep by sweep; never licensed
            const int ran = pgs_threshold<T, N>(Minv, rimp, jdi, maximp, iters, m.res_thr, dv);
            if (sweep_acc != nullptr) *sweep_acc += ran;
        }
        else if (__all(no_clamp_possible)) {
            // the MG400's motor solve never converges inside the budget: eight sweeps at a time (pgs_unclamped_blocks); the UR5 leaves after
            // 50 - 60 sweeps and keeps the sweep-by-sweep loop, as does every `pgs_full_sweeps` run (iters < 0)
            if (N == 8 && iters >= 32 && (iters & 1) == 0) sweeps = pgs_unclamped_blocks<T, N>(Minv, rimp, jdi, iters, dv);
            else sweeps = pgs_unclamped<T, N>(Minv, rimp, jdi, iters, dv);
        }
        else pgs_clamped<T, N>(Minv, rimp, jdi, maximp, iters, dv);
        if (verified != nullptr) *verified = (iters > 0 && sweeps > 0 && 5 * sweeps <= 4 * iters) ? 24 : -1;   // -1: a full solve ran and did not qualify
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += dv[i];
    }
    T dq[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        qd[i] = v[i];
        dq[i] = dt * v[i];
        q[i] += dq[i];
    }
    if (TRIG) trig_advance<T, N>(q, dq, *trig);
}

// ------------------------------------------------------------------------------------------------ arm + free body + P2P
// object_balance: a free rigid body (the pole, all links welded) tied to an arm link by a point-to-point constraint
// (object_balance_env.py:261-283).  One tick = arm forward dynamics as above, free-body forward dynamics (gravity,
// one-shot external force, gyroscopic torque, no velocity damping), then ONE projected Gauss-Seidel loop over the joint
// motor rows followed by the three P2P rows (reverse order on even sweeps) [PARITY_ASSUMPTIONS A18-A21].
// The loop is carried in the Delassus/residual form of the same recurrence:  A = J Minv_sys J^T (9x9),
//   r_j = rhs_j / A_jj - sum_q (A_jq / A_jj) lambda_q ;  row i: delta = clamp(lambda_i + r_i) - lambda_i ; r_j -= G_ji delta.
template <typename T> struct FreeBody {
    V3<T> pos;   // base (inertial) frame origin, world — what getBasePositionAndOrientation reports
    M3<T> R;     // base frame orientation
    V3<T> v, w;  // velocity of the composite centre of mass, angular velocity (world)
};
template <typename T> struct BodyConst {
    T mass;
    V3<T> com;        // composite centre of mass in the base frame
    S3<T> inertia;    // about com, base-frame axes
    int link;         // arm link carrying pivot A
    V3<T> pivot_a;    // in that link's frame
    T erp, max_impulse;
};

template <typename T> __device__ __forceinline__ S3<T> inverse(const S3<T>& A) {
    const T c00 = A.yy * A.zz - A.yz * A.yz, c01 = A.xz * A.yz - A.xy * A.zz, c02 = A.xy * A.yz - A.xz * A.yy;
    const T id = trcp(A.xx * c00 + A.xy * c01 + A.xz * c02);   // seed + Newton (1-2 ulp): on the per-tick dependent chain of the free-body envs
    return {c00 * id, c01 * id, c02 * id, (A.xx * A.zz - A.xz * A.xz) * id, (A.xy * A.xz - A.xx * A.yz) * id, (A.xx * A.yy - A.xy * A.xy) * id};
}

// Orientation update by world angular velocity w over dt: exponential map with Bullet's small-angle branch, then
// Gram-Schmidt (Bullet renormalises its quaternion).
// sin / cos of a small argument (the half rotation angle of one tick: |x| <= 0.05 for |w| <= 24 rad/s): Taylor polynomials whose
// truncation error (x^11 / 11!, x^12 / 12!) is below 1e-22, instead of the library's range-reduced sincos (~150 instructions each);
// the exact routine is taken when any lane of the wavefront is outside that range.
template <typename T> __device__ __forceinline__ void sincos_small(T x, T* s, T* c) {
    if (__any(tabs(x) > T(0.05))) { tsincos(x, s, c); return; }
    const T x2 = x * x;
    *s = x * (T(1) + x2 * (T(-1.0 / 6.0) + x2 * (T(1.0 / 120.0) + x2 * (T(-1.0 / 5040.0) + x2 * T(1.0 / 362880.0)))));
    *c = T(1) + x2 * (T(-0.5) + x2 * (T(1.0 / 24.0) + x2 * (T(-1.0 / 720.0) + x2 * (T(1.0 / 40320.0) + x2 * T(-1.0 / 3628800.0)))));
}
template <typename T> __device__ __forceinline__ void integrate_rotation(M3<T>& R, V3<T> w, T dt) {
    const T w2 = dot(w, w);
    const T rang = trsqrt(w2 > T(0) ? w2 : T(1));   // 1 / |w| by seed + Newton instead of sqrt and a division (the chain of every tick)
    const T ang = w2 > T(0) ? w2 * rang : T(0);
    T sh, qw;
    sincos_small(ang * dt * T(0.5), &sh, &qw);
    T k;
    if (ang < T(0.001)) k = T(0.5) * dt - (dt * dt * dt) * T(0.020833333333) * ang * ang;
    else k = sh * rang;
    const V3<T> ax = k * w;
    const T nrm = trsqrt(dot(ax, ax) + qw * qw);
    const T x = ax.x * nrm, y = ax.y * nrm, z = ax.z * nrm, ww = qw * nrm;
    M3<T> dR;
    dR.m[0] = T(1) - T(2) * (y * y + z * z); dR.m[1] = T(2) * (x * y - z * ww); dR.m[2] = T(2) * (x * z + y * ww);
    dR.m[3] = T(2) * (x * y + z * ww); dR.m[4] = T(1) - T(2) * (x * x + z * z); dR.m[5] = T(2) * (y * z - x * ww);
    dR.m[6] = T(2) * (x * z - y * ww); dR.m[7] = T(2) * (y * z + x * ww); dR.m[8] = T(1) - T(2) * (x * x + y * y);
    const M3<T> Rn = mul(dR, R);
    V3<T> c0{Rn.m[0], Rn.m[3], Rn.m[6]}, c1{Rn.m[1], Rn.m[4], Rn.m[7]};
    c0 = trsqrt(dot(c0, c0)) * c0;
    c1 = c1 - dot(c0, c1) * c0;
    c1 = trsqrt(dot(c1, c1)) * c1;
    const V3<T> c2 = cross(c0, c1);
    R.m[0] = c0.x; R.m[3] = c0.y; R.m[6] = c0.z;
    R.m[1] = c1.x; R.m[4] = c1.y; R.m[7] = c1.z;
    R.m[2] = c2.x; R.m[5] = c2.y; R.m[8] = c2.z;
}

// Body half of the analytic arm + body + P2P tick: with the arm moving at the motors' targets, pivot A sits at `pa` and moves with `va`; the
// three P2P impulses solve the body-only 3 x 3 system (see sim_tick_body).  lam_arm = the motor rows' share of the a-priori bound.  Returns
// true and advances the body when the test holds for the whole wavefront, false with nothing changed otherwise.
template <typename T>
__device__ __forceinline__ bool body_tick_pivot(T lam_arm, T max_force, T dt, V3<T> gravity, FreeBody<T>& b, const BodyConst<T>& bc, V3<T> pivot_b,
                                                V3<T> ext_force, V3<T> ext_pos, bool ext_pending, V3<T> pa, V3<T> va) {
    const S3<T> Iw = rotate(b.R, bc.inertia), Iwi = inverse(Iw);
    V3<T> xc = b.pos + mul(b.R, bc.com);
    V3<T> F = bc.mass * gravity, Nt = mk<T>(0, 0, 0);
    if (ext_pending) { F = F + ext_force; Nt = Nt + cross(ext_pos - xc, ext_force); }
    Nt = Nt - cross(b.w, mul(Iw, b.w));
    const V3<T> vb = b.v + (dt / bc.mass) * F, wb = b.w + dt * mul(Iwi, Nt);
    const V3<T> pb = b.pos + mul(b.R, pivot_b), rb = pb - xc;
    const V3<T> gap = pa - pb, cv = va - (vb + cross(wb, rb));
    const V3<T> rhs = (-bc.erp / dt) * gap - cv;
    const V3<T> e[3] = {mk<T>(1, 0, 0), mk<T>(0, 1, 0), mk<T>(0, 0, 1)};
    V3<T> rxe[3], Wang[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) { rxe[x] = cross(rb, e[x]); Wang[x] = mul(Iwi, rxe[x]); }
    const T im = T(1) / bc.mass;
    const S3<T> App{im + dot(rxe[0], Wang[0]), dot(rxe[0], Wang[1]), dot(rxe[0], Wang[2]), im + dot(rxe[1], Wang[1]), dot(rxe[1], Wang[2]),
                    im + dot(rxe[2], Wang[2])};
    const V3<T> lp = mul(inverse(App), rhs);
    const T lpn = tsqrt_fast(dot(lp, lp));
    const T lam_star = lam_arm + T(6) * lpn;     // per-joint form of the energy bound, see sim_tick
    if (!__all(T(4) * lam_star < max_force * dt && T(8) * lpn < bc.max_impulse)) return false;
    b.v = vb - im * lp;
    b.w = wb - mul(Iwi, cross(rb, lp));
    xc = xc + dt * b.v;
    integrate_rotation(b.R, b.w, dt);
    b.pos = xc - mul(b.R, bc.com);
    return true;
}
// world position of pivot A on link `link` and its velocity under the joint velocities `des`, from a finished forward-kinematics pass
template <typename T, int TOPO>
__device__ __forceinline__ void pivot_state(const Kin<T, TOPO>& kin, const BodyConst<T>& bc, const T (&des)[Topo<TOPO>::N], V3<T>& pa, V3<T>& va) {
    constexpr int N = Topo<TOPO>::N;
    M3<T> Rl;
    const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
    const T pav[3] = {bc.pivot_a.x, bc.pivot_a.y, bc.pivot_a.z};
    link_frame<T, TOPO>(kin, bc.link, pav, ident, pa, Rl);
    va = mk<T>(0, 0, 0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        bool on_path = false;
#pragma unroll
        for (int l = 0; l < N; ++l)
            if (l == bc.link && is_ancestor_or_self<TOPO>(i, l)) on_path = true;
        if (on_path) va = va + des[i] * cross(kin.a[i], pa - kin.o[i]);
    }
}

// The analytic fixed point of one arm + body + P2P tick (see sim_tick_body): advances q, qd, the carried sines / cosines and the body and
// returns true when its a-priori test holds for the whole wavefront; returns false with nothing changed otherwise (the caller then solves).
template <typename T, int TOPO, int MOTOR>
__device__ __forceinline__ bool body_tick_analytic(const DevRobot<T>& m, T (&q)[Topo<TOPO>::N], T (&qd)[Topo<TOPO>::N],
                                                   const T (&q_des)[Topo<TOPO>::N], const T (&qd_des)[Topo<TOPO>::N], T kp, T max_force, T dt, V3<T> gravity,
                                                   FreeBody<T>& b, const BodyConst<T>& bc, V3<T> pivot_b, V3<T> ext_force, V3<T> ext_pos, bool ext_pending,
                                                   JointTrig<T, Topo<TOPO>::N>* trig) {
    constexpr int N = Topo<TOPO>::N;
    T des[N], dvw = T(0), v2 = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        des[i] = ((MOTOR == kMotorPosition) ? kp * (q_des[i] - q[i]) / dt : T(0)) + qd_des[i];
        dvw += m.diag_sqrt[i] * tabs(des[i] - qd[i]);
        v2 += qd[i] * qd[i];
    }
    Kin<T, TOPO> kin;
    if (trig != nullptr) forward_kinematics<T, TOPO, true>(m, q, kin, trig);
    else forward_kinematics<T, TOPO>(m, q, kin);
    V3<T> pa, va;
    pivot_state<T, TOPO>(kin, bc, des, pa, va);
    const T lam_arm = m.diag_sqrt_max * dvw + dt * (m.joint_damp + T(4) * (m.lin_damp + m.ang_damp) * m.trace_bound) * T(3) * tsqrt_fast(v2);
    if (!body_tick_pivot<T>(lam_arm, max_force, dt, gravity, b, bc, pivot_b, ext_force, ext_pos, ext_pending, pa, va)) return false;
    T dq[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { qd[i] = des[i]; dq[i] = dt * des[i]; q[i] += dq[i]; }
    if (trig != nullptr) trig_advance<T, N>(q, dq, *trig);
    return true;
}

template <typename T, int TOPO, int MOTOR>
__device__ __forceinline__ void sim_tick_body(const DevRobot<T>& m, T (&q)[Topo<TOPO>::N], T (&qd)[Topo<TOPO>::N],
                                              const T (&q_des)[Topo<TOPO>::N], const T (&qd_des)[Topo<TOPO>::N], T kp, T kd, T max_force, T dt,
                                              int iters, V3<T> gravity, FreeBody<T>& b, const BodyConst<T>& bc, V3<T> pivot_b,
                                              V3<T> ext_force, V3<T> ext_pos, bool ext_pending, int* verified = nullptr,
                                              JointTrig<T, Topo<TOPO>::N>* trig = nullptr /* carried sines / cosines (k_step_body) */,
                                              int* sweep_acc = nullptr /* threshold mode: += the sweeps this tick ran */) {
    constexpr int N = Topo<TOPO>::N;
    constexpr int NR = N + 3;
    // Analytic fixed point (see sim_tick).  The motor rows still prescribe the whole arm velocity, whatever the P2P rows pull: at the
    // solution of the unclamped system the arm moves with `des` and the three P2P impulses solve the body-only system
    //   (1/m I + [rb x]^T Iw^-1 [rb x]) lambda = -erp gap/dt - (J_a des - v_pivot_b)        (3x3 SPD, solved directly),
    // the motors absorbing the reaction.  Same licence as in sim_tick: a full solve of this env step must have converged to the last
    // bit within 80 % of the sweep budget (the coupled iteration contracts like the arm's alone, ~0.5 per sweep), and no row may be
    // able to reach its limit (motor bound extended by the P2P reaction, P2P impulse far from its 500 N s cap).
    if (MOTOR != kMotorOff && iters >= 0 && kd == T(1) && verified != nullptr && *verified > 0 && !(m.res_thr > T(0))) {
        if (body_tick_analytic<T, TOPO, MOTOR>(m, q, qd, q_des, qd_des, kp, max_force, dt, gravity, b, bc, pivot_b, ext_force, ext_pos, ext_pending, trig)) {
            --*verified;
            return;
        }
    }
    asm volatile("" ::: "memory");
    T hb[N], qdm[N], Minv[N][N], traceM;
    Kin<T, TOPO> kin;
    dynamics_terms<T, TOPO, false>(m, q, qd, hb, qdm, Minv, traceM, gravity, kin);
    T v[N];
    {
        T rhs[N];
#pragma unroll
        for (int i = 0; i < N; ++i) rhs[i] = qdm[i] - m.joint_damp * qd[i];   // gravity compensation cancels hbias analytically
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += Minv[i][j] * rhs[j];
            v[i] = qd[i] + dt * acc;
        }
    }
    // free body
    const S3<T> Iw = rotate(b.R, bc.inertia), Iwi = inverse(Iw);
    const V3<T> cw = mul(b.R, bc.com);
    V3<T> xc = b.pos + cw;
    V3<T> F = bc.mass * gravity, Nt = mk<T>(0, 0, 0);
    if (ext_pending) { F = F + ext_force; Nt = Nt + cross(ext_pos - xc, ext_force); }
    Nt = Nt - cross(b.w, mul(Iw, b.w));
    V3<T> vb = b.v + (dt / bc.mass) * F, wb = b.w + dt * mul(Iwi, Nt);
    // P2P geometry: pivots, arm translational Jacobian at pivot A
    V3<T> pa; M3<T> Rl;
    {
        const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
        const T pav[3] = {bc.pivot_a.x, bc.pivot_a.y, bc.pivot_a.z};
        link_frame<T, TOPO>(kin, bc.link, pav, ident, pa, Rl);
    }
    const V3<T> pb = b.pos + mul(b.R, pivot_b);
    const V3<T> rb = pb - xc;
    T Jt[3][N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        bool on_path = false;
#pragma unroll
        for (int l = 0; l < N; ++l)
            if (l == bc.link && is_ancestor_or_self<TOPO>(i, l)) on_path = true;
        const V3<T> jt = cross(kin.a[i], pa - kin.o[i]);
        Jt[0][i] = on_path ? jt.x : T(0); Jt[1][i] = on_path ? jt.y : T(0); Jt[2][i] = on_path ? jt.z : T(0);
    }
    // W_arm[:, x] = Minv Jt[x, :]^T ; body parts: lin -e_x / m, ang -Iw^-1 (rb x e_x)
    T Wa[N][3];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += Minv[i][j] * Jt[x][j];
            Wa[i][x] = acc;
        }
    const V3<T> e[3] = {mk<T>(1, 0, 0), mk<T>(0, 1, 0), mk<T>(0, 0, 1)};
    V3<T> rxe[3], Wang[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) { rxe[x] = cross(rb, e[x]); Wang[x] = mul(Iwi, rxe[x]); }
    // Delassus matrix A (NR x NR), symmetric
    T A[NR][NR];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) A[i][j] = Minv[i][j];
#pragma unroll
        for (int x = 0; x < 3; ++x) { A[i][N + x] = Wa[i][x]; A[N + x][i] = Wa[i][x]; }
    }
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            T acc = (x == y) ? T(1) / bc.mass : T(0);
#pragma unroll
            for (int i = 0; i < N; ++i) acc += Jt[x][i] * Wa[i][y];
            A[N + x][N + y] = acc + dot(rxe[x], Wang[y]);
        }
    // velocity-level right-hand sides and limits
    T r[NR], lim[NR], lam[NR];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const T pos_term = (MOTOR == kMotorPosition) ? kp * (q_des[i] - q[i]) / dt : T(0);
        const T des = pos_term + v[i] + kd * (qd_des[i] - v[i]);
        r[i] = (MOTOR != kMotorOff) ? des - v[i] : T(0);
        lim[i] = (MOTOR != kMotorOff) ? max_force * dt : T(0);
    }
    {
        V3<T> va = mk<T>(0, 0, 0);
#pragma unroll
        for (int i = 0; i < N; ++i) va = va + v[i] * mk(Jt[0][i], Jt[1][i], Jt[2][i]);
        const V3<T> vpb = vb + cross(wb, rb);
        const V3<T> cv = va - vpb, gap = pa - pb;
        const T cvv[3] = {cv.x, cv.y, cv.z}, gp[3] = {gap.x, gap.y, gap.z};
#pragma unroll
        for (int x = 0; x < 3; ++x) { r[N + x] = (-bc.erp * gp[x] / dt) - cvv[x]; lim[N + x] = bc.max_impulse; }
    }
    T G[NR][NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const T jdi = T(1) / A[j][j];
        r[j] = r[j] * jdi;
        lam[j] = T(0);
#pragma unroll
        for (int i = 0; i < NR; ++i) G[j][i] = A[j][i] * jdi;
    }
    // same convergence exit as pgs_unclamped: with no row at its limit r -> 0 and lambda stops moving; a clamped row keeps its r_i
    // away from 0 and simply never triggers it
    T thr = T(0);
#pragma unroll
    for (int j = 0; j < NR; ++j) thr = tmax(thr, tabs(r[j]));
    thr = iters < 0 ? T(-1) : thr * (sizeof(T) == 8 ? T(1.3877787807814457e-17) : T(7.450580596923828e-09));
    const int n_it = iters < 0 ? -iters : iters;
    int conv_sweeps = -1;
    if (m.res_thr > T(0)) {                  // threshold mode (DevRobot::res_thr): Bullet's exit per env after every sweep, oracle mb_step_body
        bool live = n_it > 0;
        int ran = 0;
        auto row = [&](const int i, T& res) {
            const T t = r[i];
            const T sum = lam[i] + t;
            const T lo = sum < -lim[i] ? -lim[i] : sum;
            const T sc = lo > lim[i] ? lim[i] : lo;
            const T delta = live ? ((sc == sum) ? t : sc - lam[i]) : T(0);
            lam[i] = live ? sc : lam[i];
#pragma unroll
            for (int j = 0; j < NR; ++j) r[j] -= G[j][i] * delta;
            const T dvel = delta * A[i][i];
            res = tmax(res, dvel * dvel);
        };
        for (int it = 0; it < n_it && __any(live); ++it) {
            T res = T(0);
            if (it & 1) {
#pragma unroll
                for (int i = 0; i < NR; ++i) if (MOTOR != kMotorOff || i >= N) row(i, res);
            } else {
#pragma unroll
                for (int i = NR - 1; i >= 0; --i) if (MOTOR != kMotorOff || i >= N) row(i, res);
            }
            if (live) { ++ran; live = !(res <= m.res_thr); }
        }
        if (sweep_acc != nullptr) *sweep_acc += ran;
    } else
    for (int it = 0; it < n_it; ++it) {
        if ((it & 7) == 0 && it > 0) {
            T mx = T(0);
#pragma unroll
            for (int j = 0; j < NR; ++j) mx = tmax(mx, tabs(r[j]));
            if (__all(mx <= thr)) { conv_sweeps = it; break; }
        }
        if (it & 1) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const T t = r[i];
                const T sum = lam[i] + t;
                const T lo = sum < -lim[i] ? -lim[i] : sum;
                const T sc = lo > lim[i] ? lim[i] : lo;
                const T delta = (sc == sum) ? t : sc - lam[i];
                lam[i] = sc;
#pragma unroll
                for (int j = 0; j < NR; ++j) r[j] -= G[j][i] * delta;
            }
        } else {
#pragma unroll
            for (int i = NR - 1; i >= 0; --i) {
                const T t = r[i];
                const T sum = lam[i] + t;
                const T lo = sum < -lim[i] ? -lim[i] : sum;
                const T sc = lo > lim[i] ? lim[i] : lo;
                const T delta = (sc == sum) ? t : sc - lam[i];
                lam[i] = sc;
#pragma unroll
                for (int j = 0; j < NR; ++j) r[j] -= G[j][i] * delta;
            }
        }
    }
    // apply impulses
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < N; ++j) acc += Minv[i][j] * lam[j];
#pragma unroll
        for (int x = 0; x < 3; ++x) acc += Wa[i][x] * lam[N + x];
        qd[i] = v[i] + acc;
        q[i] += dt * qd[i];
    }
    if (verified != nullptr) *verified = (iters > 0 && conv_sweeps > 0 && 5 * conv_sweeps <= 4 * iters) ? 24 : 0;
    if (trig != nullptr) trig_init<T, N>(q, *trig);   // a full tick re-anchors the carried sines / cosines exactly
    const V3<T> lp = mk(lam[N], lam[N + 1], lam[N + 2]);
    b.v = vb - (T(1) / bc.mass) * lp;
    b.w = wb - mul(Iwi, cross(rb, lp));
    xc = xc + dt * b.v;
    integrate_rotation(b.R, b.w, dt);
    b.pos = xc - mul(b.R, bc.com);
}

// ------------------------------------------------------------------------------------------------ arm + plate + P2P + ball
// object_balance, object_mode "ball_on_plate" (object_balance_env.py:187-199, 241-260): the free body is the round plate, tied to the TCP as
// the pole is, and a ball (sphere.urdf, lateralFriction 10) rolls on it.  Restated contact model, identical to oracle/minibullet.c:
// mb_step_body_ball [PARITY_ASSUMPTIONS A39]: closest point of the plate's solid cylinder to the ball's centre, one contact while the gap is
// within the breaking distance; rows = motors and P2P (reverse order on even sweeps), then the normal (lambda >= 0), then the two friction
// rows together under the cone |lambda_t| <= mu lambda_n.  Same Delassus / residual form as sim_tick_body; the blocks that are structurally
// zero (motor rows x contact rows) are left out.  Always the full solve: the analytic fixed point of sim_tick_body has no contact rows.
template <typename T> __device__ __forceinline__ void plane_space(V3<T> n, V3<T>& t1, V3<T>& t2);   // btPlaneSpace1, below
template <typename T> struct Ball { V3<T> pos, v, w; };
template <typename T> struct BallConst { T radius, mass, inertia, mu, plate_radius, plate_half_len, breaking, erp, lin_damp, ang_damp; };
// object_balance, object_mode "spinning_plate" (object_balance_env.py:198-239): BodyConst describes the spool (the body on the constraint), this the
// dish standing on it and their contact (oracle/minibullet.h: mb_spin; tick: csrc/tg_spin.hip)
template <typename T> struct SpinConst {
    T mass; V3<T> com; S3<T> inertia;                    // the dish
    T margin, breaking, erp, mu, lin_damp, ang_damp;     // hull margin (both), contactBreakingThreshold, contact ERP, friction dish x spool, the spool's damping
    T buffer_height, embed0;                             // :202-203; the embed distance the spool's pivot was made with (:267-269: never updated)
    int n_dish, n_spool;                                 // hull vertex counts (State::spin_hulls); n_dish > 0 <=> this mode
};

template <typename T, int TOPO, int MOTOR>
__device__ __forceinline__ void sim_tick_body_ball(const DevRobot<T>& m, T (&q)[Topo<TOPO>::N], T (&qd)[Topo<TOPO>::N],
                                                   const T (&q_des)[Topo<TOPO>::N], const T (&qd_des)[Topo<TOPO>::N], T kp, T kd, T max_force, T dt,
                                                   int iters, V3<T> gravity, FreeBody<T>& b, const BodyConst<T>& bc, V3<T> pivot_b, Ball<T>& ball,
                                                   const BallConst<T>& kc, V3<T> ball_torque, bool torque_pending, T& normal_impulse,
                                                   int* sweep_acc = nullptr /* threshold mode: += the sweeps this tick ran */) {
    constexpr int N = Topo<TOPO>::N;
    constexpr int NP = N + 3;     // motor and P2P rows
    T hb[N], qdm[N], Minv[N][N], traceM;
    Kin<T, TOPO> kin;
    dynamics_terms<T, TOPO, false>(m, q, qd, hb, qdm, Minv, traceM, gravity, kin);
    T v[N];
    {
        T rhs[N];
#pragma unroll
        for (int i = 0; i < N; ++i) rhs[i] = qdm[i] - m.joint_damp * qd[i];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += Minv[i][j] * rhs[j];
            v[i] = qd[i] + dt * acc;
        }
    }
    // plate: gravity and the gyroscopic torque; no velocity damping (reset_object :338-345), no external force in this mode
    const S3<T> Iw = rotate(b.R, bc.inertia), Iwi = inverse(Iw);
    V3<T> xc = b.pos + mul(b.R, bc.com);
    const V3<T> vb = b.v + dt * gravity, wb = b.w + dt * mul(Iwi, mk<T>(0, 0, 0) - cross(b.w, mul(Iw, b.w)));
    // ball: gravity, Bullet's default damping F = -m v (K + K |v|) [A27], the one-shot torque of apply_random_torque_ball (:393-401)
    const T sv = kc.lin_damp + kc.lin_damp * norm(ball.v), sw = kc.ang_damp + kc.ang_damp * norm(ball.w);
    const V3<T> tq = torque_pending ? ball_torque : mk<T>(0, 0, 0);
    const V3<T> vk = ball.v + dt * (gravity - sv * ball.v), wk = ball.w + dt * ((T(1) / kc.inertia) * tq - sw * ball.w);
    // P2P geometry
    V3<T> pa; M3<T> Rl;
    {
        const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
        const T pav[3] = {bc.pivot_a.x, bc.pivot_a.y, bc.pivot_a.z};
        link_frame<T, TOPO>(kin, bc.link, pav, ident, pa, Rl);
    }
    const V3<T> pb = b.pos + mul(b.R, pivot_b);
    const V3<T> rb = pb - xc;
    T Jt[3][N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        bool on_path = false;
#pragma unroll
        for (int l = 0; l < N; ++l)
            if (l == bc.link && is_ancestor_or_self<TOPO>(i, l)) on_path = true;
        const V3<T> jt = cross(kin.a[i], pa - kin.o[i]);
        Jt[0][i] = on_path ? jt.x : T(0); Jt[1][i] = on_path ? jt.y : T(0); Jt[2][i] = on_path ? jt.z : T(0);
    }
    T Wa[N][3];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += Minv[i][j] * Jt[x][j];
            Wa[i][x] = acc;
        }
    const V3<T> e[3] = {mk<T>(1, 0, 0), mk<T>(0, 1, 0), mk<T>(0, 0, 1)};
    V3<T> rxe[3], Wang[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) { rxe[x] = cross(rb, e[x]); Wang[x] = mul(Iwi, rxe[x]); }
    const T im = T(1) / bc.mass;
    // contact: closest point of the plate's solid cylinder (plate frame: axis z, centred on the base frame) to the ball's centre
    bool touching = false;
    V3<T> d[3] = {mk<T>(0, 0, 1), mk<T>(1, 0, 0), mk<T>(0, 1, 0)}, rba = mk<T>(0, 0, 0), rpb = mk<T>(0, 0, 0);
    T depth = T(1e30);
    {
        const V3<T> dw = ball.pos - b.pos;
        const V3<T> p = mk(b.R.m[0] * dw.x + b.R.m[3] * dw.y + b.R.m[6] * dw.z, b.R.m[1] * dw.x + b.R.m[4] * dw.y + b.R.m[7] * dw.z,
                           b.R.m[2] * dw.x + b.R.m[5] * dw.y + b.R.m[8] * dw.z);
        const T rad = tsqrt(p.x * p.x + p.y * p.y);
        const T sr = rad > kc.plate_radius ? kc.plate_radius / rad : T(1);
        const V3<T> cl = mk(p.x * sr, p.y * sr, p.z > kc.plate_half_len ? kc.plate_half_len : (p.z < -kc.plate_half_len ? -kc.plate_half_len : p.z));
        const V3<T> g = p - cl;
        const T dist = tsqrt(dot(g, g));
        depth = dist - kc.radius;
        if (dist > T(0) && depth <= kc.breaking) {
            touching = true;
            const V3<T> gn = mk(g.x / dist, g.y / dist, g.z / dist);
            const V3<T> nw = mul(b.R, gn);                         // from the plate towards the ball
            d[0] = nw;
            plane_space(nw, d[1], d[2]);
            rba = mk<T>(0, 0, 0) - kc.radius * nw;                 // body A = the ball (+d)
            rpb = (b.pos + mul(b.R, cl)) - xc;                     // body B = the plate (-d)
        }
    }
    V3<T> cxa[3], cxb[3], Wcb[3];                                  // (rba x d), (rpb x d), Iw^-1 (rpb x d)
#pragma unroll
    for (int r = 0; r < 3; ++r) { cxa[r] = cross(rba, d[r]); cxb[r] = cross(rpb, d[r]); Wcb[r] = mul(Iwi, cxb[r]); }
    // Delassus blocks: App (motor + P2P rows, as sim_tick_body), Apc (P2P x contact), Acc (contact x contact)
    T A[NP][NP];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) A[i][j] = Minv[i][j];
#pragma unroll
        for (int x = 0; x < 3; ++x) { A[i][N + x] = Wa[i][x]; A[N + x][i] = Wa[i][x]; }
    }
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            T acc = (x == y) ? im : T(0);
#pragma unroll
            for (int i = 0; i < N; ++i) acc += Jt[x][i] * Wa[i][y];
            A[N + x][N + y] = acc + dot(rxe[x], Wang[y]);
        }
    T Apc[3][3], Acc[3][3];
    const T imk = T(1) / kc.mass, iik = T(1) / kc.inertia;
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int r = 0; r < 3; ++r) Apc[x][r] = im * dot(e[x], d[r]) + dot(rxe[x], Wcb[r]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) Acc[r][s] = (im + imk) * dot(d[r], d[s]) + dot(cxb[r], Wcb[s]) + iik * dot(cxa[r], cxa[s]);
    // right-hand sides and limits
    T r[NP], lim[NP], lam[NP], rc[3], lc[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const T pos_term = (MOTOR == kMotorPosition) ? kp * (q_des[i] - q[i]) / dt : T(0);
        const T des = pos_term + v[i] + kd * (qd_des[i] - v[i]);
        r[i] = (MOTOR != kMotorOff) ? des - v[i] : T(0);
        lim[i] = (MOTOR != kMotorOff) ? max_force * dt : T(0);
    }
    {
        V3<T> va = mk<T>(0, 0, 0);
#pragma unroll
        for (int i = 0; i < N; ++i) va = va + v[i] * mk(Jt[0][i], Jt[1][i], Jt[2][i]);
        const V3<T> cv = va - (vb + cross(wb, rb)), gap = pa - pb;
        const T cvv[3] = {cv.x, cv.y, cv.z}, gp[3] = {gap.x, gap.y, gap.z};
#pragma unroll
        for (int x = 0; x < 3; ++x) { r[N + x] = (-bc.erp * gp[x] / dt) - cvv[x]; lim[N + x] = bc.max_impulse; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T rv = (dot(d[k], vk) + dot(cxa[k], wk)) - (dot(d[k], vb) + dot(cxb[k], wb));
        rc[k] = (k == 0) ? (depth > T(0) ? (-rv - depth / dt) : (-depth * kc.erp / dt - rv)) : -rv;     // restitution 0
    }
    // Unscaled residuals R_j = rhs_j - sum_q A_jq lambda_q and the inverse diagonals: a row's step is t = R_i / A_ii, and every update reads A
    // itself - one triangle of the symmetric blocks (App upper, Acc upper, Apc shared by the P2P and the contact rows) instead of the four
    // row-scaled copies G = A / diag: 60 instead of 117 matrix doubles live through the sweeps (this tick does not fit the register file:
    // 872 B of scratch per lane with the scaled copies).
    T jdi[NP], jc[3];
#pragma unroll
    for (int j = 0; j < NP; ++j) { jdi[j] = T(1) / A[j][j]; lam[j] = T(0); }
#pragma unroll
    for (int k = 0; k < 3; ++k) jc[k] = touching ? T(1) / Acc[k][k] : T(0);
    T thr = T(0);
#pragma unroll
    for (int j = 0; j < NP; ++j) thr = tmax(thr, tabs(r[j] * jdi[j]));
    if (touching) thr = tmax(thr, tmax(tabs(rc[0] * jc[0]), tmax(tabs(rc[1] * jc[1]), tabs(rc[2] * jc[2]))));
    thr = iters < 0 ? T(-1) : thr * (sizeof(T) == 8 ? T(1.3877787807814457e-17) : T(7.450580596923828e-09));
    const int n_it = iters < 0 ? -iters : iters;
    if (m.res_thr > T(0)) {                  // threshold mode (DevRobot::res_thr): Bullet's exit per env after every sweep, oracle mb_step_body_ball
        bool live = n_it > 0;
        int ran = 0;
        for (int it = 0; it < n_it && __any(live); ++it) {
            T res = T(0);
            auto rowt = [&](const int i) {
                const T t = r[i] * jdi[i];
                const T sum = lam[i] + t;
                const T lo = sum < -lim[i] ? -lim[i] : sum;
                const T sc = lo > lim[i] ? lim[i] : lo;
                const T delta = live ? ((sc == sum) ? t : sc - lam[i]) : T(0);
                lam[i] = live ? sc : lam[i];
#pragma unroll
                for (int j = 0; j < NP; ++j) r[j] -= (j <= i ? A[j][i] : A[i][j]) * delta;
                if (i >= N) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) rc[k] -= Apc[i >= N ? i - N : 0][k] * delta;
                }
                const T dvel = delta * A[i][i];
                res = tmax(res, dvel * dvel);
            };
            if (it & 1) {
#pragma unroll
                for (int i = 0; i < NP; ++i) if (MOTOR != kMotorOff || i >= N) rowt(i);
            } else {
#pragma unroll
                for (int i = NP - 1; i >= 0; --i) if (MOTOR != kMotorOff || i >= N) rowt(i);
            }
            if (touching) {
                {   // normal
                    const T t = rc[0] * jc[0], sum = lc[0] + t;
                    const T sc = sum < T(0) ? T(0) : sum;
                    const T delta = live ? ((sc == sum) ? t : sc - lc[0]) : T(0);
                    lc[0] = live ? sc : lc[0];
#pragma unroll
                    for (int x = 0; x < 3; ++x) r[N + x] -= Apc[x][0] * delta;
#pragma unroll
                    for (int k = 0; k < 3; ++k) rc[k] -= Acc[0][k] * delta;
                    const T dvel = delta * Acc[0][0];
                    res = tmax(res, dvel * dvel);
                }
                {   // friction pair, cone: ONE residual, the sum of the pair's two velocity changes (resolveConeFrictionConstraintRows, A7c)
                    const T limit = kc.mu * lc[0];
                    T s1 = lc[1] + rc[1] * jc[1], s2 = lc[2] + rc[2] * jc[2];
                    const T tot = tsqrt(s1 * s1 + s2 * s2);
                    if (tot > limit) { const T f = tot > T(0) ? limit / tot : T(0); s1 *= f; s2 *= f; }
                    const T d1 = live ? s1 - lc[1] : T(0), d2 = live ? s2 - lc[2] : T(0);
                    if (live) { lc[1] = s1; lc[2] = s2; }
#pragma unroll
                    for (int x = 0; x < 3; ++x) r[N + x] -= Apc[x][1] * d1 + Apc[x][2] * d2;
                    rc[0] -= Acc[0][1] * d1 + Acc[0][2] * d2;
                    rc[1] -= Acc[1][1] * d1 + Acc[1][2] * d2;
                    rc[2] -= Acc[1][2] * d1 + Acc[2][2] * d2;
                    const T dvel = d1 * Acc[1][1] + d2 * Acc[2][2];
                    res = tmax(res, dvel * dvel);
                }
            }
            if (live) { ++ran; live = !(res <= m.res_thr); }
        }
        if (sweep_acc != nullptr) *sweep_acc += ran;
    } else
    for (int it = 0; it < n_it; ++it) {
        if ((it & 7) == 0 && it > 0) {
            T mx = T(0);
#pragma unroll
            for (int j = 0; j < NP; ++j) mx = tmax(mx, tabs(r[j] * jdi[j]));
            if (touching) mx = tmax(mx, tmax(tabs(rc[0] * jc[0]), tmax(tabs(rc[1] * jc[1]), tabs(rc[2] * jc[2]))));
            if (__all(mx <= thr)) break;
        }
        auto row = [&](const int i) {      // i is a compile-time constant after unrolling
            const T t = r[i] * jdi[i];
            const T sum = lam[i] + t;
            const T lo = sum < -lim[i] ? -lim[i] : sum;
            const T sc = lo > lim[i] ? lim[i] : lo;
            const T delta = (sc == sum) ? t : sc - lam[i];
            lam[i] = sc;
#pragma unroll
            for (int j = 0; j < NP; ++j) r[j] -= (j <= i ? A[j][i] : A[i][j]) * delta;
            if (i >= N) {
#pragma unroll
                for (int k = 0; k < 3; ++k) rc[k] -= Apc[i >= N ? i - N : 0][k] * delta;
            }
        };
        if (it & 1) {
#pragma unroll
            for (int i = 0; i < NP; ++i) row(i);
        } else {
#pragma unroll
            for (int i = NP - 1; i >= 0; --i) row(i);
        }
        if (touching) {
            {   // normal
                const T t = rc[0] * jc[0], sum = lc[0] + t;
                const T sc = sum < T(0) ? T(0) : sum;
                const T delta = (sc == sum) ? t : sc - lc[0];
                lc[0] = sc;
#pragma unroll
                for (int x = 0; x < 3; ++x) r[N + x] -= Apc[x][0] * delta;
#pragma unroll
                for (int k = 0; k < 3; ++k) rc[k] -= Acc[0][k] * delta;
            }
            {   // friction pair, cone (enableConeFriction = 1, base_tactile_env.py:128-130)
                const T limit = kc.mu * lc[0];
                T s1 = lc[1] + rc[1] * jc[1], s2 = lc[2] + rc[2] * jc[2];
                const T tot = tsqrt(s1 * s1 + s2 * s2);
                if (tot > limit) { const T f = tot > T(0) ? limit / tot : T(0); s1 *= f; s2 *= f; }
                const T d1 = s1 - lc[1], d2 = s2 - lc[2];
                lc[1] = s1; lc[2] = s2;
#pragma unroll
                for (int x = 0; x < 3; ++x) r[N + x] -= Apc[x][1] * d1 + Apc[x][2] * d2;
                rc[0] -= Acc[0][1] * d1 + Acc[0][2] * d2;
                rc[1] -= Acc[1][1] * d1 + Acc[1][2] * d2;
                rc[2] -= Acc[1][2] * d1 + Acc[2][2] * d2;
            }
        }
    }
    normal_impulse = touching ? lc[0] : T(0);
    // apply impulses
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < N; ++j) acc += Minv[i][j] * lam[j];
#pragma unroll
        for (int x = 0; x < 3; ++x) acc += Wa[i][x] * lam[N + x];
        qd[i] = v[i] + acc;
        q[i] += dt * qd[i];
    }
    const V3<T> lp = mk(lam[N], lam[N + 1], lam[N + 2]);
    V3<T> jl = mk<T>(0, 0, 0), jab = mk<T>(0, 0, 0), jaa = mk<T>(0, 0, 0);     // sum of d lambda, of (rpb x d) lambda, of (rba x d) lambda
    if (touching) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { jl = jl + lc[k] * d[k]; jab = jab + lc[k] * cxb[k]; jaa = jaa + lc[k] * cxa[k]; }
    }
    b.v = vb - im * (lp + jl);
    b.w = wb - mul(Iwi, cross(rb, lp) + jab);
    xc = xc + dt * b.v;
    integrate_rotation(b.R, b.w, dt);
    b.pos = xc - mul(b.R, bc.com);
    ball.v = vk + imk * jl;
    ball.w = wk + iik * jaa;
    ball.pos = ball.pos + dt * ball.v;
}

// ------------------------------------------------------------------------------------------------ arm + cube + contacts
// object_push: a free cube on the table, pushed by the collision core of the sensor tip (object_push_env.py:196-227 with the
// tip core enabled, tactile_sensor.py:58-67).  Restated contact model [PARITY_ASSUMPTIONS A23-A28], identical to
// oracle/minibullet.c:mb_step_push:
//   cube-table   the cube vertices within the breaking distance of the plane, at most 4, rows in vertex order, normal +z
//   cube-tip     the hull vertex of the tip core with the smallest signed distance to the box; one soft contact
//                (contactStiffness / contactDamping -> cfm, erp)
//   rows         joint motors (reverse order on even sweeps), then all contact normals (lambda >= 0), then per contact the two
//                friction rows with the cone |lambda_t| <= mu lambda_n; `iters` sweeps, no warm start
// Storage: the joint-space part (Minv, tip rows' arm Jacobian) stays in registers; the per-contact data lives in a
// lane-private LDS column (word k of lane l at L[k * 64]) because 23 rows do not fit the VGPR file.
template <typename T> struct PushScene {
    T table_z, half[3], mu_table, mu_tip, margin_cube, margin_tip, breaking, erp, tip_stiffness, tip_damping, lin_damp, ang_damp;
    T com[3], inertia0[6], mass0;
    int tip_link, n_tip, cone_friction;
    int narrow;   // tg_config.narrowphase: 0 closed forms; 1 GJK / EPA + persistent manifold; 2 GJK / EPA, the tick's point only (tg_narrowphase.hpp)
    // object_roll (SHAPE = 1): the free body is a sphere of radius `radius` x scale (inertia0 = that of radius0), the tip collision shape a
    // solid cylinder (axis = local z) given in the frame of tip_link (ur5_with_flat_tactip.urdf:320-325) [PARITY_ASSUMPTIONS A30]
    T radius0, cyl_pos[3], cyl_hl, cyl_r;
    M3<T> cyl_rot;
};
constexpr int kPushTab = 9;                        // staging words per table contact slot: ra[3], rhs[3], 1/A[3]
constexpr int kPushLdsWords = 4 * kPushTab;
template <typename T> using lds_ptr = __attribute__((address_space(3))) T*;

// friction limit of one contact: cone (enableConeFriction=1, base_tactile_env.py:128-130) or pyramid
template <typename T> __device__ __forceinline__ void friction_clamp(T& s1, T& s2, T limit, bool cone) {
    // Branch-free on purpose: any branch here splits the sweep into many scheduling regions and pins every LDS read right in front of
    // its use.  Both limits are evaluated and selected.
    const T tot2 = s1 * s1 + s2 * s2;
    const bool over = tot2 > limit * limit;
    T r = trsqrt(tmax(tot2, T(1e-30)));    // tot2 == 0 is never `over`
    asm volatile("" : "+v"(r));             // keep it unconditional: the optimiser would otherwise branch around the rsq
    const T f = over ? limit * r : T(1);
    const T p1 = tmin(tmax(s1, -limit), limit), p2 = tmin(tmax(s2, -limit), limit);
    s1 = cone ? s1 * f : p1;
    s2 = cone ? s2 * f : p2;
}
// btPlaneSpace1
template <typename T> __device__ __forceinline__ void plane_space(V3<T> n, V3<T>& t1, V3<T>& t2) {
    if (tabs(n.z) > T(0.7071067811865475244)) {
        const T a = n.y * n.y + n.z * n.z, k = T(1) / tsqrt(a);
        t1 = mk(T(0), -n.z * k, n.y * k);
        t2 = mk(a * k, -n.x * t1.z, n.x * t1.y);
    } else {
        const T a = n.x * n.x + n.y * n.y, k = T(1) / tsqrt(a);
        t1 = mk(-n.y * k, n.x * k, T(0));
        t2 = mk(-n.z * t1.y, n.z * t1.x, a * k);
    }
}

// SHAPE 0: box (object_push; `mass` = the episode's mass, inertia rescaled with it).  SHAPE 1: sphere (object_roll; `mass` carries the
// episode's RADIUS instead - the mass is fixed - and the inertia scales with its square).
template <typename T, int TOPO, int MOTOR, int SHAPE = 0>
__device__ __noinline__ void sim_tick_push(const DevRobot<T>& m, T (&q)[Topo<TOPO>::N], T (&qd)[Topo<TOPO>::N],
                                              const T (&q_des)[Topo<TOPO>::N], const T (&qd_des)[Topo<TOPO>::N], T kp, T kd, T max_force, T dt,
                                              int iters, FreeBody<T>& b, const PushScene<T>& sc, const T* __restrict__ tip_verts, T mass,
                                              lds_ptr<T> L, int& contact_code, int* sweep_acc = nullptr /* threshold mode: += the sweeps this tick ran */) {
    // contact_code (out): the tick's contact pairs - bits 0-7 the cube vertices kept as cube-table contacts (SHAPE 1: bit 0 = the
    // sphere-table contact), bit 8 the tip contact, bits 9+ the hull vertex of the tip core that made it (tg_state_view.contact_ids)
    constexpr int N = Topo<TOPO>::N;
    asm volatile("" ::: "memory");
    T hb[N], qdm[N], Minv[N][N], traceM;
    Kin<T, TOPO> kin;
    const V3<T> gravity = load_v3(m.gravity);
    dynamics_terms<T, TOPO, false>(m, q, qd, hb, qdm, Minv, traceM, gravity, kin);
    T v[N];
    {
        T rhs[N];
#pragma unroll
        for (int i = 0; i < N; ++i) rhs[i] = qdm[i] - m.joint_damp * qd[i];   // gravity compensation cancels hbias analytically
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += Minv[i][j] * rhs[j];
            v[i] = qd[i] + dt * acc;
        }
    }
    // ---- cube: gravity, Bullet's velocity damping, gyroscopic torque (changeDynamics(mass) rescales the inertia with the mass)
    const T radius = mass;   // SHAPE 1 only
    const T iscale = SHAPE == 1 ? (radius / sc.radius0) * (radius / sc.radius0) : mass / sc.mass0, invm = T(1) / (SHAPE == 1 ? sc.mass0 : mass);
    const S3<T> I0{sc.inertia0[0] * iscale, sc.inertia0[1] * iscale, sc.inertia0[2] * iscale, sc.inertia0[3] * iscale, sc.inertia0[4] * iscale,
                   sc.inertia0[5] * iscale};
    const S3<T> Iw = rotate(b.R, I0), Iwi = inverse(Iw);
    V3<T> xc = b.pos + mul(b.R, load_v3(sc.com));
    V3<T> vb, wb;
    {
        const T sv = sc.lin_damp + sc.lin_damp * norm(b.v), sw = sc.ang_damp + sc.ang_damp * norm(b.w);
        const V3<T> Iwv = mul(Iw, b.w);
        const V3<T> Nt = (-sw) * Iwv - cross(b.w, Iwv);
        vb = b.v + dt * (gravity - sv * b.v);
        wb = b.w + dt * mul(Iwi, Nt);
    }
    // ---- cube - table contacts.  LDS is only the staging area that turns the per-lane contact count into fixed slots: each kept
    // vertex writes (ra, rhs[3], 1/A[3]) to the next slot of its lane's column, then the four slots are read back into registers.
#pragma unroll
    for (int k = 0; k < 4 * kPushTab; ++k) L[k * 64] = T(0);
    if constexpr (SHAPE == 1) {   // sphere - table: its lowest point, one slot
        const T vz0 = (b.pos.z - radius) - sc.table_z;
        contact_code = (vz0 <= sc.breaking) ? 1 : 0;
        if (vz0 <= sc.breaking) {
            const V3<T> ra = mk(T(0), T(0), -radius);
            lds_ptr<T> S = L;
            S[0] = ra.x; S[64] = ra.y; S[128] = ra.z;
            const V3<T> Ja[3] = {mk(ra.y, -ra.x, T(0)), mk(ra.z, T(0), -ra.x), mk(T(0), ra.z, -ra.y)};
            const T lin[3] = {vb.z, -vb.y, vb.x};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const T A = invm + dot(Ja[r], mul(Iwi, Ja[r]));
                const T rv = lin[r] + dot(Ja[r], wb);
                T rhs = -rv;
                if (r == 0) rhs = (vz0 > T(0)) ? (-rv - vz0 / dt) : (-vz0 * sc.erp / dt - rv);
                S[(3 + r) * 64] = rhs;
                S[(6 + r) * 64] = T(1) / A;
            }
        }
    } else {
        T vz[8];
        int keep = 0, cnt = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const T lx = (c & 4) ? sc.half[0] : -sc.half[0], ly = (c & 2) ? sc.half[1] : -sc.half[1], lz = (c & 1) ? sc.half[2] : -sc.half[2];
            vz[c] = (b.pos.z + (b.R.m[6] * lx + b.R.m[7] * ly + b.R.m[8] * lz)) - sc.table_z;
            if (vz[c] <= sc.breaking) { keep |= 1 << c; ++cnt; }
        }
        while (cnt > 4) {   // manifold capacity: drop the shallowest (ties: the higher index)
            int worst = -1; T wz = T(0);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (((keep >> c) & 1) && (worst < 0 || vz[c] >= wz)) { worst = c; wz = vz[c]; }
            keep &= ~(1 << worst); --cnt;
        }
        contact_code = keep;
        int slot = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if ((keep >> c) & 1) {
                const T lx = (c & 4) ? sc.half[0] : -sc.half[0], ly = (c & 2) ? sc.half[1] : -sc.half[1], lz = (c & 1) ? sc.half[2] : -sc.half[2];
                const V3<T> ra = (b.pos + mul(b.R, mk(lx, ly, lz))) - xc;
                lds_ptr<T> S = L + (slot * kPushTab) * 64;
                S[0] = ra.x; S[64] = ra.y; S[128] = ra.z;
                // rows: n = (0,0,1), t1 = (0,-1,0), t2 = (1,0,0)  (btPlaneSpace1 of +z); J = [d, ra x d]
                const V3<T> Ja[3] = {mk(ra.y, -ra.x, T(0)), mk(ra.z, T(0), -ra.x), mk(T(0), ra.z, -ra.y)};
                const T lin[3] = {vb.z, -vb.y, vb.x};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const T A = invm + dot(Ja[r], mul(Iwi, Ja[r]));
                    const T rv = lin[r] + dot(Ja[r], wb);
                    T rhs = -rv;
                    if (r == 0) rhs = (vz[c] > T(0)) ? (-rv - vz[c] / dt) : (-vz[c] * sc.erp / dt - rv);
                    S[(3 + r) * 64] = rhs;
                    S[(6 + r) * 64] = T(1) / A;
                }
                ++slot;
            }
        }
    }
    T tra[4][3], trhs[4][3], tjdi[4][3], tlam[4][3];   // table contact slots (empty slots: all zero -> their rows never move)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tra[c][k] = L[(c * kPushTab + k) * 64];
            trhs[c][k] = L[(c * kPushTab + 3 + k) * 64];
            tjdi[c][k] = L[(c * kPushTab + 6 + k) * 64];
            tlam[c][k] = T(0);
        }
    // ---- cube - tip core contact: hull vertex with the smallest signed distance to the box
    constexpr int NP = Topo<TOPO>::NP;   // joints that can carry the tip (validated on the host)
    T Jt[3][NP], Wa[N][3];
    V3<T> pd[3], prb;                   // tip rows, cube part: row directions (n, t1, t2) and the contact arm; J = [-d, -(rb x d)]
    T prhs[3], pjdi[3], plam[3], pcfm = T(0);
    {
        V3<T> ol; M3<T> Rl;
        {
            const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
            const T z3[3] = {T(0), T(0), T(0)};
            link_frame<T, TOPO>(kin, sc.tip_link, z3, ident, ol, Rl);
        }
        T depth; bool active; V3<T> nrm, pa, pb;
        if constexpr (SHAPE == 1) {   // sphere - tip: closest point of the solid cylinder to the sphere centre
            const V3<T> cw = ol + mul(Rl, load_v3(sc.cyl_pos));
            const M3<T> Rw = mul(Rl, sc.cyl_rot);
            const V3<T> p = mulT(Rw, b.pos - cw);                     // centre in the cylinder frame
            const T rad = tsqrt(p.x * p.x + p.y * p.y);
            const T sr = rad > sc.cyl_r ? sc.cyl_r / rad : T(1);
            const V3<T> cl = mk(p.x * sr, p.y * sr, p.z > sc.cyl_hl ? sc.cyl_hl : (p.z < -sc.cyl_hl ? -sc.cyl_hl : p.z));
            const V3<T> g = p - cl;
            const T dist = tsqrt(dot(g, g));
            depth = dist - radius;
            active = dist > T(0) && depth <= sc.breaking;
            contact_code |= active ? (1 << 8) : 0;
            const T idist = T(1) / (dist > T(0) ? dist : T(1));
            V3<T> gw = mul(Rw, idist * g);                            // from the cylinder towards the sphere
            if (!(dist > T(0))) gw = mk<T>(0, 0, 1);                  // centre inside the cylinder: no contact (as the oracle), but the disabled rows below still need a finite frame
            nrm = mk<T>(0, 0, 0) - gw;                                // contact normal: from the sphere (body B) towards the tip (body A)
            pa = cw + mul(Rw, cl);
            pb = b.pos - radius * gw;
        } else {
        M3<T> Mr;   // cube <- link rotation  Rc^T Rl
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Mr.m[3 * i + j] = b.R.m[i] * Rl.m[j] + b.R.m[3 + i] * Rl.m[3 + j] + b.R.m[6 + i] * Rl.m[6 + j];
        const V3<T> tr = mulT(b.R, ol - b.pos);
        T best_key = T(1e30); int best_i = 0;
        // the hull is the same for every lane: a wave-uniform pointer turns the vertex fetch into scalar loads
        typedef const T __attribute__((address_space(4))) * uniform_ptr;   // constant address space: s_load for uniform addresses
        const uniform_ptr tv = (uniform_ptr)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)tip_verts >> 32)) << 32) |
                                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)tip_verts));
        const int n_tip = __builtin_amdgcn_readfirstlane(sc.n_tip);
#pragma unroll 4
        for (int i = 0; i < n_tip; ++i) {
            const V3<T> vv = mk(tv[3 * i], tv[3 * i + 1], tv[3 * i + 2]);
            const V3<T> p = tr + mul(Mr, vv);
            const T qx = tabs(p.x) - sc.half[0], qy = tabs(p.y) - sc.half[1], qz = tabs(p.z) - sc.half[2];
            const T ox = tmax(qx, T(0)), oy = tmax(qy, T(0)), oz = tmax(qz, T(0));
            const T s2 = ox * ox + oy * oy + oz * oz;
            const T mq = tmax(tmax(qx, qy), qz);
            const T key = s2 > T(0) ? s2 : mq;         // outside: squared distance (> 0); inside: largest face distance (<= 0)
            if (key < best_key) { best_key = key; best_i = i; }
        }
        const V3<T> vv = mk(tip_verts[3 * best_i], tip_verts[3 * best_i + 1], tip_verts[3 * best_i + 2]);
        const V3<T> w = ol + mul(Rl, vv);
        const V3<T> p = mulT(b.R, w - b.pos);
        const T pp[3] = {p.x, p.y, p.z};
        T qq[3], oo[3], g[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int x = 0; x < 3; ++x) { qq[x] = tabs(pp[x]) - sc.half[x]; oo[x] = qq[x] > T(0) ? qq[x] : T(0); }
        const T outside = tsqrt(oo[0] * oo[0] + oo[1] * oo[1] + oo[2] * oo[2]);
        T sdf;
        if (outside > T(0)) {
            sdf = outside;
#pragma unroll
            for (int x = 0; x < 3; ++x) g[x] = oo[x] / outside * (pp[x] < T(0) ? T(-1) : T(1));
        } else {
            const int ax = (qq[0] >= qq[1] && qq[0] >= qq[2]) ? 0 : ((qq[1] >= qq[2]) ? 1 : 2);
            sdf = ax == 0 ? qq[0] : (ax == 1 ? qq[1] : qq[2]);
#pragma unroll
            for (int x = 0; x < 3; ++x) g[x] = (x == ax) ? (pp[x] < T(0) ? T(-1) : T(1)) : T(0);
        }
        depth = sdf - (sc.margin_tip + sc.margin_cube);
        active = depth <= sc.breaking;
        contact_code |= active ? ((1 << 8) | (best_i << 9)) : 0;
        nrm = mul(b.R, mk(g[0], g[1], g[2]));   // from the cube towards the tip
        pa = w - sc.margin_tip * nrm; pb = w - (sdf - sc.margin_cube) * nrm;
        }
        V3<T> t1, t2;
        plane_space(nrm, t1, t2);
        const V3<T> dirs[3] = {nrm, t1, t2};
        const V3<T> rb = pb - xc;
        V3<T> jt[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            bool on_path = false;
#pragma unroll
            for (int l = 0; l < N; ++l)
                if (l == sc.tip_link && is_ancestor_or_self<TOPO>(i, l)) on_path = true;
            const V3<T> c = cross(kin.a[i], pa - kin.o[i]);
            jt[i] = on_path ? c : mk<T>(0, 0, 0);
        }
        const T denom = dt * sc.tip_stiffness + sc.tip_damping;   // soft contact: cfm = 1 / (dt (dt k + d)), erp = dt k / (dt k + d)
        const T cfm = (T(1) / denom) / dt, erp_c = dt * sc.tip_stiffness / denom;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const V3<T> d = dirs[r];
#pragma unroll
            for (int i = 0; i < NP; ++i) Jt[r][i] = dot(jt[i], d);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                T acc = T(0);
#pragma unroll
                for (int j = 0; j < NP; ++j) acc += Minv[i][j] * Jt[r][j];
                Wa[i][r] = acc;
            }
            const V3<T> Jl = mk<T>(0, 0, 0) - d, Ja = mk<T>(0, 0, 0) - cross(rb, d);
            const V3<T> Wg = mul(Iwi, Ja);
            T A = invm + dot(Ja, Wg), rv = dot(Jl, vb) + dot(Ja, wb);
#pragma unroll
            for (int i = 0; i < NP; ++i) { A += Jt[r][i] * Wa[i][r]; rv += Jt[r][i] * v[i]; }
            T rhs = -rv, jdi = T(1) / A;
            if (r == 0) {
                rhs = (depth > T(0)) ? (-rv - depth / dt) : (-depth * erp_c / dt - rv);
                jdi = T(1) / (A + cfm);
                pcfm = active ? cfm * jdi : T(0);
            }
            pd[r] = d; prb = rb;
            prhs[r] = active ? rhs : T(0);
            pjdi[r] = active ? jdi : T(0);
            plam[r] = T(0);
        }
    }
    // ---- motor rows; Minv keeps only its upper triangle from here on (Ms[tri(i, j)], i <= j)
    constexpr int NT = N * (N + 1) / 2;
    T Ms[NT], rm[N], jm[N], lm[N], dv[N];
    const T maximp = max_force * dt;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const T pos_term = (MOTOR == kMotorPosition) ? kp * (q_des[i] - q[i]) / dt : T(0);
        const T des = pos_term + v[i] + kd * (qd_des[i] - v[i]);
        rm[i] = des - v[i];
        jm[i] = T(1) / Minv[i][i];
        lm[i] = T(0); dv[i] = T(0);
#pragma unroll
        for (int j = i; j < N; ++j) Ms[i * N - i * (i - 1) / 2 + (j - i)] = Minv[i][j];
    }
#define TG_MS(i, j) Ms[((i) <= (j)) ? ((i) * N - (i) * ((i) - 1) / 2 + ((j) - (i))) : ((j) * N - (j) * ((j) - 1) / 2 + ((i) - (j)))]
    V3<T> dvl = mk<T>(0, 0, 0), dva = mk<T>(0, 0, 0);
    // All solver data is register resident (what does not fit the 256 architectural VGPRs is parked in AGPRs by the compiler: two
    // v_accvgpr_read per use, no memory latency).  The angular response of a table row is recomputed from Iw^-1 and ra instead of
    // being stored: six FMAs are cheaper than six AGPR reads and bring the footprint under 512 registers, i.e. no scratch.
    const T mu_table = sc.mu_table, mu_tip = sc.mu_tip;
    const bool cone = sc.cone_friction != 0;
    const V3<T> iw0 = mk(Iwi.xx, Iwi.xy, Iwi.xz), iw1 = mk(Iwi.xy, Iwi.yy, Iwi.yz), iw2 = mk(Iwi.xz, Iwi.yz, Iwi.zz);   // columns
    const int n_it = iters < 0 ? -iters : iters;   // contact problems do not reach their fixed point within 150 sweeps: no exit test
    // One sweep.  THR (threshold mode, DevRobot::res_thr > 0): the lane's impulses move only while its env is `live`, and `over` collects
    // "some row update of this sweep changed its row's velocity by more than sqrt(res_thr)" - Bullet's residual rule (oracle mb_step_push)
    // with the division by jacDiagABInv multiplied out:  (delta / jdi)^2 <= thr  <=>  delta^2 <= thr jdi^2  (empty slots: 0 <= 0), and for
    // a cone pair  (d1 / j1 + d2 / j2)^2 <= thr  <=>  (d1 j2 + d2 j1)^2 <= thr (j1 j2)^2.  THR false: the sweep as it has always been.
    const T rthr = m.res_thr;
    auto sweep = [&](auto thr_tag, const int it, const bool live, bool& over) {
        constexpr bool THR = decltype(thr_tag)::value;
        auto note = [&](const T delta, const T jdi) { if constexpr (THR) over = over || !(delta * delta <= rthr * (jdi * jdi)); };
        auto note2 = [&](const T d1, const T d2, const T j1, const T j2) {
            if constexpr (THR) {
                if (cone) { const T u = d1 * j2 + d2 * j1, w = j1 * j2; over = over || !(u * u <= rthr * (w * w)); }
                else { note(d1, j1); note(d2, j2); }
            }
        };
        // joint motors
        auto motor = [&](const int i) {
            const T sum = lm[i] + (rm[i] - dv[i]) * jm[i];
            const T scl = tmin(tmax(sum, -maximp), maximp);
            const T delta = (THR && !live) ? T(0) : scl - lm[i];
            lm[i] = (THR && !live) ? lm[i] : scl;
#pragma unroll
            for (int j = 0; j < N; ++j) dv[j] += TG_MS(j, i) * delta;
            note(delta, jm[i]);
        };
        if (MOTOR != kMotorOff || !THR) {
            if (it & 1) {
#pragma unroll
                for (int i = 0; i < N; ++i) motor(i);
            } else {
#pragma unroll
                for (int i = N - 1; i >= 0; --i) motor(i);
            }
        }
        // contact normals: table slots, then the tip
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const T rax = tra[c][0], ray = tra[c][1];
            const T jdv = dvl.z + (ray * dva.x - rax * dva.y);
            const T nl = tmax(tlam[c][0] + (trhs[c][0] - jdv) * tjdi[c][0], T(0));
            const T delta = (THR && !live) ? T(0) : nl - tlam[c][0];
            tlam[c][0] = (THR && !live) ? tlam[c][0] : nl;
            dvl.z += invm * delta;
            dva = dva + delta * (ray * iw0 - rax * iw1);                 // Iw^-1 (ra x n),  ra x n = (ra.y, -ra.x, 0)
            note(delta, tjdi[c][0]);
        }
        {   // the angular parts are rebuilt from (rb, d) each time: as many instructions as fetching them from AGPRs, 15 values fewer live
            const V3<T> ja = cross(pd[0], prb);                          // -(rb x n)
            T jdv = dot(ja, dva) - dot(pd[0], dvl);
#pragma unroll
            for (int i = 0; i < NP; ++i) jdv += Jt[0][i] * dv[i];
            const T nl = tmax(plam[0] + ((prhs[0] - jdv) * pjdi[0] - plam[0] * pcfm), T(0));
            const T delta = (THR && !live) ? T(0) : nl - plam[0];
            plam[0] = (THR && !live) ? plam[0] : nl;
#pragma unroll
            for (int i = 0; i < N; ++i) dv[i] += Wa[i][0] * delta;
            dvl = dvl - (invm * delta) * pd[0];
            dva = dva + delta * mul(Iwi, ja);
            note(delta, pjdi[0]);
        }
        // friction: table slots, then the tip
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const T rax = tra[c][0], ray = tra[c][1], raz = tra[c][2];
            const T limit = mu_table * tlam[c][0];
            const T jdv1 = -dvl.y + (raz * dva.x - rax * dva.z), jdv2 = dvl.x + (raz * dva.y - ray * dva.z);
            T s1 = tlam[c][1] + (trhs[c][1] - jdv1) * tjdi[c][1], s2 = tlam[c][2] + (trhs[c][2] - jdv2) * tjdi[c][2];
            friction_clamp(s1, s2, limit, cone);
            if (THR && !live) { s1 = tlam[c][1]; s2 = tlam[c][2]; }
            const T d1 = s1 - tlam[c][1], d2 = s2 - tlam[c][2];
            tlam[c][1] = s1; tlam[c][2] = s2;
            dvl.y -= invm * d1; dvl.x += invm * d2;
            // Iw^-1 (ra x t1) d1 + Iw^-1 (ra x t2) d2,  ra x t1 = (ra.z, 0, -ra.x),  ra x t2 = (0, ra.z, -ra.y)
            dva = dva + (raz * d1) * iw0 + (raz * d2) * iw1 - (rax * d1 + ray * d2) * iw2;
            note2(d1, d2, tjdi[c][1], tjdi[c][2]);
        }
        {
            const V3<T> ja1 = cross(pd[1], prb), ja2 = cross(pd[2], prb);
            T jdv1 = dot(ja1, dva) - dot(pd[1], dvl), jdv2 = dot(ja2, dva) - dot(pd[2], dvl);
#pragma unroll
            for (int i = 0; i < NP; ++i) { jdv1 += Jt[1][i] * dv[i]; jdv2 += Jt[2][i] * dv[i]; }
            const T limit = mu_tip * plam[0];
            T s1 = plam[1] + (prhs[1] - jdv1) * pjdi[1], s2 = plam[2] + (prhs[2] - jdv2) * pjdi[2];
            friction_clamp(s1, s2, limit, cone);
            if (THR && !live) { s1 = plam[1]; s2 = plam[2]; }
            const T d1 = s1 - plam[1], d2 = s2 - plam[2];
            plam[1] = s1; plam[2] = s2;
#pragma unroll
            for (int i = 0; i < N; ++i) dv[i] += Wa[i][1] * d1 + Wa[i][2] * d2;
            dvl = dvl - (invm * d1) * pd[1] - (invm * d2) * pd[2];
            dva = dva + mul(Iwi, d1 * ja1 + d2 * ja2);
            note2(d1, d2, pjdi[1], pjdi[2]);
        }
    };
    if (rthr > T(0)) {
        bool live = n_it > 0;
        int ran = 0;
        for (int it = 0; it < n_it && __any(live); ++it) {
            bool over = false;
            sweep(std::true_type{}, it, live, over);
            if (live) { ++ran; live = over; }
        }
        if (sweep_acc != nullptr) *sweep_acc += ran;
    } else {
        bool unused = false;
        for (int it = 0; it < n_it; ++it) sweep(std::false_type{}, it, true, unused);
    }
#undef TG_MS
    // ---- integrate
#pragma unroll
    for (int i = 0; i < N; ++i) {
        qd[i] = v[i] + dv[i];
        q[i] += dt * qd[i];
    }
    b.v = vb + dvl;
    b.w = wb + dva;
    xc = xc + dt * b.v;
    integrate_rotation(b.R, b.w, dt);
    b.pos = xc - mul(b.R, load_v3(sc.com));
}

// ------------------------------------------------------------------------------------------------ quaternion / Euler helpers
// (pybullet getQuaternionFromEuler / getEulerFromQuaternion / btMatrix3x3::getRotation; quaternions are x,y,z,w)
template <typename T> struct Q4 { T x, y, z, w; };
template <typename T> __device__ __forceinline__ Q4<T> quat_from_euler(T r, T p, T y) {
    T sr, cr, sp, cp, sy, cy;
    tsincos(T(0.5) * r, &sr, &cr); tsincos(T(0.5) * p, &sp, &cp); tsincos(T(0.5) * y, &sy, &cy);
    Q4<T> q{sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
    const T n = T(1) / tsqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x * n, q.y * n, q.z * n, q.w * n};
}
template <typename T> __device__ __forceinline__ void euler_from_quat(Q4<T> q, T& r, T& p, T& y) {
    const T sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
    const T sarg = T(-2) * (q.x * q.z - q.w * q.y);
    const T half_pi = T(1.5707963267948966);
    if (sarg <= T(-0.99999)) { r = T(0); p = -half_pi; y = T(2) * tatan2(q.x, -q.y); }
    else if (sarg >= T(0.99999)) { r = T(0); p = half_pi; y = T(2) * tatan2(-q.x, q.y); }
    else {
        r = tatan2(T(2) * (q.y * q.z + q.w * q.x), sqw - sqx - sqy + sqz);
        p = tasin(sarg);
        y = tatan2(T(2) * (q.x * q.y + q.w * q.z), sqw + sqx - sqy - sqz);
    }
}
template <typename T> __device__ __forceinline__ M3<T> mat_from_quat(Q4<T> q) {
    const T d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = T(2) / d;
    const T xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const T wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    M3<T> R;
    R.m[0] = T(1) - (yy + zz); R.m[1] = xy - wz; R.m[2] = xz + wy;
    R.m[3] = xy + wz; R.m[4] = T(1) - (xx + zz); R.m[5] = yz - wx;
    R.m[6] = xz - wy; R.m[7] = yz + wx; R.m[8] = T(1) - (xx + yy);
    return R;
}
template <typename T> __device__ __forceinline__ Q4<T> quat_from_mat(const M3<T>& R) {
    const T tr = R.m[0] + R.m[4] + R.m[8];
    Q4<T> q;
    if (tr > T(0)) {
        T s = tsqrt(tr + T(1));
        q.w = T(0.5) * s; s = T(0.5) / s;
        q.x = (R.m[7] - R.m[5]) * s; q.y = (R.m[2] - R.m[6]) * s; q.z = (R.m[3] - R.m[1]) * s;
    } else if (R.m[0] >= R.m[4] && R.m[0] >= R.m[8]) {          // i = 0
        T s = tsqrt(R.m[0] - R.m[4] - R.m[8] + T(1));
        q.x = T(0.5) * s; s = T(0.5) / s;
        q.w = (R.m[7] - R.m[5]) * s; q.y = (R.m[3] + R.m[1]) * s; q.z = (R.m[6] + R.m[2]) * s;
    } else if (R.m[4] > R.m[0] && R.m[4] >= R.m[8]) {           // i = 1
        T s = tsqrt(R.m[4] - R.m[8] - R.m[0] + T(1));
        q.y = T(0.5) * s; s = T(0.5) / s;
        q.w = (R.m[2] - R.m[6]) * s; q.z = (R.m[7] + R.m[5]) * s; q.x = (R.m[1] + R.m[3]) * s;
    } else {                                                    // i = 2
        T s = tsqrt(R.m[8] - R.m[0] - R.m[4] + T(1));
        q.z = T(0.5) * s; s = T(0.5) / s;
        q.w = (R.m[3] - R.m[1]) * s; q.x = (R.m[2] + R.m[6]) * s; q.y = (R.m[5] + R.m[7]) * s;
    }
    return q;
}
template <typename T> __device__ __forceinline__ Q4<T> quat_mul(Q4<T> a, Q4<T> b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

// ------------------------------------------------------------------------------------------------ small dense solves
// Solve A x = b for NxN A by Gaussian elimination with partial pivoting.  Row exchanges are conditional swaps
// (selects), so every index is a compile-time constant and nothing leaves the register file.
template <typename T, int N> __device__ __forceinline__ void solve_pivoted(T (&A)[N][N], T (&b)[N], T (&x)[N]) {
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            const bool sw = tabs(A[r][c]) > tabs(A[c][c]);
#pragma unroll
            for (int j = c; j < N; ++j) { const T t0 = A[c][j], t1 = A[r][j]; A[c][j] = sw ? t1 : t0; A[r][j] = sw ? t0 : t1; }
            const T b0 = b[c], b1 = b[r]; b[c] = sw ? b1 : b0; b[r] = sw ? b0 : b1;
        }
        const T piv = A[c][c];
        const T inv = (piv != T(0)) ? trcp(piv) : T(0);   // seed + Newton (1-2 ulp): the pivots are well-scaled Jacobian / mass-matrix entries
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            const T f = A[r][c] * inv;
#pragma unroll
            for (int j = c + 1; j < N; ++j) A[r][j] -= f * A[c][j];
            b[r] -= f * b[c];
        }
    }
#pragma unroll
    for (int r = N - 1; r >= 0; --r) {
        T s = b[r];
#pragma unroll
        for (int j = r + 1; j < N; ++j) s -= A[r][j] * x[j];
        x[r] = (A[r][r] != T(0)) ? s * trcp(A[r][r]) : T(0);
    }
}

// x = pinv(J) b for a 6 x N Jacobian (np.linalg.pinv(jac) @ v, mg400.py:105-107): one-sided (Hestenes) Jacobi SVD.
// Row pairs of A = J are rotated until mutually orthogonal, A_final = U^T J with rows sigma_i v_i^T, so
//   pinv(J) b = sum_i  row_i(A_final) (u_i . b) / sigma_i^2   over sigma_i > rcond * sigma_max   (numpy: rcond = 1e-15).
// Fixed sweep count, branch-free rotations, compile-time indices only.
template <typename T, int N> __device__ __forceinline__ void pinv_apply(const T (&J)[6][N], const T (&b)[6], T (&x)[N]) {
    T A[6][N], U[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < N; ++c) A[r][c] = J[r][c];
#pragma unroll
        for (int c = 0; c < 6; ++c) U[r][c] = (r == c) ? T(1) : T(0);   // U[r][:] = u_r^T (rows accumulate the same rotations)
    }
    for (int sweep = 0; sweep < 10; ++sweep) {
#pragma unroll
        for (int p = 0; p < 5; ++p)
#pragma unroll
            for (int qq = p + 1; qq < 6; ++qq) {
                T alpha = T(0), beta = T(0), gamma = T(0);
#pragma unroll
                for (int c = 0; c < N; ++c) { alpha += A[p][c] * A[p][c]; beta += A[qq][c] * A[qq][c]; gamma += A[p][c] * A[qq][c]; }
                // rotation that zeroes the inner product (skipped when already orthogonal to rounding)
                const bool rot = tabs(gamma) > T(1e-300) && gamma * gamma > (alpha * beta) * T(1e-32);
                const T zeta = rot ? (beta - alpha) / (T(2) * gamma) : T(0);
                const T tt = rot ? ((zeta >= T(0) ? T(1) : T(-1)) / (tabs(zeta) + tsqrt(T(1) + zeta * zeta))) : T(0);
                const T cs = T(1) / tsqrt(T(1) + tt * tt), sn = cs * tt;
#pragma unroll
                for (int c = 0; c < N; ++c) { const T ap = A[p][c], aq = A[qq][c]; A[p][c] = cs * ap - sn * aq; A[qq][c] = sn * ap + cs * aq; }
#pragma unroll
                for (int c = 0; c < 6; ++c) { const T up = U[p][c], uq = U[qq][c]; U[p][c] = cs * up - sn * uq; U[qq][c] = sn * up + cs * uq; }
            }
    }
    T s2[6], s2max = T(0);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        T acc = T(0);
#pragma unroll
        for (int c = 0; c < N; ++c) acc += A[r][c] * A[r][c];
        s2[r] = acc;
        s2max = acc > s2max ? acc : s2max;
    }
#pragma unroll
    for (int c = 0; c < N; ++c) x[c] = T(0);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        T ub = T(0);
#pragma unroll
        for (int c = 0; c < 6; ++c) ub += U[r][c] * b[c];
        const bool keep = s2[r] > s2max * T(1e-30);     // sigma > 1e-15 sigma_max
        const T w = keep ? ub / s2[r] : T(0);
#pragma unroll
        for (int c = 0; c < N; ++c) x[c] += A[r][c] * w;
    }
}

// Geometric Jacobian of the TCP frame origin, world frame: rows 0-2 translational, 3-5 rotational
// (calculateJacobian(link, localPosition = 0), base_robot_arm.py:300-310).
template <typename T, int TOPO>
__device__ __forceinline__ void tcp_jacobian(const DevRobot<T>& m, const Kin<T, TOPO>& k, V3<T> ptcp, T (&J)[6][Topo<TOPO>::N]) {
    constexpr int N = Topo<TOPO>::N;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        bool on_path = false;
#pragma unroll
        for (int l = 0; l < N; ++l)
            if (l == m.tcp_link && is_ancestor_or_self<TOPO>(i, l)) on_path = true;
        const V3<T> jt = cross(k.a[i], ptcp - k.o[i]);
        J[0][i] = on_path ? jt.x : T(0); J[1][i] = on_path ? jt.y : T(0); J[2][i] = on_path ? jt.z : T(0);
        J[3][i] = on_path ? k.a[i].x : T(0); J[4][i] = on_path ? k.a[i].y : T(0); J[5][i] = on_path ? k.a[i].z : T(0);
    }
}

}  // namespace tg
