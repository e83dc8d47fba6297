// tg_contact_wave.hip - object_push / object_roll env step with ONE WAVEFRONT PER ENV (gfx950).
//
// What it replaces: Robot.apply_action's 24 x stepSimulation (tactile_gym/robots/arms/robot.py:131-141,182-183) for the envs whose tick
// carries rigid contacts (object_push_env.py:196-227, object_roll_env.py:203-248): joint motors + cube-table / cube-tip contacts with
// Coulomb friction, 150 projected Gauss-Seidel sweeps per tick (base_tactile_env.py:127-130).  Same restated contact model as
// tg_physics.hpp:sim_tick_push and oracle/minibullet.c:mb_step_push [PARITY_ASSUMPTIONS A23-A30], same row order.
//
// Why a second mapping.  sim_tick_push gives every env one LANE: 1024 envs are 16 wavefronts on a 256-CU chip, each a serial chain of
// 3600 sweeps x ~800 f64 instructions per env step (7.4 ms, 98 % of the SIMDs idle).  Gauss-Seidel is sequential over rows, but what a
// row update does to the other rows is not: here every solver ROW owns a lane and carries its residual in the Delassus form
//     s_j = lambda_j + ( rhs_j - sum_k A_jk lambda_k - cfm_j lambda_j ) / (A_jj + cfm_j),      A = J Minv_sys J^T,
// i.e. "what lambda_j would become if row j were relaxed now".  Relaxing row i changes lambda_i by d (clamp of s_i minus lambda_i, one
// lane's work), and then every lane updates its own s_j -= G_ji d with ONE fused multiply-add, G_ji = A_ji / (A_jj + cfm_j) precomputed
// per tick (zero diagonal: s_i itself is invariant under its own relaxation).  A row step is therefore
//     clamp (2-3 VALU) -> v_readlane x2 (the delta becomes a scalar) -> v_fma_f64 (all rows at once)
// instead of ~34 dependent instructions, a sweep ~170 instructions instead of ~800, and 1024 envs are 1024 wavefronts: one per SIMD of
// the chip.  Lane-per-env remains the better mapping once the SIMDs are saturated (>= 4096 envs per GPU: it needs 12 instructions per
// env-sweep, this one ~170); tg_step picks by num_envs (tg_config.contact_mapping overrides).  The same mapping resets the envs
// (k_reset_contact_wave: one wavefront per env that finished).
//
// Lane layout (one env per wavefront):   lane i < N ............ joint motor i (btMultiBodyJointMotor row, J = e_i)
//                                        lane 8 + 4c + r ....... contact c (0-3: cube vertex / marble on the table, 4: sensor tip),
//                                                                r = 0 normal, 1 / 2 friction directions (btPlaneSpace1), r = 3 unused
//                                        lanes 32 .. 45 ........ accumulators of the velocity change du (8 joints, cube linear 3, angular 3)
//                                        lanes 48 .. 55 ........ accumulators of the motor impulses (limit watch)
// The arm's dynamics are spread over the lanes as well (tick_dynamics_lanes: a link per lane, the 8 x 8 inertia matrix one entry per
// lane), and so is the search over the hull vertices of the tip core (a wave argmin); contact generation and integration are evaluated
// redundantly by all lanes - the values are wave-uniform, the cost is that of one lane.
//
// Numerics: same mathematics, different summation order (A_ji lambda instead of J_j . dv) - agreement with the oracle is to rounding
// amplified by the contact dynamics, tolerances as for sim_tick_push (joints 1e-9 rad, cube pose 1e-8, images <= 3 px); the contact
// SETS (integer data) are identical.
#include "tg_contact_wave.h"

#include "tg_kernels.hpp"
#include "tg_narrowphase.hpp"

#include <cstdio>

#ifndef TG_WAVE_TIMING
#define TG_WAVE_TIMING 0          // 1: block 0 accumulates s_memtime ticks per tick phase into g_phase (development aid)
#endif

namespace tg {

namespace {

#if TG_WAVE_TIMING
__device__ unsigned long long g_phase[16];
#define TG_STAMP(K) { const unsigned long long now_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) g_phase[K] += now_ - t_prev_; t_prev_ = __builtin_readcyclecounter(); }
#else
#define TG_STAMP(K)
#endif

constexpr int kContactLane0 = 8;     // first contact lane
constexpr int kRowLanes = 28;        // lanes 0..27 carry rows
constexpr int kNG = 8 + 15;          // G entries: motors 0-7, contact (c, r) at 8 + 3c + r
constexpr int kNU = 14;              // generalised velocity components: 8 joints + cube linear 3 + angular 3

__device__ __forceinline__ double bcast(double v, int src_lane) {   // v_readlane_b32 x 2: the value of `src_lane` as a wave-uniform scalar
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane);
    return u.d;
}
// trig_init with the N evaluations on N lanes: every lane of the wavefront holds the same q (one env), so lane i takes joint i's exact sine /
// cosine (lanes 8.. repeat them) and the N pairs come back as wave-uniform values - one library sincos (~150 f64 instructions) on the
// wavefront's issue slot instead of N.  The same routine on the same argument: the same bits as trig_init.
template <typename T, int N>
__device__ __forceinline__ void trig_init_lanes(const T (&q)[N], JointTrig<T, N>& t, int lane) {
    T a = q[0];
#pragma unroll
    for (int i = 1; i < N; ++i) a = (lane & 7) == i ? q[i] : a;
    T s, c;
    tsincos(a, &s, &c);
#pragma unroll
    for (int i = 0; i < N; ++i) { t.s[i] = bcast(s, i); t.c[i] = bcast(c, i); }
}

// the value held by the other friction lane of this lane's contact quad (lanes 4k+1 <-> 4k+2): v_mov_b32_dpp quad_perm:[0,2,1,3] x 2
__device__ __forceinline__ double quad_swap12(double v) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0xD8, 0xF, 0xF, true);
    u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0xD8, 0xF, 0xF, true);
    return u.d;
}
__device__ __forceinline__ float quad_swap12(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xD8, 0xF, 0xF, true)); }
__device__ __forceinline__ float bcast(float v, int src_lane) {
    union { float f; int i; } u;
    u.f = v;
    u.i = __builtin_amdgcn_readlane(u.i, src_lane);
    return u.f;
}
// v_max_f64 / v_min_f64 as single instructions (fmax / fmin get a canonicalising v_max_f64 v, v, v in front of them; nothing here is NaN)
__device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax_neg(double a, double b) { double r; asm("v_max_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // max(a, -b)
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float vmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float vmax_neg(float a, float b) { return __builtin_fmaxf(a, -b); }
__device__ __forceinline__ double vmax_abs(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }   // max(a, |b|)
__device__ __forceinline__ float vmax_abs(float a, float b) { return __builtin_fmaxf(a, __builtin_fabsf(b)); }
__device__ __forceinline__ bool uniform_true(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0; }   // scalar branch on a wave-uniform flag (v_cmp -> s_cmp)

// One stepSimulation() tick; all arguments wave-uniform, `lane` = threadIdx.x.  scr: 23 * 16 words of LDS private to the wavefront.
// ---- LDS layout of one env (words of T).  The env's state lives here for the whole step: every phase of a tick loads what it needs
// and writes its results back (lane 0), so that nothing but a handful of scalars stays in registers from one phase to the next - the
// register file belongs to the phase that runs (the dynamics need ~400 VGPRs, the sweeps ~150).
constexpr int kLQ = 0, kLQd = 8, kLTrigS = 16, kLTrigC = 24, kLQDes = 32, kLQdDes = 40;      // arm: q, qd, sin q, cos q, motor targets
constexpr int kLBody = 48;                                                                    // free body: pos 3, R 9, v 3, w 3
constexpr int kLMinv = 66, kLV = 130;                                                         // per tick: Minv 8 x 8, unconstrained arm velocity
constexpr int kLTipF = 138;                                                                   // tip link frame: origin 3, rotation 9
constexpr int kLJa = 150, kLJo = 168;                                                         // joint axes / origins of the tip's ancestors, <= 6 x 3 each
constexpr int kLFree = 186;                                                                   // vb 3, wb 3, xc 3 of the tick
constexpr int kLW = 196;                                                                      // W rows, 23 x 16
constexpr int kCW = 120;                                                                     // per-link constants (below), padded
constexpr int kLC = kLW + kNG * 16;                                                           // link constant table, 8 x kCW
constexpr int kLS = kLC + 8 * kCW;                                                            // link slots for the inertia matrix: a, o, F, N (12) x 8
constexpr int kLHull = kLS + 8 * 12;                                                          // tip-core hull vertices, 3 per vertex

// tg_config.narrowphase != 0 (NT = 4 tip slots): a region behind the hull (its offset is a run-time value, kLHull + 3 n_tip):
//   [0, 912)      the expanding polytope's scratch during contact generation, then the solver's W rows (32 x 16) and the tip rows' J (12 x 16)
//   [912, 977)    the tip - cube manifold (tg_narrowphase.hpp layout)
constexpr int kXW = 0, kXJ = 32 * 16, kXMani = narrow::kScratchWords > (32 + 12) * 16 ? narrow::kScratchWords : (32 + 12) * 16, kXWords = kXMani + narrow::kManiWords + 7;

#define TG_PHASE_FENCE() { __syncthreads(); asm volatile("" ::: "memory"); }

// ---- lane-parallel arm dynamics ---------------------------------------------------------------------------------------------------
// value of `v` held by lane `src` (per-lane source): ds_bpermute_b32 x 2
__device__ __forceinline__ double lane_fetch(double v, int src) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[0]);
    u.i[1] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[1]);
    return u.d;
}
__device__ __forceinline__ float lane_fetch(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v))); }

// v_mov_b32_dpp x 2 with a compile-time control word (row_shr:n = 0x110 + n within rows of 16 lanes, quad_perm, row_half_mirror = 0x141):
// the lane-to-lane moves of the dynamics that a ds_bpermute (~55 cycles each, measured) would otherwise make
template <int CTRL> __device__ __forceinline__ double dpp_move(double v) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], CTRL, 0xF, 0xF, true);
    u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], CTRL, 0xF, 0xF, true);
    return u.d;
}
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
// the parent link's copy of v: every parent is the lane below (DFS numbering) except MG400's link 5, whose parent is link 0
template <int TOPO, typename T> __device__ __forceinline__ T parent_fetch(T v, bool far_parent) {
    const T near = dpp_move<0x111>(v);
    if constexpr (TOPO == 1) { const T far = dpp_move<0x115>(v); return far_parent ? far : near; }
    return near;
}

// Per-link constants in LDS (row = link), copied from the DevRobot once per step: what link i contributes to the arm's kinematics, inertia
// and velocity damping.  Offsets within a row:
constexpr int kCfkA = 0, kCfkB = 9, kCfkC = 18, kCjpos = 27, kCaxis = 30, kClcom = 33, kClinert = 36, kClmass = 42, kClang = 43, kCbmass = 49, kCbcom = 53;   // .. 65
template <typename T, int TOPO>
__device__ __forceinline__ void stage_link_constants(const DevRobot<T>* __restrict__ mp, lds_ptr<T> L, int lane) {
    constexpr int N = Topo<TOPO>::N;
    const int li = lane & 7;
    if (lane < 8 && li < N) {
        const DevRobot<T>& m = *mp;
        const lds_ptr<T> C = L + kLC + li * kCW;
        for (int e = 0; e < 9; ++e) { C[kCfkA + e] = m.fkA[li][e]; C[kCfkB + e] = m.fkB[li][e]; C[kCfkC + e] = m.fkC[li][e]; }
        for (int e = 0; e < 3; ++e) { C[kCjpos + e] = m.jpos[li][e]; C[kCaxis + e] = m.jaxis[li][e]; C[kClcom + e] = m.lcom[li][e]; }
        for (int e = 0; e < 6; ++e) { C[kClinert + e] = m.linert[li][e]; C[kClang + e] = m.lang[li][e]; }
        C[kClmass] = m.lmass[li];
        for (int b = 0; b < kMaxBodiesPerLink; ++b) {
            C[kCbmass + b] = m.bmass[li][b] > T(0) ? m.bmass[li][b] : T(0);
            for (int e = 0; e < 3; ++e) C[kCbcom + 3 * b + e] = m.bcom[li][b][e];
        }
    }
}

// Phase 1 of a tick, spread over the lanes (the wave-uniform version below evaluates one scalar program on 64 lanes and streams ~400
// robot constants through scalar loads and spilled SGPRs every tick: 44 k of a tick's 200 k cycles).  Lane l works on link l & 7 (eight
// replicas, so that lane (r, c) = (l >> 3, l & 7) of the 8 x 8 inertia matrix already holds link c):
//   kinematics     each lane takes its parent's frame and velocity from the parent's lane (ds_bpermute) and composes its own; after d rounds
//                  every link of depth <= d is final (tree depth: UR5 5, MG400 4);
//   composites     link i's composite inertia / first moment / damping wrench about its joint origin = sum over its descendants j of link j's
//                  own terms shifted by o_j - o_i (the same parallel-axis terms the leaf-to-root recursion applies, summed directly):
//                  N broadcast rounds, no dependency between them;
//   M              lane (r, c): a_i . (N_j + (o_j - o_i) x F_j), i = min, j = max (0 unless i is an ancestor of j), link r's terms from LDS;
//   Minv           Gauss-Jordan in place over the 64 lanes (SPD: no pivoting), row / column of the pivot through ds_bpermute;
//   v              qd + dt Minv (qdamp - joint_damp qd): lane products, xor-reduction over the 8 lanes of a row.
// Same outputs in LDS as the wave-uniform version (Minv, v, tip link frame, joint axes / origins on the tip's path).
template <typename T, int TOPO>
__device__ __forceinline__ void tick_dynamics_lanes(const DevRobot<T>* __restrict__ mp, lds_ptr<T> L, int tip_link, T dt, int lane
#if TG_WAVE_TIMING
                                                    , unsigned long long& t_prev_
#endif
                                                    ) {
    constexpr int N = Topo<TOPO>::N;
    constexpr int NP = Topo<TOPO>::NP;
    const int lc = lane & 7, lr = lane >> 3;
    const int li = lc < N ? lc : N - 1;                      // this lane's link (columns >= N of the UR5 repeat the last link, unused)
    int par = 0, depth = 0;
    unsigned desc = 0;                                        // bit j: link j is li or one of its descendants
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int d = 0;
        for (int a = i; Topo<TOPO>::parent(a) >= 0; a = Topo<TOPO>::parent(a)) ++d;
        unsigned ds = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) ds |= is_ancestor_or_self<TOPO>(i, j) ? (1u << j) : 0u;
        if (li == i) { par = Topo<TOPO>::parent(i) < 0 ? i : Topo<TOPO>::parent(i); depth = d; desc = ds; }
    }
    constexpr int max_depth = TOPO == 0 ? 5 : 4;
    const bool is_root = depth == 0;
    const lds_ptr<T> C = L + kLC + li * kCW;
    const T qdi = L[kLQd + li], sq = L[kLTrigS + li], cq = L[kLTrigC + li];
    M3<T> Rl;
#pragma unroll
    for (int e = 0; e < 9; ++e) Rl.m[e] = C[kCfkA + e] + cq * C[kCfkB + e] + sq * C[kCfkC + e];
    const V3<T> jpos = mk(C[kCjpos], C[kCjpos + 1], C[kCjpos + 2]), axis = mk(C[kCaxis], C[kCaxis + 1], C[kCaxis + 2]);
    // ---- kinematics and velocities, root -> leaf
    M3<T> R = Rl;
    V3<T> o = jpos, a = mul(R, axis), w = qdi * a, vo = mk<T>(0, 0, 0);
    const bool far_parent = TOPO == 1 && li == 5;             // (Topo<1>::parent(5) == 0; every other parent is li - 1)
    static_assert(TOPO == 0 || (Topo<TOPO>::parent(5) == 0 && Topo<TOPO>::parent(6) == 5 && Topo<TOPO>::parent(4) == 3), "parent_fetch: MG400 numbering");
#pragma unroll
    for (int d = 0; d < max_depth; ++d) {
        M3<T> Rp;
#pragma unroll
        for (int e = 0; e < 9; ++e) Rp.m[e] = parent_fetch<TOPO>(R.m[e], far_parent);
        const V3<T> op = mk(parent_fetch<TOPO>(o.x, far_parent), parent_fetch<TOPO>(o.y, far_parent), parent_fetch<TOPO>(o.z, far_parent));
        const V3<T> wp = mk(parent_fetch<TOPO>(w.x, far_parent), parent_fetch<TOPO>(w.y, far_parent), parent_fetch<TOPO>(w.z, far_parent));
        const V3<T> vp = mk(parent_fetch<TOPO>(vo.x, far_parent), parent_fetch<TOPO>(vo.y, far_parent), parent_fetch<TOPO>(vo.z, far_parent));
        const M3<T> Rn = mul(Rp, Rl);
        const V3<T> on = op + mul(Rp, jpos);
        const V3<T> an = mul(Rn, axis);
        const V3<T> wn = wp + qdi * an;
        const V3<T> vn = vp + cross(wp, on - op);
        if (!is_root) { R = Rn; o = on; a = an; w = wn; vo = vn; }
    }
    TG_STAMP(7)
    // ---- this link's own inertia terms about its joint origin, damping wrench (per-body linear part, merged angular part)
    const T lmass = C[kClmass];
    const V3<T> rc = mul(R, mk(C[kClcom], C[kClcom + 1], C[kClcom + 2]));
    const S3<T> Il{C[kClinert], C[kClinert + 1], C[kClinert + 2], C[kClinert + 3], C[kClinert + 4], C[kClinert + 5]};
    const S3<T> Io_own = rotate(R, Il) + point_inertia(lmass, rc);
    const V3<T> hc_own = lmass * rc;
    const DevRobot<T>& m = *mp;
    const T ang_damp = m.ang_damp, lin_damp = m.lin_damp, joint_damp = m.joint_damp;
    const T sw = ang_damp + ang_damp * tsqrt_fast(dot(w, w));
    const S3<T> Ia{C[kClang], C[kClang + 1], C[kClang + 2], C[kClang + 3], C[kClang + 4], C[kClang + 5]};
    V3<T> dF = mk<T>(0, 0, 0);
    V3<T> dN = (-sw) * mul(R, mul(Ia, mulT(R, w)));
#pragma unroll
    for (int b = 0; b < kMaxBodiesPerLink; ++b) {             // a slot without a body has mass 0: its terms vanish
        const T bm = C[kCbmass + b];
        const V3<T> rb = mul(R, mk(C[kCbcom + 3 * b], C[kCbcom + 3 * b + 1], C[kCbcom + 3 * b + 2]));
        const V3<T> vb = vo + cross(w, rb);
        const T sv = lin_damp + lin_damp * tsqrt_fast(dot(vb, vb));
        const V3<T> Fb = (-bm * sv) * vb;
        dF = dF + Fb;
        dN = dN + cross(rb, Fb);
    }
    TG_STAMP(8)
    // ---- composites: sum over the descendants (link j's own terms broadcast from lane j of replica 0, shifted to this link's origin)
    S3<T> Io{T(0), T(0), T(0), T(0), T(0), T(0)};
    V3<T> hc = mk<T>(0, 0, 0), DN = mk<T>(0, 0, 0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const T keep = ((desc >> j) & 1u) ? T(1) : T(0);
        const V3<T> oj = mk(bcast(o.x, j), bcast(o.y, j), bcast(o.z, j));
        const V3<T> r = oj - o;
        const T mj = bcast(lmass, j);
        const V3<T> hj = mk(bcast(hc_own.x, j), bcast(hc_own.y, j), bcast(hc_own.z, j));
        const S3<T> Ij{bcast(Io_own.xx, j), bcast(Io_own.xy, j), bcast(Io_own.xz, j), bcast(Io_own.yy, j), bcast(Io_own.yz, j), bcast(Io_own.zz, j)};
        const V3<T> Fj = mk(bcast(dF.x, j), bcast(dF.y, j), bcast(dF.z, j)), Nj = mk(bcast(dN.x, j), bcast(dN.y, j), bcast(dN.z, j));
        const S3<T> It = Ij + point_inertia(mj, r) + cross_inertia(hj, r);
        Io.xx = __builtin_fma(keep, It.xx, Io.xx); Io.xy = __builtin_fma(keep, It.xy, Io.xy); Io.xz = __builtin_fma(keep, It.xz, Io.xz);
        Io.yy = __builtin_fma(keep, It.yy, Io.yy); Io.yz = __builtin_fma(keep, It.yz, Io.yz); Io.zz = __builtin_fma(keep, It.zz, Io.zz);
        const V3<T> ht = hj + mj * r, Nt = Nj + cross(r, Fj);
        hc = hc + keep * ht;
        DN = DN + keep * Nt;
    }
    const T qdamp = dot(a, DN);
    TG_STAMP(9)
    // ---- joint-space inertia: lane (lr, lc)
    const V3<T> F = cross(a, hc), Nv = mul(Io, a);
    if (lane < 8) {                                           // replica 0 publishes its links
        const lds_ptr<T> S = L + kLS + lc * 12;
        S[0] = a.x; S[1] = a.y; S[2] = a.z; S[3] = o.x; S[4] = o.y; S[5] = o.z;
        S[6] = F.x; S[7] = F.y; S[8] = F.z; S[9] = Nv.x; S[10] = Nv.y; S[11] = Nv.z;
    }
    __syncthreads();
    T A;
    {
        const int rr = lr < N ? lr : N - 1;
        const lds_ptr<T> S = L + kLS + rr * 12;
        const V3<T> ar = mk(S[0], S[1], S[2]), orr = mk(S[3], S[4], S[5]), Fr = mk(S[6], S[7], S[8]), Nr = mk(S[9], S[10], S[11]);
        const bool c_is_j = lc >= lr;                         // j = max(r, c), i = min(r, c)
        const V3<T> ai = c_is_j ? ar : a, oi = c_is_j ? orr : o;
        const V3<T> oj = c_is_j ? o : orr, Fj = c_is_j ? F : Fr, Nj = c_is_j ? Nv : Nr;
        const int i_ = c_is_j ? lr : lc, j_ = c_is_j ? lc : lr;
        bool anc = false;
#pragma unroll
        for (int ii = 0; ii < N; ++ii)
#pragma unroll
            for (int jj = ii; jj < N; ++jj)
                if (is_ancestor_or_self<TOPO>(ii, jj)) anc = anc || (i_ == ii && j_ == jj);
        const T val = dot(ai, Nj + cross(oj - oi, Fj));
        A = (lr < N && lc < N) ? (anc ? val : T(0)) : (lr == lc ? T(1) : T(0));      // identity padding beyond N
    }
    TG_STAMP(10)
    // ---- in-place Gauss-Jordan inverse of the symmetric positive definite M
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const T pk = bcast(A, 9 * k);
        const T ark = lane_fetch(A, (lane & ~7) + k), akc = lane_fetch(A, 8 * k + lc);
        const T inv = T(1) / pk;
        const T t = akc * inv;
        const T An = __builtin_fma(-ark, t, A);
        A = lr == k ? (lc == k ? inv : t) : (lc == k ? -ark * inv : An);
    }
    TG_STAMP(11)
    // ---- outputs
    L[kLMinv + 8 * lr + lc] = A;
    {
        T prod = A * (qdamp - joint_damp * qdi);              // column lc: this lane's own link
        if (lc >= N) prod = T(0);
        prod += dpp_move<0xB1>(prod);                         // quad_perm [1,0,3,2]
        prod += dpp_move<0x4E>(prod);                         // quad_perm [2,3,0,1]
        prod += dpp_move<0x141>(prod);                        // row_half_mirror: lane i <-> 7 - i of each 8-lane row group
        if (lc == 0 && lr < N) L[kLV + lr] = L[kLQd + lr] + dt * prod;
    }
    if (lane < 8) {
        if (lc == tip_link) {
            L[kLTipF + 0] = o.x; L[kLTipF + 1] = o.y; L[kLTipF + 2] = o.z;
#pragma unroll
            for (int e = 0; e < 9; ++e) L[kLTipF + 3 + e] = R.m[e];
        }
        if (lc < NP) {
            bool on_path = false;                             // joints off the tip's path get a zero axis: their Jacobian column vanishes
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int l = 0; l < N; ++l)
                    if (is_ancestor_or_self<TOPO>(i, l)) on_path = on_path || (lc == i && l == tip_link);
            const T keep = on_path ? T(1) : T(0);
            L[kLJa + 3 * lc] = keep * a.x; L[kLJa + 3 * lc + 1] = keep * a.y; L[kLJa + 3 * lc + 2] = keep * a.z;
            L[kLJo + 3 * lc] = o.x; L[kLJo + 3 * lc + 1] = o.y; L[kLJo + 3 * lc + 2] = o.z;
        }
    }
}

// Phase 1 of a tick: articulated-body dynamics of the arm (wave-uniform: every lane evaluates the same thing, lane 0 writes).  In: q, qd,
// carried sines / cosines (LDS).  Out (LDS): Minv, the unconstrained arm velocity v, the frame of the link that carries the tip and the
// axes / origins of the joints on its path.  (Measured as an out-of-line call: 3x slower - with a call in the kernel the sweep loop's
// coefficient rows are spilled to scratch and reloaded inside the sweeps.)
template <typename T, int TOPO>
__device__ __forceinline__ void tick_dynamics(const DevRobot<T>* __restrict__ mp, lds_ptr<T> L, int tip_link, T dt, int lane) {
    constexpr int N = Topo<TOPO>::N;
    constexpr int NP = Topo<TOPO>::NP;
    // (The ~250 robot constants become loop invariants of the tick loop, overflow the SGPR file and come back through v_readlane from
    // spill registers; measured alternatives were slower: re-issuing the s_loads every tick 42 -> 99 k cycles per tick, the constants
    // in LDS 42 -> 77 k (VGPR spills), the phase as an out-of-line call 3x on the whole step.)
    const DevRobot<T>& m = *mp;
    const V3<T> gravity = load_v3(m.gravity);
    {
        T q[N], qd[N], Minv[N][N], v[N];
        JointTrig<T, N> trig;
#pragma unroll
        for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; trig.s[i] = L[kLTrigS + i]; trig.c[i] = L[kLTrigC + i]; }
        Kin<T, TOPO> kin;
        {
            T hb[N], qdm[N], traceM;
#ifdef TG_WAVE_NOTRIG
            dynamics_terms<T, TOPO, false, false>(m, q, qd, hb, qdm, Minv, traceM, gravity, kin);
#else
            dynamics_terms<T, TOPO, false, true>(m, q, qd, hb, qdm, Minv, traceM, gravity, kin, &trig);   // gravity compensation cancels the bias force
#endif
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
        V3<T> ol; M3<T> Rl;
        {
            const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
            const T z3[3] = {T(0), T(0), T(0)};
            link_frame<T, TOPO>(kin, tip_link, z3, ident, ol, Rl);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                L[kLV + i] = v[i];
#pragma unroll
                for (int j = 0; j < N; ++j) L[kLMinv + 8 * i + j] = Minv[i][j];
            }
            L[kLTipF + 0] = ol.x; L[kLTipF + 1] = ol.y; L[kLTipF + 2] = ol.z;
#pragma unroll
            for (int e = 0; e < 9; ++e) L[kLTipF + 3 + e] = Rl.m[e];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                bool on_path = false;
#pragma unroll
                for (int l = 0; l < N; ++l)
                    if (l == tip_link && is_ancestor_or_self<TOPO>(i, l)) on_path = true;
                const T keep = on_path ? T(1) : T(0);           // joints off the tip's path get a zero axis: their Jacobian column vanishes
                L[kLJa + 3 * i] = keep * kin.a[i].x; L[kLJa + 3 * i + 1] = keep * kin.a[i].y; L[kLJa + 3 * i + 2] = keep * kin.a[i].z;
                L[kLJo + 3 * i] = kin.o[i].x; L[kLJo + 3 * i + 1] = kin.o[i].y; L[kLJo + 3 * i + 2] = kin.o[i].z;
            }
        }
    }
}

// One stepSimulation() tick on the LDS-resident env state.  Returns the tick's contact code (tg_state_view.contact_ids).
template <typename T, int TOPO, int MOTOR, int SHAPE, bool CONE, int NT = 1>
__device__ __forceinline__ int sim_tick_contact_wave(const DevRobot<T>& m, const PushScene<T>& sc, lds_ptr<T> L, T kp, T kd, T max_force, T dt, int iters,
                                                     T mass, int lane_in, int xbase = 0, int* sweep_acc = nullptr /* threshold mode: += the sweeps this tick ran */) {
    constexpr int N = Topo<TOPO>::N;
    constexpr int NP = Topo<TOPO>::NP;
    // NT = 1: one tip contact per tick from the closed forms (the default).  NT = 4 (object_push, f64): up to four tip contacts from the
    // persistent manifold fed by GJK / EPA; the row lanes grow to 40, the accumulator lanes move to 40.., the watch lanes to 56..
    static_assert(NT == 1 || (NT == 4 && SHAPE == 0), "the manifold variant is object_push's");
    constexpr int RL = NT == 1 ? kRowLanes : 40, ACC0 = NT == 1 ? 32 : 40, W0 = NT == 1 ? 48 : 56, NGT = NT == 1 ? kNG : 32, NCT = 4 + NT;
    const int lw = NT == 1 ? kLW : xbase + kXW;          // where the W rows of this tick live
    int contact_code = 0;
    // The lane index goes through an opaque copy per tick: everything derived from it alone (row masks, the H table, accumulator selects:
    // ~60 values) would otherwise be hoisted out of the 24-tick loop as loop invariants and, with the register file full, live in scratch
    // memory and come back through scratch loads every tick (measured: 0.55 GB of HBM traffic per step on the MG400 build).
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
#if TG_WAVE_TIMING
    unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
    const V3<T> gravity = load_v3(m.gravity);
    // =============================================================== phase 1: articulated-body dynamics (wave-uniform)
#ifdef TG_WAVE_SCALAR_DYNAMICS
    tick_dynamics<T, TOPO>(&m, L, sc.tip_link, dt, lane);
#else
#if TG_WAVE_TIMING
    tick_dynamics_lanes<T, TOPO>(&m, L, sc.tip_link, dt, lane, t_prev_);
#else
    tick_dynamics_lanes<T, TOPO>(&m, L, sc.tip_link, dt, lane);
#endif
#endif
    TG_PHASE_FENCE()
    TG_STAMP(0)
    // =============================================================== phase 2: free body, contact generation
    FreeBody<T> b;
    b.pos = mk(L[kLBody + 0], L[kLBody + 1], L[kLBody + 2]);
#pragma unroll
    for (int e = 0; e < 9; ++e) b.R.m[e] = L[kLBody + 3 + e];
    b.v = mk(L[kLBody + 12], L[kLBody + 13], L[kLBody + 14]);
    b.w = mk(L[kLBody + 15], L[kLBody + 16], L[kLBody + 17]);
    const T radius = mass;   // SHAPE 1: `mass` carries the episode's radius
    const T iscale = SHAPE == 1 ? (radius / sc.radius0) * (radius / sc.radius0) : mass / sc.mass0, invm = T(1) / (SHAPE == 1 ? sc.mass0 : mass);
    const S3<T> I0{sc.inertia0[0] * iscale, sc.inertia0[1] * iscale, sc.inertia0[2] * iscale, sc.inertia0[3] * iscale, sc.inertia0[4] * iscale,
                   sc.inertia0[5] * iscale};
    const S3<T> Iw = rotate(b.R, I0), Iwi = inverse(Iw);
    const V3<T> xc = b.pos + mul(b.R, load_v3(sc.com));
    V3<T> vb, wb;
    {
        const T sv = sc.lin_damp + sc.lin_damp * norm(b.v), sw = sc.ang_damp + sc.ang_damp * norm(b.w);
        const V3<T> Iwv = mul(Iw, b.w);
        const V3<T> Nt = (-sw) * Iwv - cross(b.w, Iwv);
        vb = b.v + dt * (gravity - sv * b.v);
        wb = b.w + dt * mul(Iwi, Nt);
    }
    if (lane == 0) {
        L[kLFree + 0] = vb.x; L[kLFree + 1] = vb.y; L[kLFree + 2] = vb.z;
        L[kLFree + 3] = wb.x; L[kLFree + 4] = wb.y; L[kLFree + 5] = wb.z;
        L[kLFree + 6] = xc.x; L[kLFree + 7] = xc.y; L[kLFree + 8] = xc.z;
    }
    // ---- this lane's row
    const int cl = lane - kContactLane0;
    const int cc = cl >> 2, rr_ = cl & 3;                                   // contact slot and row within it
    const bool contact_lane = lane >= kContactLane0 && lane < RL && rr_ < 3;
    const bool tip_lane = contact_lane && cc >= 4, table_lane = contact_lane && cc < 4, motor_lane = lane < N;
    const int ts_ = cc >= 4 ? (cc - 4 < NT ? cc - 4 : NT - 1) : 0;       // this lane's tip slot (NT = 4)
    // ---- body - table contacts: the kept cube vertices take slots 0.. in vertex order; a table lane finds the vertex of its slot
    V3<T> my_ra = mk<T>(0, 0, 0);
    T my_depth = T(0);
    int n_table = 0;
    if constexpr (SHAPE == 1) {                                             // marble: its lowest point, one slot
        const T vz0 = (b.pos.z - radius) - sc.table_z;
        n_table = (vz0 <= sc.breaking) ? 1 : 0;
        contact_code = n_table;
        my_ra = mk(T(0), T(0), -radius);
        my_depth = vz0;
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
        n_table = cnt;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool mine = ((keep >> c) & 1) && __builtin_popcount(keep & ((1 << c) - 1)) == cc;
            if (mine) {
                const T lx = (c & 4) ? sc.half[0] : -sc.half[0], ly = (c & 2) ? sc.half[1] : -sc.half[1], lz = (c & 1) ? sc.half[2] : -sc.half[2];
                my_ra = (b.pos + mul(b.R, mk(lx, ly, lz))) - xc;
                my_depth = vz[c];
            }
        }
    }
    TG_STAMP(1)
    // ---- body - tip contact (wave-uniform result)
    V3<T> jt[NP];                       // translational Jacobian columns of the tip point
    V3<T> tdir[3], trb;                 // row directions (n, t1, t2), contact arm on the body
    T tip_depth; bool tip_active;
    int tip_mask = 0;                   // NT = 4: the manifold's live slots
    if constexpr (NT == 4) {
        // ---- the general narrowphase: broadphase AABB overlap of the pair, GJK / EPA on the core shapes in the box frame, one new point per
        // tick into the persistent manifold, refresh; every surviving point is a contact (oracle/minibullet.c, narrowphase == 1) [A35-A38]
        if constexpr (sizeof(T) == 8) {
            const narrow::lptr<double> xs = (narrow::lptr<double>)(L + xbase), mf = (narrow::lptr<double>)(L + xbase + kXMani);
            double ol[3], Rl[9], bp[3], bR[9];
#pragma unroll
            for (int e = 0; e < 3; ++e) { ol[e] = L[kLTipF + e]; bp[e] = L[kLBody + e]; }
#pragma unroll
            for (int e = 0; e < 9; ++e) { Rl[e] = L[kLTipF + 3 + e]; bR[e] = L[kLBody + 3 + e]; }
            const int n_tip = __builtin_amdgcn_readfirstlane(sc.n_tip);
            narrow::Hull H;
            H.n = n_tip;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int k = 0; k < narrow::kSlots; ++k) {
                    const int i = 64 * k + lane;
                    const int ii = i < n_tip ? i : 0;
                    const double v0 = L[kLHull + 3 * ii], v1 = L[kLHull + 3 * ii + 1], v2 = L[kLHull + 3 * ii + 2];
                    double w[3], dd[3];
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        w[x] = ol[x] + ((Rl[3 * x] * v0 + Rl[3 * x + 1] * v1) + Rl[3 * x + 2] * v2);
                        dd[x] = w[x] - bp[x];
                        if (i < n_tip) { lo[x] = w[x] < lo[x] ? w[x] : lo[x]; hi[x] = w[x] > hi[x] ? w[x] : hi[x]; }
                    }
                    H.x[k] = (bR[0] * dd[0] + bR[3] * dd[1]) + bR[6] * dd[2];
                    H.y[k] = (bR[1] * dd[0] + bR[4] * dd[1]) + bR[7] * dd[2];
                    H.z[k] = (bR[2] * dd[0] + bR[5] * dd[1]) + bR[8] * dd[2];
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
                for (int x = 0; x < 3; ++x) { lo[x] = vmin(lo[x], __shfl_xor(lo[x], off)); hi[x] = vmax(hi[x], __shfl_xor(hi[x], off)); }
            const double half[3] = {sc.half[0], sc.half[1], sc.half[2]};
            bool overlap = n_tip > 0;
            {
#pragma clang fp contract(off)
                const double pad = (sc.margin_tip + sc.margin_cube) + sc.breaking;
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const double ext = (fabs(bR[3 * x]) * half[0] + fabs(bR[3 * x + 1]) * half[1]) + fabs(bR[3 * x + 2]) * half[2];
                    if (lo[x] - pad > bp[x] + ext || hi[x] + pad < bp[x] - ext) overlap = false;
                }
            }
            __syncthreads();
            if (!uniform_true(overlap) || sc.narrow == 2) { if (lane == 0) mf[narrow::kMcount] = 0.0; }
            __syncthreads();
            if (uniform_true(overlap)) {
                double sd = 0.0, nb[3], ab[3], bb[3];
                if (narrow::gjk_epa_hull_box(H, half, xs + kXW, sd, nb, ab, bb, lane)) {
#pragma clang fp contract(off)
                    const double depth = sd - (sc.margin_tip + sc.margin_cube);
                    double nw[3], aw[3], bw[3], pa_[3], pb_[3];
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        nw[x] = (bR[3 * x] * nb[0] + bR[3 * x + 1] * nb[1]) + bR[3 * x + 2] * nb[2];
                        aw[x] = (bR[3 * x] * ab[0] + bR[3 * x + 1] * ab[1]) + bR[3 * x + 2] * ab[2];
                        bw[x] = (bR[3 * x] * bb[0] + bR[3 * x + 1] * bb[1]) + bR[3 * x + 2] * bb[2];
                    }
#pragma unroll
                    for (int x = 0; x < 3; ++x) { pa_[x] = (bp[x] + aw[x]) - nw[x] * sc.margin_tip; pb_[x] = (bp[x] + bw[x]) + nw[x] * sc.margin_cube; }
                    narrow::manifold_add(mf, sc.breaking, ol, Rl, bp, bR, pa_, pb_, nw, depth, lane);
                }
                narrow::manifold_refresh(mf, sc.breaking, ol, Rl, bp, bR, lane);
            }
            __syncthreads();
            const int mc = __builtin_amdgcn_readfirstlane((int)mf[narrow::kMcount]);
            tip_mask = (1 << mc) - 1;
            contact_code |= (tip_mask << 8) | (1 << 30);
            // this lane's tip slot: normal, arm points, depth (lanes that are not tip lanes read slot 0: finite values nobody uses)
            const V3<T> nrm = mk((T)mf[narrow::kMn + 3 * ts_], (T)mf[narrow::kMn + 3 * ts_ + 1], (T)mf[narrow::kMn + 3 * ts_ + 2]);
            const V3<T> pa = mk((T)mf[narrow::kMpa + 3 * ts_], (T)mf[narrow::kMpa + 3 * ts_ + 1], (T)mf[narrow::kMpa + 3 * ts_ + 2]);
            const V3<T> pb = mk((T)mf[narrow::kMpb + 3 * ts_], (T)mf[narrow::kMpb + 3 * ts_ + 1], (T)mf[narrow::kMpb + 3 * ts_ + 2]);
            const bool live = ((tip_mask >> ts_) & 1) != 0;
            tip_depth = live ? (T)mf[narrow::kMdepth + ts_] : T(0);
            tip_active = live;
            V3<T> t1, t2;
            plane_space(live ? nrm : mk<T>(0, 0, 1), t1, t2);
            tdir[0] = live ? nrm : mk<T>(0, 0, 1); tdir[1] = t1; tdir[2] = t2;
            trb = pb - xc;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const V3<T> ai = mk(L[kLJa + 3 * i], L[kLJa + 3 * i + 1], L[kLJa + 3 * i + 2]);
                const V3<T> oi = mk(L[kLJo + 3 * i], L[kLJo + 3 * i + 1], L[kLJo + 3 * i + 2]);
                jt[i] = cross(ai, pa - oi);
            }
            __syncthreads();   // the scratch region becomes the W / J rows below
        }
    } else {
        const V3<T> ol = mk(L[kLTipF + 0], L[kLTipF + 1], L[kLTipF + 2]);
        M3<T> Rl;
#pragma unroll
        for (int e = 0; e < 9; ++e) Rl.m[e] = L[kLTipF + 3 + e];
        V3<T> nrm, pa, pb;
        if constexpr (SHAPE == 1) {     // marble - tip: closest point of the solid cylinder to the sphere centre
            const V3<T> cw = ol + mul(Rl, load_v3(sc.cyl_pos));
            const M3<T> Rw = mul(Rl, sc.cyl_rot);
            const V3<T> p = mulT(Rw, b.pos - cw);
            const T rad = tsqrt(p.x * p.x + p.y * p.y);
            const T sr = rad > sc.cyl_r ? sc.cyl_r / rad : T(1);
            const V3<T> clp = mk(p.x * sr, p.y * sr, p.z > sc.cyl_hl ? sc.cyl_hl : (p.z < -sc.cyl_hl ? -sc.cyl_hl : p.z));
            const V3<T> g = p - clp;
            const T dist = tsqrt(dot(g, g));
            tip_depth = dist - radius;
            tip_active = dist > T(0) && tip_depth <= sc.breaking;
            contact_code |= tip_active ? (1 << 8) : 0;
            const T idist = T(1) / (dist > T(0) ? dist : T(1));
            V3<T> gw = mul(Rw, idist * g);
            if (!(dist > T(0))) gw = mk<T>(0, 0, 1);   // centre inside the cylinder: no contact, but the disabled rows still need a finite frame
            nrm = mk<T>(0, 0, 0) - gw;
            pa = cw + mul(Rw, clp);
            pb = b.pos - radius * gw;
        } else {                        // cube - tip core: hull vertex with the smallest signed distance to the box; lanes share the search
            M3<T> Mr;   // cube <- link rotation  Rc^T Rl
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) Mr.m[3 * i + j] = b.R.m[i] * Rl.m[j] + b.R.m[3 + i] * Rl.m[3 + j] + b.R.m[6 + i] * Rl.m[6 + j];
            const V3<T> tr = mulT(b.R, ol - b.pos);
            T best_key = T(1e30); int best_i = 0x7fffffff;
            const int n_tip = __builtin_amdgcn_readfirstlane(sc.n_tip);
            for (int i0 = 0; i0 < n_tip; i0 += 64) {       // lane l takes vertices l, l + 64, ... (LDS, stride-3 words: conflict free)
                const int i = i0 + lane;
                const int ii = i < n_tip ? i : 0;
                const V3<T> vv = mk(L[kLHull + 3 * ii], L[kLHull + 3 * ii + 1], L[kLHull + 3 * ii + 2]);
                const V3<T> p = tr + mul(Mr, vv);
                const T qx = tabs(p.x) - sc.half[0], qy = tabs(p.y) - sc.half[1], qz = tabs(p.z) - sc.half[2];
                const T ox = tmax(qx, T(0)), oy = tmax(qy, T(0)), oz = tmax(qz, T(0));
                const T s2 = ox * ox + oy * oy + oz * oz;
                const T mq = tmax(tmax(qx, qy), qz);
                const T key = s2 > T(0) ? s2 : mq;         // outside: squared distance (> 0); inside: largest face distance (<= 0)
                if (i < n_tip && key < best_key) { best_key = key; best_i = i; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {      // wave argmin; equal keys: the lower index, as a sequential scan finds it
                const T ok = __shfl_xor(best_key, off);
                const int oi = __shfl_xor(best_i, off);
                if (ok < best_key || (ok == best_key && oi < best_i)) { best_key = ok; best_i = oi; }
            }
            best_i = n_tip > 0 ? __builtin_amdgcn_readfirstlane(best_i) : 0;
            const V3<T> vv = mk(L[kLHull + 3 * best_i], L[kLHull + 3 * best_i + 1], L[kLHull + 3 * best_i + 2]);
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
            tip_depth = sdf - (sc.margin_tip + sc.margin_cube);
            tip_active = tip_depth <= sc.breaking && n_tip > 0;
            contact_code |= tip_active ? ((1 << 8) | (best_i << 9)) : 0;
            nrm = mul(b.R, mk(g[0], g[1], g[2]));   // from the cube towards the tip
            pa = w - sc.margin_tip * nrm; pb = w - (sdf - sc.margin_cube) * nrm;
        }
        V3<T> t1, t2;
        plane_space(nrm, t1, t2);
        tdir[0] = nrm; tdir[1] = t1; tdir[2] = t2;
        trb = pb - xc;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const V3<T> ai = mk(L[kLJa + 3 * i], L[kLJa + 3 * i + 1], L[kLJa + 3 * i + 2]);      // zero for joints off the tip's path
            const V3<T> oi = mk(L[kLJo + 3 * i], L[kLJo + 3 * i + 1], L[kLJo + 3 * i + 2]);
            jt[i] = cross(ai, pa - oi);
        }
    }
    TG_STAMP(2)
    const T denom = dt * sc.tip_stiffness + sc.tip_damping;   // soft contact: cfm = 1 / (dt (dt k + d)), erp = dt k / (dt k + d)
    const T cfm_tip = (T(1) / denom) / dt, erp_tip = dt * sc.tip_stiffness / denom;
    // ---- row of this lane: J = [ja | sg d | sg (rr x d)], W = Minv_sys J^T
    V3<T> d;
    {
        const V3<T> dt_ = rr_ == 0 ? tdir[0] : (rr_ == 1 ? tdir[1] : tdir[2]);
        const V3<T> dw = rr_ == 0 ? mk(T(0), T(0), T(1)) : (rr_ == 1 ? mk(T(0), T(-1), T(0)) : mk(T(1), T(0), T(0)));   // btPlaneSpace1 of +z
        d = tip_lane ? dt_ : dw;
    }
    const V3<T> rarm = tip_lane ? trb : my_ra;
    const T sg = tip_lane ? T(-1) : T(1);
    T ja[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const T jt_d = i < NP ? dot(jt[i < NP ? i : 0], d) : T(0);
        ja[i] = motor_lane ? (lane == i ? T(1) : T(0)) : (tip_lane ? jt_d : T(0));
    }
    T Warm[N];
    T rv = T(0), rm = T(0);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc += L[kLMinv + 8 * k + i] * ja[i];
        Warm[k] = acc;
        const T vk = L[kLV + k];
        rv += ja[k] * vk;
        const T pos_term = (MOTOR == kMotorPosition) ? kp * (L[kLQDes + k] - L[kLQ + k]) / dt : T(0);   // motor: velocity-level target minus
        const T des = pos_term + vk + kd * (L[kLQdDes + k] - vk);                                       // the unconstrained velocity
        rm = lane == k ? des - vk : rm;
    }
    const V3<T> Jl = contact_lane ? sg * d : mk<T>(0, 0, 0);
    const V3<T> Ja = contact_lane ? sg * cross(rarm, d) : mk<T>(0, 0, 0);
    const V3<T> Wl = invm * Jl, Wa = mul(Iwi, Ja);
    T A = dot(Jl, Wl) + dot(Ja, Wa);
    rv += dot(Jl, vb) + dot(Ja, wb);
#pragma unroll
    for (int i = 0; i < N; ++i) A += ja[i] * Warm[i];
    bool active;
    T rhs, cfm = T(0);
    {
        const T depth = tip_lane ? tip_depth : my_depth;
        const T erp = tip_lane ? erp_tip : sc.erp;
        const T rhs_n = (depth > T(0)) ? (-rv - depth / dt) : (-depth * erp / dt - rv);
        rhs = motor_lane ? rm : (rr_ == 0 ? rhs_n : -rv);
        if (tip_lane && rr_ == 0) cfm = cfm_tip;
        active = motor_lane ? (MOTOR != kMotorOff) : (tip_lane ? tip_active : (table_lane && cc < n_table));
    }
    const T jdi = active ? T(1) / (A + cfm) : T(0);
    const T adiag = active ? A + cfm : T(0);       // 1 / jacDiagABInv of this lane's row (threshold mode: a row's velocity change = delta * adiag)
    TG_STAMP(3)
    // ---- coefficient row of this lane, one entry per solver row i (motor i -> i, contact (c, r) -> 8 + 3c + r):
    //   row lanes        G[i] = (J_i . W_mine) / (A + cfm); own entry 1 on a motor / normal lane (it carries the residual r, which its
    //                    own relaxation takes down by the full delta), 0 on a friction lane (it carries s = lambda + r, invariant there)
    //   lanes 32 + k     G[i] = -W_i[k]: the lane accumulates component k of the velocity change  du = sum_i W_i lambda_i  (k < 14)
    //   lanes 48 + j     G[i] = -[i == j]: the lane accumulates the impulse of joint motor j (checked against the motor limit)
    // so that ONE v_fma_f64 per row step, x -= G[i] delta, advances the residuals, the velocities and the motor impulses together.
    // HN[c] = 1 on the three lanes of contact c: lamN += HN[c] delta keeps the normal impulse on its own lane AND on the contact's two friction
    // lanes (their cone limit is then lane-local); HF[c] = 1 on the two friction lanes: lamF += HF[c] dl.  In-lane impulse updates without masks.
    const bool fric_lane = contact_lane && rr_ > 0;
    const T own_diag = fric_lane ? T(0) : T(1);
    const int my_gi = motor_lane ? lane : (contact_lane ? 8 + 3 * cc + rr_ : -1);   // this lane's own row index
    T G[NGT], HN[NCT], HF[NCT];
#pragma unroll
    for (int i = 0; i < 8; ++i) G[i] = i < N ? Warm[i < N ? i : 0] * jdi : T(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int src = kContactLane0 + 4 * c;
        if (SHAPE == 1 && c > 0) { G[8 + 3 * c] = G[8 + 3 * c + 1] = G[8 + 3 * c + 2] = T(0); continue; }
        const T rax = bcast(my_ra.x, src), ray = bcast(my_ra.y, src), raz = bcast(my_ra.z, src);
        G[8 + 3 * c + 0] = (Wl.z + (ray * Wa.x - rax * Wa.y)) * jdi;         // n  = (0, 0, 1):  J = [n,  ra x n]
        G[8 + 3 * c + 1] = (-Wl.y + (raz * Wa.x - rax * Wa.z)) * jdi;        // t1 = (0,-1, 0)
        G[8 + 3 * c + 2] = (Wl.x + (raz * Wa.y - ray * Wa.z)) * jdi;         // t2 = (1, 0, 0)
    }
    if constexpr (NT == 1) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const V3<T> dr = tdir[r];
            T a = -dot(dr, Wl) - dot(cross(trb, dr), Wa);
#pragma unroll
            for (int i = 0; i < NP; ++i) a += dot(jt[i], dr) * Warm[i];
            G[8 + 12 + r] = a * jdi;
        }
    } else {
        // the twelve tip rows differ from lane to lane: every tip lane puts its J row (arm part, cube linear, cube angular) into LDS, every
        // lane then forms  J_r . W_mine  for the twelve of them from broadcast reads
        const int xj = xbase + kXJ;
        if (tip_lane) {
            const int row = 3 * ts_ + rr_;
#pragma unroll
            for (int k = 0; k < 8; ++k) L[xj + row * 16 + k] = k < N ? ja[k < N ? k : 0] : T(0);
            L[xj + row * 16 + 8] = Jl.x; L[xj + row * 16 + 9] = Jl.y; L[xj + row * 16 + 10] = Jl.z;
            L[xj + row * 16 + 11] = Ja.x; L[xj + row * 16 + 12] = Ja.y; L[xj + row * 16 + 13] = Ja.z;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            T a = T(0);
#pragma unroll
            for (int k = 0; k < N; ++k) a += L[xj + r * 16 + k] * Warm[k];
            a += L[xj + r * 16 + 8] * Wl.x + L[xj + r * 16 + 9] * Wl.y + L[xj + r * 16 + 10] * Wl.z;
            a += L[xj + r * 16 + 11] * Wa.x + L[xj + r * 16 + 12] * Wa.y + L[xj + r * 16 + 13] * Wa.z;
            G[8 + 12 + r] = ((tip_mask >> (r / 3)) & 1) ? a * jdi : T(0);
        }
    }
    {   // W rows to LDS (row-major [row][k], zero for rows that are not active), then the accumulator lanes pick up their columns
        if (my_gi >= 0) {
            const T keep_ = active ? T(1) : T(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) L[lw + my_gi * 16 + k] = k < N ? Warm[k < N ? k : 0] * keep_ : T(0);
            L[lw + my_gi * 16 + 8] = Wl.x * keep_; L[lw + my_gi * 16 + 9] = Wl.y * keep_; L[lw + my_gi * 16 + 10] = Wl.z * keep_;
            L[lw + my_gi * 16 + 11] = Wa.x * keep_; L[lw + my_gi * 16 + 12] = Wa.y * keep_; L[lw + my_gi * 16 + 13] = Wa.z * keep_;
        }
        __syncthreads();
        const int ku = lane - ACC0;
#pragma unroll
        for (int i = 0; i < NGT; ++i) {
            const T wv = L[lw + i * 16 + (ku >= 0 && ku < kNU ? ku : 0)];
            const T g_row = lane == (i < 8 ? i : kContactLane0 + 4 * ((i - 8) / 3) + (i - 8) % 3) ? own_diag : G[i];
            G[i] = (ku >= 0 && ku < kNU) ? -wv : ((lane >= W0 && lane < W0 + N) ? (i == lane - W0 ? T(-1) : T(0)) : (lane < ACC0 ? g_row : T(0)));
        }
    }
    // ---- joint-motor sweep as ONE linear map (the sweeps run the motors without their clamp, see below).  A Gauss-Seidel pass over the
    // motors k = 0 .. N-1 takes  dd_k = x[k]  (lane k, after the steps before it) and  x -= G_k dd_k  on every lane; with
    // L[i][k] = G_k on lane i that is  dd = T xm,  T = (I + strict_lower(L))^-1  (strict_upper for the reverse pass), xm = the motor
    // lanes' x before the pass, hence  x -= sum_m C_m xm[m]  with  C_m = sum_k G_k T[k][m]  per lane: N independent v_readlane + N v_fma
    // instead of N dependent (v_readlane -> v_fma) round trips.  Built once per tick (~10^3 cycles) for 150 sweeps.
    T Cf[N], Cr[N];
    if (MOTOR != kMotorOff) {
        __syncthreads();                 // the W rows in LDS have been consumed: the region is reused for L
        if (lane < N) {
#pragma unroll
            for (int k = 0; k < N; ++k) L[lw + lane * 8 + k] = G[k];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < N; ++m) {    // forward pass: column m of the unit lower triangular T, then C_m
            T t[N];
            t[m] = T(1);
            T c = G[m];
#pragma unroll
            for (int i = m + 1; i < N; ++i) {
                T acc = T(0);
#pragma unroll
                for (int k = m; k < i; ++k) acc = __builtin_fma(L[lw + i * 8 + k], t[k], acc);
                t[i] = -acc;
                c = __builtin_fma(G[i], t[i], c);
            }
            Cf[m] = c;
        }
#pragma unroll
        for (int m = 0; m < N; ++m) {    // reverse pass: unit upper triangular
            T t[N];
            t[m] = T(1);
            T c = G[m];
#pragma unroll
            for (int i = m - 1; i >= 0; --i) {
                T acc = T(0);
#pragma unroll
                for (int k = i + 1; k <= m; ++k) acc = __builtin_fma(L[lw + i * 8 + k], t[k], acc);
                t[i] = -acc;
                c = __builtin_fma(G[i], t[i], c);
            }
            Cr[m] = c;
        }
    }
#pragma unroll
    for (int c5 = 0; c5 < NCT; ++c5) {
        const bool mine = contact_lane && cc == c5;
        HN[c5] = mine ? T(1) : T(0);
        HF[c5] = (mine && rr_ > 0) ? T(1) : T(0);
    }
    const T x0 = lane < ACC0 ? rhs * jdi : T(0);
    int tip_i;                           // wave-uniform flag in an SGPR (s_cmp + s_cbranch_scc in the sweeps, no lane-mask round trip)
    if constexpr (NT == 1) { asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(tip_i) : "v"(tip_active ? 1 : 0)); }
    else tip_i = __builtin_amdgcn_readfirstlane(tip_mask);     // bit k: manifold slot k is live
    TG_PHASE_FENCE()
    TG_STAMP(4)
    // =============================================================== phase 3: 150 projected Gauss-Seidel sweeps
    // Per lane: x (the residual r on motor / normal lanes, s = lambda + r on friction lanes, an accumulator above lane 31) and lambda.
    // With one wavefront per SIMD nothing hides the latency of a dependent f64 instruction (~10 cycles; a v_readlane -> scalar operand ->
    // v_fma round trip ~30, measured), so a sweep costs its dependent chain and every step is written for depth:
    //     motor (no limit in reach)   v_readlane x2 -> v_fma
    //     motor (limits)              v_max -> v_min -> v_readlane x2 -> v_fma
    //     normal                      v_max(x, -lambda) -> v_readlane x2 -> v_fma
    //     friction pair               v_readlane x4 -> |s|^2 (2) -> v_rsq_f64 + one Newton step folded into limit / |s| (3) -> v_min(., 1)
    //                                 -> v_fma (delta) -> v_readlane x4 -> 2 v_fma
    // Motor rows and table-contact rows do not interact (their G entries are exact zeros: the table touches only the free body, a motor
    // only the arm), so the table normals are issued between the motor steps - same results bit for bit, their chains overlap.
    // Joint motors: the impulse limit max_force dt (1000 N m x 1/240 s) is ~1e4 times what these arms ever need, so the sweeps run without
    // the motor clamp while lanes 48.. watch every motor impulse of every sweep; a tick in which the limit would have been reached is
    // solved again with the clamped step (identical arithmetic otherwise).
    T x = x0, lam = T(0), lamF = T(0);   // lam: impulse of a motor / normal row (on friction lanes: their contact's normal impulse); lamF: friction rows
    const T maximp = max_force * dt;
#pragma unroll
    for (int i = 0; i < NGT; ++i) asm volatile("" : "+v"(G[i]));          // coefficient rows in architectural VGPRs for the sweeps (the allocator
#pragma unroll
    for (int i = 0; i < NCT; ++i) asm volatile("" : "+v"(HN[i]), "+v"(HF[i]));   // otherwise parks some of them in AGPRs: two v_accvgpr_read per use)
    T mu_lane = tip_lane ? sc.mu_tip : sc.mu_table;   // friction coefficient of this lane's contact (the cone limit is evaluated lane-locally)
    asm volatile("" : "+v"(mu_lane));
    const int n_it = __builtin_amdgcn_readfirstlane(iters < 0 ? -iters : iters);   // contact problems do not reach their fixed point within 150 sweeps: no exit test
    const bool watch_lane = lane >= W0 && lane < W0 + N;

#define TG_MOTOR_STEP(I)                                                               \
    {                                                                                  \
        T dd_;                                                                         \
        if (CLAMPED) {                                                                 \
            dd_ = bcast(vmin(vmax(x, -maximp - lam), maximp - lam), (I));              \
            lam = lane == (I) ? lam + dd_ : lam;                                       \
        } else dd_ = bcast(x, (I));                                                    \
        x = __builtin_fma(-G[(I)], dd_, x);                                            \
        if (THRM) { const T dvel_ = dd_ * Ad[(I)]; res_ = vmax(res_, dvel_ * dvel_); } \
    }
#define TG_NORMAL_STEP(ROW_LANE, GI)                                                   \
    {                                                                                  \
        const T dd_ = bcast(vmax_neg(x, lam), (ROW_LANE));                             \
        x = __builtin_fma(-G[(GI)], dd_, x);                                           \
        lam = __builtin_fma(HN[((GI) - 8) / 3], dd_, lam);                             \
        if (THRM) { const T dvel_ = dd_ * Ad[(GI)]; res_ = vmax(res_, dvel_ * dvel_); } \
    }
#define TG_FRICTION_STEP(NLANE, GI, MU)                                                \
    {                                                                                  \
        /* lane-local on the contact's two friction lanes (every other lane computes something finite that is never used): the partner's */ \
        /* s through a DPP quad permute, the normal impulse from the lane's own copy - no broadcast before the cone factor */ \
        T dl_;                                                                         \
        const T limit_ = mu_lane * lam;                                                \
        if (CONE) {                                                                    \
            const T xp_ = quad_swap12(x);                                              \
            const T tot2_ = __builtin_fma(xp_, xp_, x * x);                            \
            const T y_ = __builtin_amdgcn_rsq(tot2_);                                  \
            const T t_ = tot2_ * y_, ly_ = limit_ * y_;                                \
            const T lh_ = T(0.5) * ly_;                                                \
            const T e_ = __builtin_fma(-t_, y_, T(1));                                 \
            /* min(1, limit / |s|): one Newton step on the hardware seed (5e-8 -> 4e-15 relative, measured); |s| = 0 gives NaN, which */ \
            /* v_min_f64 drops in favour of the 1 */                                   \
            const T f_ = vmin(__builtin_fma(lh_, e_, ly_), T(1));                      \
            dl_ = __builtin_fma(x, f_, -lamF);                                         \
        } else {                                                                       \
            dl_ = vmin(vmax(x, -limit_), limit_) - lamF;                               \
        }                                                                              \
        lamF = __builtin_fma(HF[((GI) - 8) / 3], dl_, lamF);                           \
        const T d1_ = bcast(dl_, (NLANE) + 1), d2_ = bcast(dl_, (NLANE) + 2);          \
        x = __builtin_fma(-G[(GI)], d1_, x);                                           \
        x = __builtin_fma(-G[(GI) + 1], d2_, x);                                       \
        if (THRM) {   /* a cone pair counts once, with the sum of its two velocity changes (resolveConeFrictionConstraintRows, PARITY A7c) */ \
            const T v1_ = d1_ * Ad[(GI)], v2_ = d2_ * Ad[(GI) + 1];                    \
            if (CONE) { const T dvel_ = v1_ + v2_; res_ = vmax(res_, dvel_ * dvel_); } \
            else res_ = vmax(res_, vmax(v1_ * v1_, v2_ * v2_));                        \
        }                                                                              \
    }
    // one sweep: motors in forward (FWD) or reverse order with the table normals between them, tip normal, friction pairs
#define TG_SWEEP(FWD)                                                                  \
    {                                                                                  \
        if (MOTOR != kMotorOff && !CLAMPED) {                                          \
            /* motors: one linear map from the motor lanes' x (chain: readlanes -> two FMA chains); table normals: their own chain on */ \
            /* x (motor and table rows do not interact: the C entries on table lanes and the table G entries on motor lanes are exact zeros) */ \
            T xm_[N];                                                                  \
            _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_) xm_[k_] = bcast(x, k_);   \
            T da_ = T(0), db_ = T(0);                                                  \
            _Pragma("unroll") for (int k_ = 0; k_ < N; k_ += 2) {                      \
                da_ = __builtin_fma((FWD) ? Cf[k_] : Cr[k_], xm_[k_], da_);            \
                if (k_ + 1 < N) db_ = __builtin_fma((FWD) ? Cf[k_ + 1] : Cr[k_ + 1], xm_[k_ + 1], db_); \
            }                                                                          \
            _Pragma("unroll") for (int k_ = 0; k_ < (SHAPE == 1 ? 1 : 4); ++k_) TG_NORMAL_STEP(kContactLane0 + 4 * k_, 8 + 3 * k_) \
            x -= da_ + db_;                                                            \
            /* threshold mode: watch lane W0 + j carries G = -e_j, so its share of the map, -(da_ + db_), IS motor j's impulse change of this pass */ \
            if (THRM) { const T dvm_ = (da_ + db_) * adw; resl_ = vmax(resl_, dvm_ * dvm_); } \
        } else if (MOTOR != kMotorOff) {                                               \
            _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_) {                         \
                const int i_ = (FWD) ? k_ : N - 1 - k_;                                \
                TG_MOTOR_STEP(i_)                                                      \
                if (k_ < (SHAPE == 1 ? 1 : 4)) TG_NORMAL_STEP(kContactLane0 + 4 * k_, 8 + 3 * k_) \
            }                                                                          \
        } else {                                                                       \
            _Pragma("unroll") for (int k_ = 0; k_ < (SHAPE == 1 ? 1 : 4); ++k_) TG_NORMAL_STEP(kContactLane0 + 4 * k_, 8 + 3 * k_) \
        }                                                                              \
        if (CLAMPED == 0) wmax = vmax_abs(wmax, x);   /* lanes 48..: the largest motor impulse any sweep has seen */ \
        if (NT == 1) { if (tip_i) TG_NORMAL_STEP(kContactLane0 + 16, 8 + 12) }         \
        else { _Pragma("unroll") for (int k_ = 0; k_ < NT; ++k_) if (tip_i & (1 << k_)) TG_NORMAL_STEP(kContactLane0 + 16 + 4 * k_, 8 + 12 + 3 * k_) } \
        _Pragma("unroll") for (int k_ = 0; k_ < (SHAPE == 1 ? 1 : 4); ++k_) TG_FRICTION_STEP(kContactLane0 + 4 * k_, 8 + 3 * k_ + 1, mu_table) \
        if (NT == 1) { if (tip_i) TG_FRICTION_STEP(kContactLane0 + 16, 8 + 13, mu_tip) } \
        else { _Pragma("unroll") for (int k_ = 0; k_ < NT; ++k_) if (tip_i & (1 << k_)) TG_FRICTION_STEP(kContactLane0 + 16 + 4 * k_, 8 + 13 + 3 * k_, mu_tip) } \
    }
#define TG_SOLVE()                                                                     \
    {                                                                                  \
        int it_ = 0;                                                                   \
        for (; it_ + 1 < n_it; it_ += 2) { TG_SWEEP(false) TG_SWEEP(true) } \
        if (it_ < n_it) TG_SWEEP(false)                                                \
    }
    T wmax = T(0);
    T res_ = T(0);
    if (uniform_true(m.res_thr > T(0))) {
        // Threshold mode (tg_config.solver_residual_threshold, PARITY A7b): the literal clamped row steps, and after EVERY sweep Bullet's exit -
        // the largest squared velocity change delta / jacDiagABInv of the sweep's row updates <= the threshold (oracle mb_step_push).  Every
        // delta is wave-uniform after its broadcast; the rows' 1 / jacDiagABInv are fetched once from their lanes.
        // The motors run unclamped as ONE linear map per pass, as in the default mode (their impulse changes are read off the watch lanes, `adw` =
        // the motor's 1 / jacDiagABInv on its watch lane, 0 elsewhere); a tick in which a motor limit is reached is solved again with the literal
        // clamped steps.  Contact deltas are wave-uniform after their broadcast; the rows' 1 / jacDiagABInv are fetched once from their lanes.
        constexpr int THRM = 1;
        T Ad[NGT];
#pragma unroll
        for (int i = 0; i < NGT; ++i) Ad[i] = bcast(adiag, i < 8 ? i : kContactLane0 + 4 * ((i - 8) / 3) + (i - 8) % 3);
        const T adw_f = lane_fetch(adiag, watch_lane ? lane - W0 : 0);
        const T adw = watch_lane ? adw_f : T(0);
        T resl_ = T(0);
        int ran_ = 0;
        {
            constexpr int CLAMPED = 0;
            for (int it_ = 0; it_ < n_it; ++it_) {
                res_ = T(0); resl_ = T(0);
                if (it_ & 1) { TG_SWEEP(true) } else { TG_SWEEP(false) }
                ++ran_;
                if (__builtin_amdgcn_ballot_w64(!(res_ <= m.res_thr) || !(resl_ <= m.res_thr)) == 0) break;
            }
        }
        if (__builtin_amdgcn_ballot_w64(watch_lane && wmax > maximp) != 0) {
            constexpr int CLAMPED = 1;
            x = x0; lam = T(0); lamF = T(0); ran_ = 0;
            for (int it_ = 0; it_ < n_it; ++it_) {
                res_ = T(0);
                if (it_ & 1) { TG_SWEEP(true) } else { TG_SWEEP(false) }
                ++ran_;
                if (uniform_true(res_ <= m.res_thr)) break;
            }
        }
        if (sweep_acc != nullptr) *sweep_acc += ran_;
    } else {
    constexpr int THRM = 0;
    const T* Ad = nullptr;
    const T adw = T(0);
    T resl_ = T(0);
    {
        constexpr int CLAMPED = 0;
        TG_SOLVE()
    }
    const uint64_t viol = __builtin_amdgcn_ballot_w64(watch_lane && wmax > maximp);
    if (viol != 0) {                      // a motor impulse reached its limit somewhere in this tick: the literal clamped iteration
        constexpr int CLAMPED = 1;
        x = x0; lam = T(0); lamF = T(0);
        TG_SOLVE()
    }
    (void)Ad; (void)adw; (void)resl_;
    }
#undef TG_SOLVE
#undef TG_SWEEP
#undef TG_FRICTION_STEP
#undef TG_NORMAL_STEP
#undef TG_MOTOR_STEP
    TG_STAMP(5)
    // =============================================================== phase 4: integrate (lanes 32 .. 45 hold du)
    T du[kNU];
#pragma unroll
    for (int k = 0; k < kNU; ++k) du[k] = bcast(x, ACC0 + k);
    asm volatile("" ::: "memory");
    {
        T q[N], dq[N], qd[N];
        JointTrig<T, N> trig;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            q[i] = L[kLQ + i]; trig.s[i] = L[kLTrigS + i]; trig.c[i] = L[kLTrigC + i];
            qd[i] = L[kLV + i] + du[i];
            dq[i] = dt * qd[i];
            q[i] += dq[i];
        }
        trig_advance<T, N>(q, dq, trig);
        FreeBody<T> bn;
#pragma unroll
        for (int e = 0; e < 9; ++e) bn.R.m[e] = L[kLBody + 3 + e];
        bn.v = mk(L[kLFree + 0], L[kLFree + 1], L[kLFree + 2]) + mk(du[8], du[9], du[10]);
        bn.w = mk(L[kLFree + 3], L[kLFree + 4], L[kLFree + 5]) + mk(du[11], du[12], du[13]);
        const V3<T> xn = mk(L[kLFree + 6], L[kLFree + 7], L[kLFree + 8]) + dt * bn.v;
        integrate_rotation(bn.R, bn.w, dt);
        bn.pos = xn - mul(bn.R, load_v3(sc.com));
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) { L[kLQ + i] = q[i]; L[kLQd + i] = qd[i]; L[kLTrigS + i] = trig.s[i]; L[kLTrigC + i] = trig.c[i]; }
            L[kLBody + 0] = bn.pos.x; L[kLBody + 1] = bn.pos.y; L[kLBody + 2] = bn.pos.z;
#pragma unroll
            for (int e = 0; e < 9; ++e) L[kLBody + 3 + e] = bn.R.m[e];
            L[kLBody + 12] = bn.v.x; L[kLBody + 13] = bn.v.y; L[kLBody + 14] = bn.v.z;
            L[kLBody + 15] = bn.w.x; L[kLBody + 16] = bn.w.y; L[kLBody + 17] = bn.w.z;
        }
    }
    TG_PHASE_FENCE()
    TG_STAMP(6)
    return contact_code;
}

// BaseTactileEnv.step (base_tactile_env.py:166-185) for object_push (SHAPE 0) / object_roll (SHAPE 1), one wavefront per env.
template <typename T, int TOPO, bool POS, int SHAPE, bool CONE, int NT = 1>
__global__ __launch_bounds__(64) void k_step_contact_wave(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                          const float* __restrict__ actions) {
    // (no KtScope: its one more live scalar pair took this kernel from 0 to 204 B of scratch per lane - 15 MB of spill traffic per launch by the
    //  PMC - at a duration in milliseconds, where a HIP event pair's 5 us do not matter)
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double wave_lds_raw[];
    const lds_ptr<T> L = (lds_ptr<T>)wave_lds_raw;
    const DevRobot<T>& m = *mp;
    const int env = blockIdx.x, lane = threadIdx.x;
    const EnvConst<T>& c = *cp;
    const int n = c.num_envs;
    const T mass_or_radius = (T)st.obj_mass[env];
    T work_dz = T(0);
    if constexpr (SHAPE == 1) work_dz = (T)((2.0 * st.obj_mass[env] - st.embed[env]) - (double)c.work_pos[2]);   // update_workframe (object_roll_env.py:192-201)
    const int step_count = st.step_count[env] + 1;
    if constexpr (SHAPE == 0) {          // tip-core hull vertices: HBM -> LDS once per step (coalesced)
        const T* tipv = (const T*)st.tip_verts;
        const int nw = 3 * c.push.n_tip;
        for (int w = lane; w < nw; w += 64) L[kLHull + w] = tipv[w];
    }
    const int xbase = kLHull + 3 * c.push.n_tip;          // NT = 4: scratch / W rows / manifold behind the hull
    if constexpr (NT == 4) {             // the tip - cube manifold lives in LDS for the step's ticks
        if (lane < 37) L[xbase + kXMani + (lane < 36 ? lane : narrow::kMcount)] = (T)st.mani[(size_t)lane * n + env];
    }
    stage_link_constants<T, TOPO>(mp, L, lane);
    V3<T> tpos; Q4<T> tq;                // TCP_position_control: the pose target of the blocking move
    {   // ---- controller (once per step): action -> joint targets, state -> LDS
        T q[N], qd[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
        const FreeBody<T> b = load_body<T>(st, n, env);
        const float* a = actions + (size_t)env * c.act_dim;
        T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        if constexpr (SHAPE == 1) {                           // encode_actions "xy" (object_roll_env.py:297-309)
            enc[0] = (T)a[0]; enc[1] = (T)a[1];
        } else {                                              // encode_actions (object_push_env.py:372-454)
            if (c.movement_mode == TG_PMOVE_Y) { enc[0] = c.max_action; enc[1] = (T)a[0]; }
            else if (c.movement_mode == TG_PMOVE_YRZ) { enc[0] = c.max_action; enc[1] = (T)a[0]; enc[5] = (T)a[1]; }
            else if (c.movement_mode == TG_PMOVE_XYRZ) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[5] = (T)a[2]; }
            else {                                            // TCP-frame moves: along / across the sensor's pointing direction
                Kin<T, TOPO> k;
                forward_kinematics<T, TOPO>(m, q, k);
                V3<T> ptcp; M3<T> Rtcp;
                link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
                const M3<T> Rq = mat_from_quat(quat_from_mat(Rtcp));
                const V3<T> par = mul(c.work_Rinv, mul(Rq, mk(T(1), T(0), T(0)))), perp = mul(c.work_Rinv, mul(Rq, mk(T(0), T(-1), T(0))));
                if (c.movement_mode == TG_PMOVE_TYRZ) {
                    const T pa_ = T(1) * c.max_action;
                    enc[0] += perp.x * (T)a[0] + par.x * pa_;
                    enc[1] += perp.y * (T)a[0] + par.y * pa_;
                    enc[5] += (T)a[1];
                } else {
                    enc[0] += perp.x * (T)a[1] + par.x * (T)a[0];
                    enc[1] += perp.y * (T)a[1] + par.y * (T)a[0];
                    enc[5] += (T)a[2];
                }
            }
        }
        T vels[6];
        scale_actions<T>(c, enc, vels);
        T qd_des[N];
        if constexpr (POS) tcp_position_target<T, TOPO>(m, c, q, vels, tpos, tq, qd_des, work_dz);   // qd_des carries the joint targets
        else tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des, nullptr, work_dz);
        JointTrig<T, N> trig;
        trig_init<T, N>(q, trig);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool in = i < N;
                L[kLQ + i] = in ? q[in ? i : 0] : T(0); L[kLQd + i] = in ? qd[in ? i : 0] : T(0);
                L[kLTrigS + i] = in ? trig.s[in ? i : 0] : T(0); L[kLTrigC + i] = in ? trig.c[in ? i : 0] : T(1);
                L[kLQDes + i] = (POS && in) ? qd_des[in ? i : 0] : T(0);
                L[kLQdDes + i] = (!POS && in) ? qd_des[in ? i : 0] : T(0);
            }
            L[kLBody + 0] = b.pos.x; L[kLBody + 1] = b.pos.y; L[kLBody + 2] = b.pos.z;
#pragma unroll
            for (int e = 0; e < 9; ++e) L[kLBody + 3 + e] = b.R.m[e];
            L[kLBody + 12] = b.v.x; L[kLBody + 13] = b.v.y; L[kLBody + 14] = b.v.z;
            L[kLBody + 15] = b.w.x; L[kLBody + 16] = b.w.y; L[kLBody + 17] = b.w.z;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) st.qd_target[i * n + env] = POS ? 0.0 : (double)qd_des[i];
    }
    TG_PHASE_FENCE()
    int ccode = 0, sweeps = 0;
    if constexpr (POS) {                                  // blocking_move(max_steps, constant_vel=None), robot.py:188-260
        for (int t = 0; t < c.max_blocking; ++t) {
            T q[N], qd[N];
#pragma unroll
            for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; }
            const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
            ccode = sim_tick_contact_wave<T, TOPO, kMotorPosition, SHAPE, CONE, NT>(m, c.push, L, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters,
                                                                                    mass_or_radius, lane, xbase, &sweeps);
            if (uniform_true(stop)) break;
        }
    } else {
        for (int t = 0; t < c.action_repeat; ++t)
            ccode = sim_tick_contact_wave<T, TOPO, kMotorVelocity, SHAPE, CONE, NT>(m, c.push, L, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters,
                                                                                    mass_or_radius, lane, xbase, &sweeps);
    }
    if constexpr (NT == 4) {
        if (lane < 37) st.mani[(size_t)lane * n + env] = (double)L[xbase + kXMani + (lane < 36 ? lane : narrow::kMcount)];
    }
    // ---- results: every lane holds the same values; identical values go to identical addresses
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; }
    FreeBody<T> b;
    b.pos = mk(L[kLBody + 0], L[kLBody + 1], L[kLBody + 2]);
#pragma unroll
    for (int e = 0; e < 9; ++e) b.R.m[e] = L[kLBody + 3 + e];
    b.v = mk(L[kLBody + 12], L[kLBody + 13], L[kLBody + 14]);
    b.w = mk(L[kLBody + 15], L[kLBody + 16], L[kLBody + 17]);
    st.step_count[env] = step_count;
    st.contact_code[env] = ccode;
    if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; }
    store_body<T>(st, n, env, b);
    if constexpr (SHAPE == 1) finish_roll<T, TOPO>(m, c, st, env, q, b, mass_or_radius / (T)c.roll_radius, step_count, true);
    else finish_push<T, TOPO>(m, c, st, env, q, b, step_count, true);
}

// ---- object_balance: the full tick (arm + pole + point-to-point constraint) on one wavefront ---------------------------------------------
// sim_tick_body's full solve (tg_physics.hpp) evaluates the arm dynamics, the (N + 3)^2 Delassus matrix and up to 150 Gauss-Seidel sweeps
// over N motor rows and 3 P2P rows as one scalar program per lane: ~80 k cycles, which a lane-per-env wavefront pays in every env step as
// soon as ONE of its 64 envs has no licence for the analytic fixed point (after a reset, every 8th step) - in object_balance that is
// always.  Here the tick runs on the env's own wavefront: tick_dynamics_lanes for the arm, one solver row per lane (lane i < N motor i,
// lane 8 + x the P2P row along world axis x), residual form with accumulator lanes as in sim_tick_contact_wave.  No row of this system
// has a one-sided limit, so while no impulse reaches its bound (watched on lanes 48..) a whole sweep is ONE linear map of the row lanes'
// residuals, x -= sum_m C_m x_m, with C built once per tick for the forward and the reverse row order (see the motor pass above).
// Same convergence exit and licence rule as sim_tick_body; returns the new value of `verified` (24: converged within 80 % of the sweep
// budget, 0: not).  State in / out through LDS like sim_tick_contact_wave.
template <typename T, int TOPO, int MOTOR>
__device__ __forceinline__ int sim_tick_p2p_wave(const DevRobot<T>& m, const BodyConst<T>& bc, lds_ptr<T> L, T kp, T kd, T max_force, T dt, int iters,
                                                 V3<T> gravity, V3<T> pivot_b, V3<T> ext_force, V3<T> ext_pos, bool ext_pending, int lane_in,
                                                 int* sweep_acc = nullptr /* threshold mode: += the sweeps this tick ran */) {
    constexpr int N = Topo<TOPO>::N;
    constexpr int NP = Topo<TOPO>::NP;
    constexpr int NR = N + 3;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
#if TG_WAVE_TIMING
    unsigned long long t_prev_ = __builtin_readcyclecounter();
    tick_dynamics_lanes<T, TOPO>(&m, L, bc.link, dt, lane, t_prev_);
#else
    tick_dynamics_lanes<T, TOPO>(&m, L, bc.link, dt, lane);
#endif
    TG_PHASE_FENCE()
    // ---- pole: unconstrained velocities, pivots
    FreeBody<T> b;
    b.pos = mk(L[kLBody + 0], L[kLBody + 1], L[kLBody + 2]);
#pragma unroll
    for (int e = 0; e < 9; ++e) b.R.m[e] = L[kLBody + 3 + e];
    b.v = mk(L[kLBody + 12], L[kLBody + 13], L[kLBody + 14]);
    b.w = mk(L[kLBody + 15], L[kLBody + 16], L[kLBody + 17]);
    const S3<T> Iw = rotate(b.R, bc.inertia), Iwi = inverse(Iw);
    const V3<T> xc = b.pos + mul(b.R, bc.com);
    V3<T> F = bc.mass * gravity, Nt = mk<T>(0, 0, 0);
    if (ext_pending) { F = F + ext_force; Nt = Nt + cross(ext_pos - xc, ext_force); }
    Nt = Nt - cross(b.w, mul(Iw, b.w));
    const T invm = T(1) / bc.mass;
    const V3<T> vb = b.v + (dt * invm) * F, wb = b.w + dt * mul(Iwi, Nt);
    const V3<T> ol = mk(L[kLTipF + 0], L[kLTipF + 1], L[kLTipF + 2]);
    M3<T> Rl;
#pragma unroll
    for (int e = 0; e < 9; ++e) Rl.m[e] = L[kLTipF + 3 + e];
    const V3<T> pa = ol + mul(Rl, bc.pivot_a), pb = b.pos + mul(b.R, pivot_b), rb = pb - xc;
    V3<T> jt[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        jt[i] = cross(mk(L[kLJa + 3 * i], L[kLJa + 3 * i + 1], L[kLJa + 3 * i + 2]), pa - mk(L[kLJo + 3 * i], L[kLJo + 3 * i + 1], L[kLJo + 3 * i + 2]));
    // ---- this lane's row: J = [ja | -e_x | -(rb x e_x)] on a P2P lane, e_i on a motor lane; W = Minv_sys J^T
    const bool motor_lane = lane < N, p2p_lane = lane >= 8 && lane < 11;
    const int ax = lane - 8;
    const V3<T> ex = mk(ax == 0 ? T(1) : T(0), ax == 1 ? T(1) : T(0), ax == 2 ? T(1) : T(0));
    T ja[N];
#pragma unroll
    for (int i = 0; i < N; ++i) ja[i] = motor_lane ? (lane == i ? T(1) : T(0)) : ((p2p_lane && i < NP) ? dot(jt[i < NP ? i : 0], ex) : T(0));
    T Warm[N], rv = T(0), rm = T(0);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc += L[kLMinv + 8 * k + i] * ja[i];
        Warm[k] = acc;
        const T vk = L[kLV + k];
        rv += ja[k] * vk;
        const T pos_term = (MOTOR == kMotorPosition) ? kp * (L[kLQDes + k] - L[kLQ + k]) / dt : T(0);
        const T des = pos_term + vk + kd * (L[kLQdDes + k] - vk);
        rm = lane == k ? des - vk : rm;
    }
    const V3<T> Jl = p2p_lane ? mk<T>(0, 0, 0) - ex : mk<T>(0, 0, 0), Ja = p2p_lane ? mk<T>(0, 0, 0) - cross(rb, ex) : mk<T>(0, 0, 0);
    const V3<T> Wl = invm * Jl, Wa = mul(Iwi, Ja);
    T A = dot(Jl, Wl) + dot(Ja, Wa);
#pragma unroll
    for (int i = 0; i < N; ++i) A += ja[i] * Warm[i];
    rv += dot(Jl, vb) + dot(Ja, wb);                      // J . (unconstrained velocity): va - (vb + wb x rb) on a P2P lane
    const V3<T> gap = pa - pb;
    const T rhs = motor_lane ? rm : (-bc.erp * dot(gap, ex) / dt - rv);
    const bool active = motor_lane ? (MOTOR != kMotorOff) : p2p_lane;
    const T jdi = active ? T(1) / A : T(0);
    // ---- coefficient rows (row index i < N: motor i, N + x: P2P axis x), accumulator lanes 32.. (du: 8 joints, pole linear 3, angular 3),
    //      watch lanes 48 + i (impulse of row i)
    T G[NR];
#pragma unroll
    for (int i = 0; i < N; ++i) G[i] = Warm[i] * jdi;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const V3<T> e = mk(x == 0 ? T(1) : T(0), x == 1 ? T(1) : T(0), x == 2 ? T(1) : T(0));
        T a = -dot(e, Wl) - dot(cross(rb, e), Wa);
#pragma unroll
        for (int i = 0; i < NP; ++i) a += dot(jt[i], e) * Warm[i];
        G[N + x] = a * jdi;
    }
    const int my_row = motor_lane ? lane : (p2p_lane ? N + ax : -1);
    {
        __syncthreads();
        if (my_row >= 0) {
            const T keep_ = active ? T(1) : T(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) L[kLW + my_row * 16 + k] = k < N ? Warm[k < N ? k : 0] * keep_ : T(0);
            L[kLW + my_row * 16 + 8] = Wl.x * keep_; L[kLW + my_row * 16 + 9] = Wl.y * keep_; L[kLW + my_row * 16 + 10] = Wl.z * keep_;
            L[kLW + my_row * 16 + 11] = Wa.x * keep_; L[kLW + my_row * 16 + 12] = Wa.y * keep_; L[kLW + my_row * 16 + 13] = Wa.z * keep_;
        }
        __syncthreads();
        const int ku = lane - 32;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const T wv = L[kLW + i * 16 + (ku >= 0 && ku < kNU ? ku : 0)];
            const T g_row = my_row == i ? T(1) : G[i];
            G[i] = (ku >= 0 && ku < kNU) ? -wv : ((lane >= 48 && lane < 48 + NR) ? (i == lane - 48 ? T(-1) : T(0)) : (my_row >= 0 ? g_row : T(0)));
        }
    }
    // ---- one Gauss-Seidel pass over the NR rows as a linear map of the row lanes' residuals, forward and reverse order
    const bool thr_mode = uniform_true(m.res_thr > T(0));    // tg_config.solver_residual_threshold: the literal row-by-row iteration below, no maps
    T Cf[NR], Cr[NR];
    if (!thr_mode) {
        __syncthreads();
        if (my_row >= 0) {
#pragma unroll
            for (int k = 0; k < NR; ++k) L[kLW + my_row * 16 + k] = G[k];     // L[i][k] = G_k on the lane of row i
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < NR; ++mm) {
            T t[NR];
            t[mm] = T(1);
            T c = G[mm];
#pragma unroll
            for (int i = mm + 1; i < NR; ++i) {
                T acc = T(0);
#pragma unroll
                for (int k = mm; k < i; ++k) acc = __builtin_fma(L[kLW + i * 16 + k], t[k], acc);
                t[i] = -acc;
                c = __builtin_fma(G[i], t[i], c);
            }
            Cf[mm] = c;
        }
#pragma unroll
        for (int mm = 0; mm < NR; ++mm) {
            T t[NR];
            t[mm] = T(1);
            T c = G[mm];
#pragma unroll
            for (int i = mm - 1; i >= 0; --i) {
                T acc = T(0);
#pragma unroll
                for (int k = i + 1; k <= mm; ++k) acc = __builtin_fma(L[kLW + i * 16 + k], t[k], acc);
                t[i] = -acc;
                c = __builtin_fma(G[i], t[i], c);
            }
            Cr[mm] = c;
        }
    }
    const T x0 = my_row >= 0 ? rhs * jdi : T(0);
    T thr = tabs(x0);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) thr = tmax(thr, __shfl_xor(thr, o));     // rows live in lanes 0..10: max over the first 16 lanes
    thr = bcast(thr, 0);
    thr = iters < 0 ? T(-1) : thr * (sizeof(T) == 8 ? T(1.3877787807814457e-17) : T(7.450580596923828e-09));
    const int n_it = iters < 0 ? -iters : iters;
    const T lim = lane >= 48 + N ? bc.max_impulse : max_force * dt;           // watch lanes: motor rows, then the P2P rows
    const bool watch_lane = lane >= 48 && lane < 48 + NR;
    auto row_lane = [](int i) { return i < N ? i : 8 + (i - N); };
    T x = x0, wmax = T(0);
    int conv_sweeps = -1;
    if (thr_mode) {
        // Threshold mode: Bullet's exit after every sweep (oracle mb_step_body).  The literal clamped row step; its delta `dd` is wave-uniform,
        // the row's velocity change is dd / jacDiagABInv = dd A_ii with the diagonals fetched once from the row lanes.
        T Ad[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) Ad[i] = bcast(A, row_lane(i));
        T lam = T(0);
        const T limr = p2p_lane ? bc.max_impulse : max_force * dt;
        int ran = 0;
        for (int it = 0; it < n_it; ++it) {
            T res = T(0);
#pragma unroll
            for (int kk = 0; kk < NR; ++kk) {
                const int i = (it & 1) ? kk : NR - 1 - kk;
                const T dd = bcast(vmin(vmax(x, -limr - lam), limr - lam), row_lane(i));
                lam = my_row == i ? lam + dd : lam;
                x = __builtin_fma(-G[i], dd, x);
                const T dvel = dd * Ad[i];
                res = tmax(res, dvel * dvel);
            }
            ++ran;
            if (uniform_true(res <= m.res_thr)) break;
        }
        if (sweep_acc != nullptr) *sweep_acc += ran;
    } else
    for (int it = 0; it < n_it; ++it) {
        if ((it & 7) == 0 && it > 0) {                    // the exit of sim_tick_body: every residual below 2^-56 of the largest start value
            T mx = my_row >= 0 ? tabs(x) : T(0);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = tmax(mx, __shfl_xor(mx, o));
            if (uniform_true(bcast(mx, 0) <= thr)) { conv_sweeps = it; break; }
        }
        T xm[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) xm[k] = bcast(x, row_lane(k));
        T da = T(0), db = T(0);
        if (it & 1) {
#pragma unroll
            for (int k = 0; k < NR; k += 2) { da = __builtin_fma(Cf[k], xm[k], da); if (k + 1 < NR) db = __builtin_fma(Cf[k + 1], xm[k + 1], db); }
        } else {
#pragma unroll
            for (int k = 0; k < NR; k += 2) { da = __builtin_fma(Cr[k], xm[k], da); if (k + 1 < NR) db = __builtin_fma(Cr[k + 1], xm[k + 1], db); }
        }
        x -= da + db;
        wmax = vmax_abs(wmax, x);
    }
    if (!thr_mode && uniform_true(watch_lane && wmax > lim)) {
        // an impulse reached its bound: the literal clamped iteration, row by row (never seen with the reference's limits; kept for safety)
        x = x0;
        T lam = T(0);
        conv_sweeps = -1;
        const T limr = p2p_lane ? bc.max_impulse : max_force * dt;
        for (int it = 0; it < n_it; ++it) {
#pragma unroll
            for (int kk = 0; kk < NR; ++kk) {
                const int i = (it & 1) ? kk : NR - 1 - kk;
                const T dd = bcast(vmin(vmax(x, -limr - lam), limr - lam), row_lane(i));
                lam = my_row == i ? lam + dd : lam;
                x = __builtin_fma(-G[i], dd, x);
            }
        }
    }
    // ---- integrate
    T du[kNU];
#pragma unroll
    for (int k = 0; k < kNU; ++k) du[k] = bcast(x, 32 + k);
    {
        T q[N], qd[N];
        JointTrig<T, N> trig;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            qd[i] = L[kLV + i] + du[i];
            q[i] = L[kLQ + i] + dt * qd[i];
        }
        trig_init_lanes<T, N>(q, trig, lane);             // a full tick re-anchors the carried sines / cosines exactly
        FreeBody<T> bn;
        bn.R = b.R;
        bn.v = vb + mk(du[8], du[9], du[10]);
        bn.w = wb + mk(du[11], du[12], du[13]);
        const V3<T> xn = xc + dt * bn.v;
        integrate_rotation(bn.R, bn.w, dt);
        bn.pos = xn - mul(bn.R, bc.com);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) { L[kLQ + i] = q[i]; L[kLQd + i] = qd[i]; L[kLTrigS + i] = trig.s[i]; L[kLTrigC + i] = trig.c[i]; }
            L[kLBody + 0] = bn.pos.x; L[kLBody + 1] = bn.pos.y; L[kLBody + 2] = bn.pos.z;
#pragma unroll
            for (int e = 0; e < 9; ++e) L[kLBody + 3 + e] = bn.R.m[e];
            L[kLBody + 12] = bn.v.x; L[kLBody + 13] = bn.v.y; L[kLBody + 14] = bn.v.z;
            L[kLBody + 15] = bn.w.x; L[kLBody + 16] = bn.w.y; L[kLBody + 17] = bn.w.z;
        }
    }
    TG_PHASE_FENCE()
    return (iters > 0 && conv_sweeps > 0 && 5 * conv_sweeps <= 4 * iters) ? 24 : 0;
}

// BaseTactileEnv.step for object_balance (TCP_velocity_control), one wavefront per env: the licence for the analytic fixed point is the env's
// own (wave-uniform here), so the 12 ticks of a licensed step are the analytic tick (body_tick_analytic, evaluated by every lane alike) and a
// full tick - after a reset, every 8th step, or when the analytic tick's a-priori test fails - is sim_tick_p2p_wave.
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_step_body_wave(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                       const float* __restrict__ actions, int inline_reset) {
    KtScope kt_scope_(st.kt);
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double wave_lds_raw[];
    const lds_ptr<T> L = (lds_ptr<T>)wave_lds_raw;
    const int env = blockIdx.x, lane = threadIdx.x;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int n = (int)gridDim.x;         // = c.num_envs (launch_step_body_wave): the state loads below do not wait for the constants' scalar loads
    const bool w0 = lane == 0;
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    FreeBody<T> b = load_body<T>(st, n, env);
    T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};      // encode_actions (object_balance_env.py:398-424)
    float abuf[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    unsigned long long ticket = 0;
    if (st.draw != nullptr) {
        // tg_step_random: action_space.sample() for this env inside the step (element i = env * act_dim + j of draw `counter`: the arithmetic
        // of k_sample_actions, as in step_env) - the sampler as a launch of its own was 4.7 us + a graph gap in front of every step.  The
        // wavefront takes its ticket for the counter's election once its draws are computed and looks at it at the end of the launch.
        const uint64_t counter = st.draw[0] + 1, seed = st.draw[1];
        const float lo = (float)c.min_action, hi = (float)c.max_action;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (j < c.act_dim) {
                const int i = env * c.act_dim + j;
                const uint64_t z = mix64(mix64(seed + kGolden * (counter + 1)) + kGolden * (uint64_t)(i + 1));
                const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
                abuf[j] = lo + (hi - lo) * u;
                if (w0) st.act_out[i] = abuf[j];
            }
        }
        if (w0) ticket = draw_ticket_take(st, abuf[0]);
    } else {
        const float* ap = actions + (size_t)env * c.act_dim;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < c.act_dim) abuf[j] = ap[j];
    }
    const float* a = abuf;
    if (c.movement_mode == TG_BMOVE_XY) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; }
    else if (c.movement_mode == TG_BMOVE_XYZ) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[2] = (T)a[2]; }
    else if (c.movement_mode == TG_BMOVE_RXRY) { enc[3] = (T)a[0]; enc[4] = (T)a[1]; }
    else { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[3] = (T)a[2]; enc[4] = (T)a[3]; }
    T vels[6];
    scale_actions<T>(c, enc, vels);
    const int step_count = st.step_count[env] + 1;
    T qd_des[N];
    JointTrig<T, N> trig;
    trig_init_lanes<T, N>(q, trig, lane);
    tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des, &trig);
    const T embed = (T)st.embed[env];
    const V3<T> grav = mk(T(0), T(0), (T)st.gravity[env]);
    const V3<T> pivot_b = mk(T(0), T(0), -c.obj_base_height / T(2) + embed);
    const V3<T> fext = load_v3(c.ext_force);
    const V3<T> pext = mk((T)st.ext_pos[0 * n + env], (T)st.ext_pos[1 * n + env], (T)st.ext_pos[2 * n + env]);
    const bool pending = st.ext_pending[env] != 0;
    const int lic = st.licence[env];
    bool staged = false;                  // the full tick's per-link constants go to LDS only when a full tick is due
    T qdummy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qdummy[i] = T(0);
    int verified = lic > 0 ? 24 : 0, sweeps = 0;
    bool ran_full = false;
    // The frames finish_body_frames wants (TCP, sensor link) at the step's last q: lane k of a licensed walk evaluates the kinematics at
    // q + k dt des anyway, so when the walk ends the step its lane `adv` holds them (the same additions in the same order as q's own) and
    // the step's third forward-kinematics pass (~10 k of its ~120 k cycles) is not run.  frames_lane < 0: no such lane (a full tick came last).
    // Round 5, 1024 envs: 76.5 -> 62 us per launch together with trig_init_lanes and n = gridDim.x.  Tried with it and dropped: the robot's and
    // the env kind's constants staged in LDS (the controller's share fell from 17 k to 6 k cycles - its ~100 scalar loads each miss a scalar
    // cache the launch starts cold - but 260 spilled VGPRs and 900 B of scratch made the kernel slower), and touching every line of the two
    // structs up front with scalar loads (+10 us: at most 15 are in flight).
    int frames_lane = -1, tmpl_lane = -1;
    T tmpl_q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) tmpl_q[i] = inline_reset ? (T)st.reset_tmpl[i] : T(0);
    V3<T> ptcp_l = mk<T>(0, 0, 0), pb_l = mk<T>(0, 0, 0);
    M3<T> Rtcp_l, Rb_l;
#pragma unroll
    for (int e = 0; e < 9; ++e) { Rtcp_l.m[e] = T(0); Rb_l.m[e] = T(0); }
    const T kdamp = c.dt * (m.joint_damp + T(4) * (m.lin_damp + m.ang_damp) * m.trace_bound) * T(3);
    int t = 0;
    while (t < c.action_repeat) {
        // Licensed ticks: the arm's path is known beforehand - the motors hold qd = des, so the k-th tick from here starts at q + k dt des -
        // and only the pole's update is sequential.  Lane k evaluates that tick's kinematics (exact sines / cosines, pivot A and its velocity),
        // all remaining ticks at once; the wavefront then walks them with the body half of the analytic tick alone (body_tick_pivot), fetching
        // pivot A from lane k.  Same arithmetic per tick as body_tick_analytic.  A tick whose a-priori test fails ends the walk: it is solved in
        // full below (which renews or withdraws the licence) and the walk resumes behind it.
        if (verified > 0 && c.solver_iters >= 0 && m.vel_gain == T(1) && c.action_repeat <= 64) {
            const int left = c.action_repeat - t;
            T dq[N], qt[N], dvw = T(0), v2 = T(0), v2d = T(0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                dq[i] = c.dt * qd_des[i]; qt[i] = q[i];
                dvw += m.diag_sqrt[i] * tabs(qd_des[i] - qd[i]); v2 += qd[i] * qd[i]; v2d += qd_des[i] * qd_des[i];
            }
            for (int r = 0; r < left; ++r)
                if (r < lane) {
#pragma unroll
                    for (int i = 0; i < N; ++i) qt[i] += dq[i];
                }
            if (inline_reset && left < 63) {              // the walk's last lane is idle: it evaluates the kinematics at the reset template's q, for
                tmpl_lane = 63;                           // a finished env's reset at the end of this launch (which then runs no forward kinematics)
#pragma unroll
                for (int i = 0; i < N; ++i) qt[i] = lane == 63 ? tmpl_q[i] : qt[i];
            }
            V3<T> pa_l, va_l;
            {
                Kin<T, TOPO> kin;
                forward_kinematics<T, TOPO>(m, qt, kin);
                pivot_state<T, TOPO>(kin, c.body, qd_des, pa_l, va_l);
                link_frame<T, TOPO>(kin, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp_l, Rtcp_l);
                link_frame<T, TOPO>(kin, m.sensor_link, m.sensor_pos, m.sensor_rot, pb_l, Rb_l);
            }
            int adv = 0;
            for (int k = 0; k < left; ++k) {
                const int kl = __builtin_amdgcn_readfirstlane(k);
                const V3<T> pa = mk(bcast(pa_l.x, kl), bcast(pa_l.y, kl), bcast(pa_l.z, kl)), va = mk(bcast(va_l.x, kl), bcast(va_l.y, kl), bcast(va_l.z, kl));
                const T lam_arm = k == 0 ? m.diag_sqrt_max * dvw + kdamp * tsqrt_fast(v2) : m.diag_sqrt_max * T(0) + kdamp * tsqrt_fast(v2d);
                if (!body_tick_pivot<T>(lam_arm, m.max_force, c.dt, grav, b, c.body, pivot_b, fext, pext, pending && t + k == 0, pa, va)) break;
                ++adv;
            }
            if (adv > 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) qd[i] = qd_des[i];
                for (int r = 0; r < adv; ++r) {
#pragma unroll
                    for (int i = 0; i < N; ++i) q[i] += dq[i];
                }
                verified -= adv;
                t += adv;
                frames_lane = adv < 64 ? adv : -1;          // (action_repeat == 64: the walk has no lane standing at its last q)
                continue;
            }
        }
        {
            if (!staged) { stage_link_constants<T, TOPO>(mp, L, lane); staged = true; }
            frames_lane = -1;
            trig_init_lanes<T, N>(q, trig, lane);
            __syncthreads();
            if (w0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool in = i < N;
                    L[kLQ + i] = in ? q[in ? i : 0] : T(0); L[kLQd + i] = in ? qd[in ? i : 0] : T(0);
                    L[kLTrigS + i] = in ? trig.s[in ? i : 0] : T(0); L[kLTrigC + i] = in ? trig.c[in ? i : 0] : T(1);
                    L[kLQDes + i] = T(0); L[kLQdDes + i] = in ? qd_des[in ? i : 0] : T(0);
                }
                L[kLBody + 0] = b.pos.x; L[kLBody + 1] = b.pos.y; L[kLBody + 2] = b.pos.z;
#pragma unroll
                for (int e = 0; e < 9; ++e) L[kLBody + 3 + e] = b.R.m[e];
                L[kLBody + 12] = b.v.x; L[kLBody + 13] = b.v.y; L[kLBody + 14] = b.v.z;
                L[kLBody + 15] = b.w.x; L[kLBody + 16] = b.w.y; L[kLBody + 17] = b.w.z;
            }
            TG_PHASE_FENCE()
            verified = sim_tick_p2p_wave<T, TOPO, kMotorVelocity>(m, c.body, L, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, pivot_b, fext, pext,
                                                                  pending && t == 0, lane, &sweeps);
            ran_full = true;
#pragma unroll
            for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; }
            b.pos = mk(L[kLBody + 0], L[kLBody + 1], L[kLBody + 2]);
#pragma unroll
            for (int e = 0; e < 9; ++e) b.R.m[e] = L[kLBody + 3 + e];
            b.v = mk(L[kLBody + 12], L[kLBody + 13], L[kLBody + 14]);
            b.w = mk(L[kLBody + 15], L[kLBody + 16], L[kLBody + 17]);
            ++t;
        }
    }
    if (w0) {
        st.step_count[env] = step_count;
        st.licence[env] = ran_full ? (verified > 0 ? 8 : 0) : lic - 1;
        if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
        st.ext_pending[env] = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = (double)qd_des[i]; }
        store_body<T>(st, n, env, b);
    }
    {
        V3<T> ptcp, pb; M3<T> Rtcp, Rb;
        bool env_done;
        if (frames_lane >= 0) {
            const int fl = __builtin_amdgcn_readfirstlane(frames_lane);
            ptcp = mk(bcast(ptcp_l.x, fl), bcast(ptcp_l.y, fl), bcast(ptcp_l.z, fl));
            pb = mk(bcast(pb_l.x, fl), bcast(pb_l.y, fl), bcast(pb_l.z, fl));
#pragma unroll
            for (int e = 0; e < 9; ++e) { Rtcp.m[e] = bcast(Rtcp_l.m[e], fl); Rb.m[e] = bcast(Rb_l.m[e], fl); }
        } else {
            trig_init_lanes<T, N>(q, trig, lane);
            Kin<T, TOPO> kin;
            forward_kinematics<T, TOPO, true>(m, q, kin, &trig);
            link_frame<T, TOPO>(kin, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
            link_frame<T, TOPO>(kin, m.sensor_link, m.sensor_pos, m.sensor_rot, pb, Rb);
        }
        env_done = finish_body_frames<T, TOPO>(m, c, st, env, ptcp, Rtcp, pb, Rb, b, embed, step_count, true);
        // Auto-reset in the step's own launch (round 5): with the reset template valid (the host knows: tg_ctx::tmpl_ready) a finished env's reset is
        // the template-only form - four to six draws, a teleport, and the frames at the template's q, which the walk's idle last lane evaluated
        // on the side - and its launch of its own (k_reset_body: 16 wavefronts that mostly find nothing to do) cost the step 16 us + a graph gap;
        // here it is the tail of a finished env's wavefront (with ~4 of 1024 envs finishing per step the launch nearly always has one: 5.7 us
        // with the reset's own forward kinematics).  Every lane runs it alike, as with finish_body_frames above (the same loads, the same stores).
        if (inline_reset && env_done) {
            if (tmpl_lane >= 0) {
                LinkFrames<T> fr;
                fr.ptcp = mk(bcast(ptcp_l.x, 63), bcast(ptcp_l.y, 63), bcast(ptcp_l.z, 63));
                fr.pb = mk(bcast(pb_l.x, 63), bcast(pb_l.y, 63), bcast(pb_l.z, 63));
#pragma unroll
                for (int e = 0; e < 9; ++e) { fr.Rtcp.m[e] = bcast(Rtcp_l.m[e], 63); fr.Rb.m[e] = bcast(Rb_l.m[e], 63); }
                reset_body_env<T, TOPO, false, true>(m, c, st, env, &fr);
            } else {
                reset_body_env<T, TOPO, false, true>(m, c, st, env);
            }
        }
    }
    if (st.draw != nullptr && w0) draw_ticket_resolve(st, ticket);
}

// ---- contact-free arms (edge_follow, surface_follow): the full tick on one wavefront -------------------------------------------------------
// The lane-per-env k_step runs a full tick - articulated-body dynamics + up to 150 Gauss-Seidel sweeps over the N joint motors - as one
// scalar program per lane: 1024 envs are 16 wavefronts on a chip with 1024 SIMDs.  Here the env has its own wavefront: tick_dynamics_lanes
// for the arm, one motor row per lane (lane i < N), a whole forward or reverse Gauss-Seidel pass as ONE linear map of the row lanes'
// residuals (x -= sum_m C_m x_m, C built once per tick; no motor row has a limit in reach, watched on lanes 48.. with the literal clamped
// iteration as the fallback), accumulator lanes 32 + k for the velocity change.  Same row order, same convergence exit (every 8 sweeps:
// all residuals below 2^-56 of the largest start value) and same licence rule as sim_tick; returns the new `verified` (24: converged
// within 80 % of the sweep budget).  State in / out through LDS (q, qd, sines / cosines).
// MEASURED (MI355X, 1024 envs): slower than the lane mapping - UR5 literal k_step 0.886 ms against 0.512 ms, MG400 1.05 against ~0.95 ms: a
// wave64 instruction costs its 4 issue cycles whether 6 or 64 lanes do useful work, so a sweep is ~190 cycles for one env here and 144 for
// 64 envs there, and 1024 mostly idle wavefronts on 1024 SIMDs are no faster than 16 full ones.  Selected by TG_CONTACT_MAP_WAVE only.
template <typename T, int TOPO, int MOTOR>
__device__ __forceinline__ int sim_tick_arm_wave(const DevRobot<T>& m, lds_ptr<T> L, T kp, T kd, T max_force, T dt, int iters, int lane_in,
                                                 int* sweep_acc = nullptr /* threshold mode: += the sweeps this tick ran */) {
    constexpr int N = Topo<TOPO>::N;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
#if TG_WAVE_TIMING
    unsigned long long t_prev_ = __builtin_readcyclecounter();
    tick_dynamics_lanes<T, TOPO>(&m, L, m.tcp_link, dt, lane, t_prev_);
#else
    tick_dynamics_lanes<T, TOPO>(&m, L, m.tcp_link, dt, lane);
#endif
    TG_PHASE_FENCE()
    // ---- this lane's row: J = e_lane, W = Minv e_lane (column `lane` of Minv), A = Minv[lane][lane]
    const bool motor_lane = lane < N;
    const int col = motor_lane ? lane : 0;
    T Warm[N], rm = T(0);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        Warm[k] = motor_lane ? L[kLMinv + 8 * k + col] : T(0);
        const T vk = L[kLV + k];
        const T pos_term = (MOTOR == kMotorPosition) ? kp * (L[kLQDes + k] - L[kLQ + k]) / dt : T(0);
        const T des = pos_term + vk + kd * (L[kLQdDes + k] - vk);
        rm = lane == k ? des - vk : rm;
    }
    T A = T(0);
#pragma unroll
    for (int k = 0; k < N; ++k) A = lane == k ? Warm[k] : A;
    const bool active = motor_lane && MOTOR != kMotorOff;
    const T jdi = active ? T(1) / A : T(0);
    // coefficient rows: row lanes G[i] = W_mine[i] / A (own entry 1: the lane carries its residual), lanes 32 + k accumulate du_k = sum_i
    // Minv[k][i] lambda_i, lanes 48 + j the impulse of motor j
    T G[N];
    {
        const int ku = lane - 32;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const T wv = L[kLMinv + 8 * (ku >= 0 && ku < N ? ku : 0) + i];
            const T g_row = lane == i ? T(1) : Warm[i] * jdi;
            G[i] = (ku >= 0 && ku < N) ? (MOTOR != kMotorOff ? -wv : T(0))
                                       : ((lane >= 48 && lane < 48 + N) ? (i == lane - 48 ? T(-1) : T(0)) : (motor_lane ? g_row : T(0)));
        }
    }
    // ---- one Gauss-Seidel pass over the N rows as a linear map of the row lanes' residuals, forward and reverse order (as in
    //      sim_tick_contact_wave's motor pass)
    T Cf[N], Cr[N];
    {
        __syncthreads();
        if (motor_lane) {
#pragma unroll
            for (int k = 0; k < N; ++k) L[kLW + lane * 16 + k] = G[k];     // L[i][k] = G_k on the lane of row i
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < N; ++mm) {
            T t[N];
            t[mm] = T(1);
            T c = G[mm];
#pragma unroll
            for (int i = mm + 1; i < N; ++i) {
                T acc = T(0);
#pragma unroll
                for (int k = mm; k < i; ++k) acc = __builtin_fma(L[kLW + i * 16 + k], t[k], acc);
                t[i] = -acc;
                c = __builtin_fma(G[i], t[i], c);
            }
            Cf[mm] = c;
        }
#pragma unroll
        for (int mm = 0; mm < N; ++mm) {
            T t[N];
            t[mm] = T(1);
            T c = G[mm];
#pragma unroll
            for (int i = mm - 1; i >= 0; --i) {
                T acc = T(0);
#pragma unroll
                for (int k = i + 1; k <= mm; ++k) acc = __builtin_fma(L[kLW + i * 16 + k], t[k], acc);
                t[i] = -acc;
                c = __builtin_fma(G[i], t[i], c);
            }
            Cr[mm] = c;
        }
    }
    const T x0 = active ? rm * jdi : T(0);
    T thr = tabs(x0);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) thr = tmax(thr, __shfl_xor(thr, o));      // rows live in lanes 0..7
    thr = bcast(thr, 0);
    thr = iters < 0 ? T(-1) : thr * (sizeof(T) == 8 ? T(1.3877787807814457e-17) : T(7.450580596923828e-09));
    const int n_it = iters < 0 ? -iters : iters;
    const T lim = max_force * dt;
    const bool watch_lane = lane >= 48 && lane < 48 + N;
    T x = x0, wmax = T(0);
    int conv_sweeps = -1;
    const bool thr_mode = uniform_true(m.res_thr > T(0));
    if (thr_mode) {                                       // threshold mode: Bullet's exit after every sweep (oracle mb_step), see sim_tick_p2p_wave
        T Ad[N];
#pragma unroll
        for (int i = 0; i < N; ++i) Ad[i] = bcast(A, i);
        T lam = T(0);
        int ran = 0;
        for (int it = 0; it < n_it; ++it) {
            T res = T(0);
#pragma unroll
            for (int kk = 0; kk < N; ++kk) {
                const int i = (it & 1) ? kk : N - 1 - kk;
                const T dd = bcast(vmin(vmax(x, -lim - lam), lim - lam), i);
                lam = lane == i ? lam + dd : lam;
                x = __builtin_fma(-G[i], dd, x);
                const T dvel = dd * Ad[i];
                res = tmax(res, dvel * dvel);
            }
            ++ran;
            if (uniform_true(res <= m.res_thr)) break;
        }
        if (sweep_acc != nullptr) *sweep_acc += ran;
    } else
    for (int it = 0; it < n_it; ++it) {
        if ((it & 7) == 0 && it > 0) {
            T mx = motor_lane ? tabs(x) : T(0);
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) mx = tmax(mx, __shfl_xor(mx, o));
            if (uniform_true(bcast(mx, 0) <= thr)) { conv_sweeps = it; break; }
        }
        T xm[N];
#pragma unroll
        for (int k = 0; k < N; ++k) xm[k] = bcast(x, k);
        T da = T(0), db = T(0);
        if (it & 1) {
#pragma unroll
            for (int k = 0; k < N; k += 2) { da = __builtin_fma(Cf[k], xm[k], da); if (k + 1 < N) db = __builtin_fma(Cf[k + 1], xm[k + 1], db); }
        } else {
#pragma unroll
            for (int k = 0; k < N; k += 2) { da = __builtin_fma(Cr[k], xm[k], da); if (k + 1 < N) db = __builtin_fma(Cr[k + 1], xm[k + 1], db); }
        }
        x -= da + db;
        wmax = vmax_abs(wmax, x);
    }
    if (!thr_mode && uniform_true(watch_lane && wmax > lim)) {
        // a motor impulse reached its bound: the literal clamped iteration, row by row
        x = x0;
        T lam = T(0);
        conv_sweeps = -1;
        for (int it = 0; it < n_it; ++it) {
#pragma unroll
            for (int kk = 0; kk < N; ++kk) {
                const int i = (it & 1) ? kk : N - 1 - kk;
                const T dd = bcast(vmin(vmax(x, -lim - lam), lim - lam), i);
                lam = lane == i ? lam + dd : lam;
                x = __builtin_fma(-G[i], dd, x);
            }
        }
    }
    // ---- integrate
    T du[N];
#pragma unroll
    for (int k = 0; k < N; ++k) du[k] = bcast(x, 32 + k);
    {
        T q[N], qd[N];
        JointTrig<T, N> trig;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            qd[i] = L[kLV + i] + du[i];
            q[i] = L[kLQ + i] + dt * qd[i];
        }
        trig_init<T, N>(q, trig);                         // every full tick re-anchors the carried sines / cosines exactly
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) { L[kLQ + i] = q[i]; L[kLQd + i] = qd[i]; L[kLTrigS + i] = trig.s[i]; L[kLTrigC + i] = trig.c[i]; }
        }
    }
    TG_PHASE_FENCE()
    return (iters > 0 && conv_sweeps > 0 && 5 * conv_sweeps <= 4 * iters) ? 24 : 0;
}

// BaseTactileEnv.step of the contact-free arm tasks (TCP_velocity_control), one wavefront per env: prologue and epilogue as k_step
// (tg_kernels.hpp; evaluated by every lane alike), every one of the action_repeat ticks a full tick on the wave mapping.
template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_step_arm_wave(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                      const float* __restrict__ actions) {
    KtScope kt_scope_(st.kt);
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double wave_lds_raw[];
    const lds_ptr<T> L = (lds_ptr<T>)wave_lds_raw;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x, lane = threadIdx.x;
    const int n = c.num_envs;
    const bool w0 = lane == 0;
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = (T)st.q[i * n + env]; qd[i] = (T)st.qd[i * n + env]; }
    T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    encode_arm_actions<T>(c, st, env, actions + (size_t)env * c.act_dim, enc);
    T vels[6];
    scale_actions<T>(c, enc, vels);
    const int step_count = st.step_count[env] + 1;
    JointTrig<T, N> trig;
    trig_init<T, N>(q, trig);
    T qd_des[N];
    tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des, &trig);
    stage_link_constants<T, TOPO>(mp, L, lane);
    __syncthreads();
    if (w0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool in = i < N;
            L[kLQ + i] = in ? q[in ? i : 0] : T(0); L[kLQd + i] = in ? qd[in ? i : 0] : T(0);
            L[kLTrigS + i] = in ? trig.s[in ? i : 0] : T(0); L[kLTrigC + i] = in ? trig.c[in ? i : 0] : T(1);
            L[kLQDes + i] = T(0); L[kLQdDes + i] = in ? qd_des[in ? i : 0] : T(0);
        }
    }
    TG_PHASE_FENCE()
    int verified = 0, sweeps = 0;
    for (int t = 0; t < c.action_repeat; ++t)
        verified = sim_tick_arm_wave<T, TOPO, kMotorVelocity>(m, L, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, lane, &sweeps);
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; trig.s[i] = L[kLTrigS + i]; trig.c[i] = L[kLTrigC + i]; }
    if (w0) {
        st.step_count[env] = step_count;
        st.licence[env] = 0;                       // this kernel solves every tick in full: nothing is carried over for k_step's shortcut
        if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = (double)qd_des[i];
            st.trig_sc[i * n + env] = (double)trig.s[i]; st.trig_sc[(8 + i) * n + env] = (double)trig.c[i];
        }
    }
    (void)verified;
    finish_env<T, TOPO>(m, c, st, env, q, (T)st.edge_ang[env], step_count, true, &trig, true);
}

// env.reset() of object_push (SHAPE 0) / object_roll (SHAPE 1) for the envs flagged in `mask`, one wavefront per resetting env (the
// lane-per-env k_reset_push / k_reset_roll run every resetting env's blocking move at the pace of a 64-env wavefront: 0.8 ms per launch with
// a reset in object_roll, where some env finishes on nearly every step).  Same sequence as those kernels: episode draws, rest pose,
// inverse kinematics of the start pose (wave-uniform, every lane the same), then the blocking move - each tick sim_tick_contact_wave with
// position motors (max force 100 000 as robot.py:188-260 / base_robot_arm.py pass it) against the PREVIOUS episode's object - then the object
// is put back and the first observation's transforms are written.
template <typename T, int TOPO, int SHAPE, bool CONE, int NT = 1>
__global__ __launch_bounds__(64) void k_reset_contact_wave(const DevRobot<T>* __restrict__ mp, const EnvConst<T>* __restrict__ cp, State st,
                                                           const uint8_t* __restrict__ mask, int optimistic) {
    // (no KtScope: its one more live scalar pair took this kernel from 0 to 204 B of scratch per lane - 15 MB of spill traffic per launch by the
    //  PMC - at a duration in milliseconds, where a HIP event pair's 5 us do not matter)
    constexpr int N = Topo<TOPO>::N;
    extern __shared__ double wave_lds_raw[];
    const lds_ptr<T> L = (lds_ptr<T>)wave_lds_raw;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x, lane = threadIdx.x;
    const int n = c.num_envs;
    if (mask != nullptr && mask[env] == 0) return;
    const bool w0 = lane == 0;                                  // state goes back to HBM once
    // ---- episode draws (the order of the draws is the lane kernels')
    uint64_t rs = st.rng[env];
    const double old_mr = st.obj_mass[env];                     // push: the cube's mass; roll: the marble's radius
    double new_mr = old_mr, ang = 0.0, scaling = 1.0, embed = st.embed[env], ix = 0.0, iy = 0.0;
    if constexpr (SHAPE == 0) {
        ang = c.rand_init_orn ? rng_uniform(rs, -c.init_orn_range, c.init_orn_range) : 0.0;
        new_mr = c.rand_obj_mass ? rng_uniform(rs, c.mass_lo, c.mass_hi) : old_mr;
        if (c.traj_type == TG_TRAJ_SIMPLEX) { const int64_t seed = (int64_t)rng_uniform(rs, 0.0, 1.0e8); if (w0) st.noise_seed[env] = seed; }
        else {
            const double ta = rng_uniform(rs, -c.traj_ang_range, c.traj_ang_range);
            double ys[TG_MAX_TRAJ_POINTS];
            for (int i = 0; i < c.traj_n; ++i) {
                const double dist = (double)i * c.traj_spacing;
                ys[i] = dist * sin(ta);
                if (w0) { st.traj[(0 * TG_MAX_TRAJ_POINTS + i) * n + env] = c.traj_init_offset + dist * cos(ta); st.traj[(1 * TG_MAX_TRAJ_POINTS + i) * n + env] = ys[i]; }
            }
            for (int i = 0; i < c.traj_n; ++i) {
                double g;
                if (i == 0) g = (ys[1] - ys[0]) / c.traj_spacing;
                else if (i == c.traj_n - 1) g = (ys[i] - ys[i - 1]) / c.traj_spacing;
                else g = (ys[i + 1] - ys[i - 1]) / (2.0 * c.traj_spacing);
                if (w0) st.traj[(2 * TG_MAX_TRAJ_POINTS + i) * n + env] = g;
            }
        }
    } else {
        scaling = c.roll_rand_size ? rng_uniform(rs, 1.0, 2.0) : 1.0;
        new_mr = c.roll_radius * scaling;
        if (c.roll_rand_embed) embed = rng_uniform(rs, c.embed_lo, c.embed_hi);
        if (c.roll_rand_init_pos) { ix = rng_uniform(rs, -c.roll_init_range, c.roll_init_range); iy = rng_uniform(rs, -c.roll_init_range, c.roll_init_range); }
        const double gang = rng_uniform(rs, -3.141592653589793, 3.141592653589793);
        const double gdist = rng_uniform(rs, c.roll_goal_lo, c.roll_goal_hi);
        if (w0) { st.embed[env] = embed; st.goal[0 * n + env] = gdist * cos(gang); st.goal[1 * n + env] = gdist * sin(gang); st.goal[2 * n + env] = 0.0; }
    }
    if (w0) { st.rng[env] = rs; st.step_count[env] = 0; }
    if constexpr (SHAPE == 0) { if (w0) st.goal_id[env] = c.reset_goal_id; }
    // ---- rest pose, start-pose inverse kinematics
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = m.rest_q[i]; qd[i] = T(0); }
    const FreeBody<T> b0 = load_body<T>(st, n, env);
    int used = 0, ccode = 0;
    constexpr int kTmplPath = 64;
    const bool tmpl_can = SHAPE == 0 && optimistic != 0 && NT == 1 && !(m.res_thr > T(0)) && st.reset_tmpl != nullptr;
    bool tmpl_valid = false;
    if (tmpl_can) tmpl_valid = st.reset_tmpl[2 * N + 1] != 0.0;
    bool took_template = false;
    if (uniform_true(tmpl_valid)) {
        const int npts = (int)st.reset_tmpl[2 * N + 2];
        const T bs_r = (T)st.reset_tmpl[2 * N + 3];
        bool clear = true;
        for (int k2 = 0; k2 < npts; ++k2) {
            const V3<T> cw = mk((T)st.reset_tmpl[2 * N + 4 + 3 * k2], (T)st.reset_tmpl[2 * N + 5 + 3 * k2], (T)st.reset_tmpl[2 * N + 6 + 3 * k2]);
            const V3<T> loc = mulT(b0.R, cw - b0.pos);
            const T ox = tmax(tabs(loc.x) - c.push.half[0], T(0)), oy = tmax(tabs(loc.y) - c.push.half[1], T(0)), oz = tmax(tabs(loc.z) - c.push.half[2], T(0));
            clear = clear && (tsqrt(ox * ox + oy * oy + oz * oz) - bs_r > T(0.004));
        }
        if (uniform_true(clear)) {
#pragma unroll
            for (int i = 0; i < N; ++i) { q[i] = (T)st.reset_tmpl[i]; qd[i] = (T)st.reset_tmpl[N + i]; }
            used = (int)st.reset_tmpl[2 * N];
            {   // the cube's table contacts as sim_tick_contact_wave selects them: vertices within the breaking distance, at most 4 (the deepest)
                T vz[8]; int keep = 0, cnt = 0;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const T lx = (cc & 4) ? c.push.half[0] : -c.push.half[0], ly = (cc & 2) ? c.push.half[1] : -c.push.half[1], lz = (cc & 1) ? c.push.half[2] : -c.push.half[2];
                    vz[cc] = (b0.pos.z + (b0.R.m[6] * lx + b0.R.m[7] * ly + b0.R.m[8] * lz)) - c.push.table_z;
                    if (vz[cc] <= c.push.breaking) { keep |= 1 << cc; ++cnt; }
                }
                while (cnt > 4) {
                    int worst = -1; T wz = T(0);
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc)
                        if (((keep >> cc) & 1) && (worst < 0 || vz[cc] >= wz)) { worst = cc; wz = vz[cc]; }
                    keep &= ~(1 << worst); --cnt;
                }
                ccode = keep;
            }
            took_template = true;
        }
    }
    if (!took_template) {
    const V3<T> tpos = SHAPE == 0 ? load_v3(c.work_pos) : mk(c.work_pos[0], c.work_pos[1], (T)(2.0 * new_mr - embed));   // work-frame origin, rpy 0
    T trpy[3];
    euler_from_quat(quat_mul(c.work_q, quat_from_euler(T(0), T(0), T(0))), trpy[0], trpy[1], trpy[2]);
    const Q4<T> tq = quat_from_euler(trpy[0], trpy[1], trpy[2]);
    const M3<T> Rt = mat_from_quat(tq);
    T qik[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qik[i] = q[i];
    inverse_kinematics<T, TOPO>(m, tpos, Rt, qik, 100, T(1e-8));
    if (N == 8) { qik[N - 3] = qik[1]; qik[N - 2] = -qik[1]; qik[N - 1] = qik[1] + qik[2]; }
    // ---- env state to LDS (the previous episode's object where it was left)
    if constexpr (SHAPE == 0) {
        const T* tipv = (const T*)st.tip_verts;
        const int nw = 3 * c.push.n_tip;
        for (int w = lane; w < nw; w += 64) L[kLHull + w] = tipv[w];
    }
    const int xbase = kLHull + 3 * c.push.n_tip;
    if constexpr (NT == 4) {             // the blocking move runs against the previous episode's cube: its manifold is still there
        if (lane < 37) L[xbase + kXMani + (lane < 36 ? lane : narrow::kMcount)] = (T)st.mani[(size_t)lane * n + env];
    }
    stage_link_constants<T, TOPO>(mp, L, lane);
    // The OPTIMISTIC blocking move (round 6).  Robot.reset (robot.py:114-125) teleports the arm to its rest pose and drives it to the start pose
    // with the PREVIOUS episode's object still in the world; reset_object then teleports that object (base_object_env.py:146-173), so whatever
    // the object did during the move is discarded - unless the tip touched it.  While the tip stays clear of the object the arm's motor rows and
    // the object's table rows do not interact (their Delassus entries are exact zeros), i.e. the arm moves exactly as in an arm-only tick with the
    // same 150 sweeps: those ticks run sim_tick_arm_wave (no contact generation, no friction chains: ~1/3 of a contact tick) with the object
    // frozen where the finished episode left it.  "Clear" = the bounding sphere of the tip's collision shape further than 4 mm from the object
    // (margins 1.1 mm + breaking 0.1 mm + what a decelerating object and the tip travel in a tick), tested before every tick; the first tick is a
    // full contact tick (it also yields the contact code a reset leaves behind).  A move in which the test fails once is run AGAIN from the
    // rest pose with full contact ticks throughout: the results are those of the literal move either way (arm to rounding, reset tick counts
    // exactly: tests/test_gpu_parity.py, tests/test_gpu_config_scale.py).  Not in threshold mode (the loop's exit there couples all rows).
    V3<T> bs_c = mk<T>(0, 0, 0);
    T bs_r = T(0);
    bool literal = !(optimistic != 0 && NT == 1 && !(m.res_thr > T(0)));
    if (!literal) {
        if constexpr (SHAPE == 0) {
            T lo[3] = {T(1e30), T(1e30), T(1e30)}, hi[3] = {T(-1e30), T(-1e30), T(-1e30)};
            __syncthreads();
            for (int v = lane; v < c.push.n_tip; v += 64)
#pragma unroll
                for (int x = 0; x < 3; ++x) { const T w = L[kLHull + 3 * v + x]; lo[x] = tmin(lo[x], w); hi[x] = tmax(hi[x], w); }
#pragma unroll
            for (int x = 0; x < 3; ++x)
                for (int o = 1; o < 64; o <<= 1) { lo[x] = tmin(lo[x], __shfl_xor(lo[x], o)); hi[x] = tmax(hi[x], __shfl_xor(hi[x], o)); }
            bs_c = mk(T(0.5) * (lo[0] + hi[0]), T(0.5) * (lo[1] + hi[1]), T(0.5) * (lo[2] + hi[2]));
            T r2 = T(0);
            for (int v = lane; v < c.push.n_tip; v += 64) {
                const V3<T> d = mk(L[kLHull + 3 * v], L[kLHull + 3 * v + 1], L[kLHull + 3 * v + 2]) - bs_c;
                r2 = tmax(r2, dot(d, d));
            }
            for (int o = 1; o < 64; o <<= 1) r2 = tmax(r2, __shfl_xor(r2, o));
            bs_r = tsqrt(r2);
        } else {
            bs_c = load_v3(c.push.cyl_pos);
            bs_r = tsqrt(c.push.cyl_r * c.push.cyl_r + c.push.cyl_hl * c.push.cyl_hl);
        }
    }
    T cv = T(0.001);
    // The reset TEMPLATE (object_push; round 6).  With the tip clear of the object the whole move is a function of constants - rest pose, start pose,
    // gains: nothing of the episode, nothing of the env - so its result (q, qd, tick count) and the tip sphere's path are kept from the first
    // optimistic move that ran through (State.reset_tmpl: [0, N) q, [N, 2N) qd, [2N] ticks, [2N + 1] valid, [2N + 2] path points, [2N + 4 + 3k] the
    // sphere's centre before tick k and, last, at the final pose).  A later reset whose object is further than the 4 mm from EVERY point of that path
    // takes the template: no tick at all (the same bits the optimistic move would produce - it is deterministic); otherwise it moves as above.
    // object_roll's start pose depends on the episode's marble: no template there.  The contact code a reset leaves behind is the object's
    // table contacts where the episode left it (the literal move reports those of its last tick: the same set for an object at rest).
    const bool record = tmpl_can && !tmpl_valid;           // this move may become the template: leave the sphere's path behind
    int n_path = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
    bool violated = false;
    n_path = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = m.rest_q[i]; qd[i] = T(0); }
    {
        JointTrig<T, N> trig;
        trig_init<T, N>(q, trig);
        __syncthreads();
        if (w0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool in = i < N;
                L[kLQ + i] = in ? q[in ? i : 0] : T(0); L[kLQd + i] = T(0);
                L[kLTrigS + i] = in ? trig.s[in ? i : 0] : T(0); L[kLTrigC + i] = in ? trig.c[in ? i : 0] : T(1);
                L[kLQDes + i] = in ? q[in ? i : 0] : T(0); L[kLQdDes + i] = T(0);
            }
            L[kLBody + 0] = b0.pos.x; L[kLBody + 1] = b0.pos.y; L[kLBody + 2] = b0.pos.z;
#pragma unroll
            for (int e = 0; e < 9; ++e) L[kLBody + 3 + e] = b0.R.m[e];
            L[kLBody + 12] = b0.v.x; L[kLBody + 13] = b0.v.y; L[kLBody + 14] = b0.v.z;
            L[kLBody + 15] = b0.w.x; L[kLBody + 16] = b0.w.y; L[kLBody + 17] = b0.w.z;
        }
    }
    TG_PHASE_FENCE()
    // ---- blocking_move(max_steps = 1000, constant_vel = 0.001), robot.py:188-260
    cv = T(0.001);
    used = 0; ccode = 0;
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; }
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(m, q, k);
        V3<T> p; M3<T> R;
        link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, p, R);
        const Q4<T> cq = quat_from_mat(R);
        T diff[N], nrm2 = T(0), total_v = T(0);
        bool all_small = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diff[i] = qik[i] - q[i];
            nrm2 += diff[i] * diff[i];
            all_small = all_small && (tabs(diff[i]) < cv);
            total_v += tabs(qd[i]);
        }
        const T nrm = tsqrt(nrm2);
        __syncthreads();
        if (w0) {
#pragma unroll
            for (int i = 0; i < N; ++i) L[kLQDes + i] = q[i] + ((nrm > T(0)) ? diff[i] / nrm : T(0)) * cv;
        }
        if (all_small) cv = cv / T(2);
        bool arm_only = false;
        if (!literal) {                  // is the tip clear of the (frozen) object before this tick?  (the first tick is a contact tick either way)
            V3<T> ol; M3<T> Rl;
            const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}, z3[3] = {T(0), T(0), T(0)};
            link_frame<T, TOPO>(k, c.push.tip_link, z3, ident, ol, Rl);
            const V3<T> cw = ol + mul(Rl, bs_c);
            if (record && w0 && n_path < kTmplPath - 1) {
                st.reset_tmpl[2 * N + 4 + 3 * n_path] = (double)cw.x; st.reset_tmpl[2 * N + 5 + 3 * n_path] = (double)cw.y; st.reset_tmpl[2 * N + 6 + 3 * n_path] = (double)cw.z;
            }
            ++n_path;
            const V3<T> bp = mk(L[kLBody + 0], L[kLBody + 1], L[kLBody + 2]);
            T dist;
            if constexpr (SHAPE == 1) dist = norm(cw - bp) - (T)old_mr;
            else {
                M3<T> Rb;
#pragma unroll
                for (int e = 0; e < 9; ++e) Rb.m[e] = L[kLBody + 3 + e];
                const V3<T> loc = mulT(Rb, cw - bp);
                const T ox = tmax(tabs(loc.x) - c.push.half[0], T(0)), oy = tmax(tabs(loc.y) - c.push.half[1], T(0)), oz = tmax(tabs(loc.z) - c.push.half[2], T(0));
                dist = tsqrt(ox * ox + oy * oy + oz * oz);
            }
            arm_only = dist - bs_r > T(0.004);
            if (it == 0) arm_only = false;
            else if (!uniform_true(arm_only)) { violated = true; break; }
        }
        TG_PHASE_FENCE()
        if (uniform_true(arm_only)) (void)sim_tick_arm_wave<T, TOPO, kMotorPosition>(m, L, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, lane);
        else ccode = sim_tick_contact_wave<T, TOPO, kMotorPosition, SHAPE, CONE, NT>(m, c.push, L, m.pos_gain, m.vel_gain, T(100000), c.dt, c.solver_iters, (T)old_mr, lane, xbase);
        ++used;
        const T pos_err = tabs(tpos.x - p.x) + tabs(tpos.y - p.y) + tabs(tpos.z - p.z);
        const T ip = tq.x * cq.x + tq.y * cq.y + tq.z * cq.z + tq.w * cq.w;
        T ca = T(2) * ip * ip - T(1);
        ca = ca > T(1) ? T(1) : (ca < T(-1) ? T(-1) : ca);
        if (uniform_true(pos_err < T(2e-4) && tacos(ca) < T(1e-3) && total_v < T(0.1))) break;
    }
    if (!violated) {
        if (record && !literal && n_path < kTmplPath - 1) {
            // this move ran through with the tip clear: it is the template.  Every env that gets here writes the same values (the move is a function
            // of constants); the flag goes last, behind a fence, and is read at kernel start only (a later launch sees all of it).
#pragma unroll
            for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; }
            Kin<T, TOPO> kf;
            forward_kinematics<T, TOPO>(m, q, kf);
            V3<T> ol; M3<T> Rl;
            const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}, z3[3] = {T(0), T(0), T(0)};
            link_frame<T, TOPO>(kf, c.push.tip_link, z3, ident, ol, Rl);
            const V3<T> cw = ol + mul(Rl, bs_c);
            if (w0) {
                st.reset_tmpl[2 * N + 4 + 3 * n_path] = (double)cw.x; st.reset_tmpl[2 * N + 5 + 3 * n_path] = (double)cw.y; st.reset_tmpl[2 * N + 6 + 3 * n_path] = (double)cw.z;
#pragma unroll
                for (int i = 0; i < N; ++i) { st.reset_tmpl[i] = (double)q[i]; st.reset_tmpl[N + i] = (double)qd[i]; }
                st.reset_tmpl[2 * N] = (double)used;
                st.reset_tmpl[2 * N + 2] = (double)(n_path + 1);
                st.reset_tmpl[2 * N + 3] = (double)bs_r;
                __threadfence();
                st.reset_tmpl[2 * N + 1] = 1.0;
            }
        }
        break;
    }
    literal = true;                      // the tip came near the object: the whole move again, every tick a full contact tick
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = L[kLQ + i]; qd[i] = L[kLQd + i]; }
    }   // !took_template
    // ---- the object goes back to its start (reset_object), results to HBM
    FreeBody<T> b;
    if constexpr (SHAPE == 0) {
        b.pos = load_v3(c.obj_init_pos);
        b.R = mat_from_quat(quat_from_euler((T)c.obj_init_rpy[0], (T)c.obj_init_rpy[1], (T)(c.obj_init_rpy[2] + ang)));
    } else {
        b.pos = mk((T)((double)c.obj_init_pos[0] + ix), (T)((double)c.obj_init_pos[1] + iy), (T)new_mr);
        const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
#pragma unroll
        for (int e = 0; e < 9; ++e) b.R.m[e] = ident[e];
    }
    b.v = mk<T>(0, 0, 0); b.w = mk<T>(0, 0, 0);
    if constexpr (NT == 4) {             // reset_object teleports the cube: the cached contact points go with it [A38]
        if (lane < 37) st.mani[(size_t)lane * n + env] = 0.0;
    }
    if (w0) {
        if (st.tmpl_stats != nullptr) atomicAdd(st.tmpl_stats + (took_template ? 0 : 1), 1ull);
        st.reset_ticks[env] = used;
        st.contact_code[env] = ccode;
        st.obj_mass[env] = new_mr;
#pragma unroll
        for (int i = 0; i < N; ++i) { st.q[i * n + env] = (double)q[i]; st.qd[i * n + env] = (double)qd[i]; st.qd_target[i * n + env] = 0.0; }
        store_body<T>(st, n, env, b);
    }
    if constexpr (SHAPE == 1) finish_roll<T, TOPO>(m, c, st, env, q, b, (T)scaling, 0, false);
    else finish_push<T, TOPO>(m, c, st, env, q, b, 0, false);
}

template <typename T, int TOPO, int SHAPE>
int launch_wave_t(int control_mode, int cone, int n, int n_tip, hipStream_t stream, const void* d_robot, const void* d_const, const State& st, const float* d_actions,
                  int narrowphase = 0) {
    const size_t lds_bytes = (size_t)(kLHull + (SHAPE == 1 ? 0 : 3 * n_tip) + (narrowphase ? kXWords : 0)) * sizeof(T);   // env state + per-tick hand-offs + the tip-core hull
    if (lds_bytes > 60 * 1024) return -1;                                                   // (1089 vertices: 31 KB; 4 envs per CU fit 160 KB)
    if (!cone) return -1;                // pyramid friction (enableConeFriction = 0, not what the reference sets): the lane-per-env kernels
    if constexpr (SHAPE == 0 && sizeof(T) == 8) {
        if (narrowphase) {               // GJK / EPA + manifold: the four-slot variant of the same kernel
            if (n_tip > 64 * narrow::kSlots) return -1;
            if (control_mode == TG_CONTROL_TCP_POSITION)
                hipLaunchKernelGGL((k_step_contact_wave<T, TOPO, true, 0, true, 4>), dim3(n), dim3(64), lds_bytes, stream, (const DevRobot<T>*)d_robot,
                                   (const EnvConst<T>*)d_const, st, d_actions);
            else
                hipLaunchKernelGGL((k_step_contact_wave<T, TOPO, false, 0, true, 4>), dim3(n), dim3(64), lds_bytes, stream, (const DevRobot<T>*)d_robot,
                                   (const EnvConst<T>*)d_const, st, d_actions);
            return 0;
        }
    }
    if (narrowphase) return -1;
    if (control_mode == TG_CONTROL_TCP_POSITION)
        hipLaunchKernelGGL((k_step_contact_wave<T, TOPO, true, SHAPE, true>), dim3(n), dim3(64), lds_bytes, stream, (const DevRobot<T>*)d_robot,
                           (const EnvConst<T>*)d_const, st, d_actions);
    else
        hipLaunchKernelGGL((k_step_contact_wave<T, TOPO, false, SHAPE, true>), dim3(n), dim3(64), lds_bytes, stream, (const DevRobot<T>*)d_robot,
                           (const EnvConst<T>*)d_const, st, d_actions);
    return 0;
}

template <typename T, int TOPO, int SHAPE>
int launch_reset_wave_t(int cone, int n, int n_tip, hipStream_t stream, const void* d_robot, const void* d_const, const State& st, const uint8_t* d_mask,
                        int narrowphase = 0) {
    const size_t lds_bytes = (size_t)(kLHull + (SHAPE == 1 ? 0 : 3 * n_tip) + (narrowphase ? kXWords : 0)) * sizeof(T);
    if (lds_bytes > 60 * 1024 || !cone) return -1;
    if constexpr (SHAPE == 0 && sizeof(T) == 8) {
        if (narrowphase) {
            if (n_tip > 64 * narrow::kSlots) return -1;
            hipLaunchKernelGGL((k_reset_contact_wave<T, TOPO, 0, true, 4>), dim3(n), dim3(64), lds_bytes, stream, (const DevRobot<T>*)d_robot,
                               (const EnvConst<T>*)d_const, st, d_mask, 0);
            return 0;
        }
    }
    if (narrowphase) return -1;
    const int optimistic = getenv("TG_LITERAL_RESET") == nullptr ? 1 : 0;             // TG_LITERAL_RESET: every reset tick a full contact tick (A/B, tests)
    hipLaunchKernelGGL((k_reset_contact_wave<T, TOPO, SHAPE, true>), dim3(n), dim3(64), lds_bytes, stream, (const DevRobot<T>*)d_robot,
                       (const EnvConst<T>*)d_const, st, d_mask, optimistic);
    return 0;
}

}  // namespace

int launch_step_body_wave(int physics_dtype, int topology, int control_mode, int num_envs, hipStream_t stream, const void* d_robot, const void* d_const,
                          const State& st, const float* d_actions, int inline_reset) {
    if (physics_dtype != TG_PHYSICS_F64 || topology != 0 || control_mode != TG_CONTROL_TCP_VELOCITY) return -1;
    hipLaunchKernelGGL((k_step_body_wave<double, 0>), dim3(num_envs), dim3(64), (size_t)kLHull * sizeof(double), stream, (const DevRobot<double>*)d_robot,
                       (const EnvConst<double>*)d_const, st, d_actions, inline_reset);
    return 0;
}

int launch_step_arm_wave(int physics_dtype, int topology, int control_mode, int num_envs, hipStream_t stream, const void* d_robot, const void* d_const,
                         const State& st, const float* d_actions) {
    if (physics_dtype != TG_PHYSICS_F64 || control_mode != TG_CONTROL_TCP_VELOCITY) return -1;
    if (topology == 0)
        hipLaunchKernelGGL((k_step_arm_wave<double, 0>), dim3(num_envs), dim3(64), (size_t)kLHull * sizeof(double), stream, (const DevRobot<double>*)d_robot,
                           (const EnvConst<double>*)d_const, st, d_actions);
    else
        hipLaunchKernelGGL((k_step_arm_wave<double, 1>), dim3(num_envs), dim3(64), (size_t)kLHull * sizeof(double), stream, (const DevRobot<double>*)d_robot,
                           (const EnvConst<double>*)d_const, st, d_actions);
    return 0;
}

int launch_reset_contact_wave(int env_kind, int physics_dtype, int topology, int cone_friction, int num_envs, int n_tip_verts, hipStream_t stream,
                              const void* d_robot, const void* d_const, const State& st, const uint8_t* d_mask, int narrowphase) {
    if (physics_dtype != TG_PHYSICS_F64) return -1;
    if (env_kind == TG_ENV_OBJECT_PUSH) {
        if (topology == 0) return launch_reset_wave_t<double, 0, 0>(cone_friction, num_envs, n_tip_verts, stream, d_robot, d_const, st, d_mask, narrowphase);
        return launch_reset_wave_t<double, 1, 0>(cone_friction, num_envs, n_tip_verts, stream, d_robot, d_const, st, d_mask, narrowphase);
    }
    if (narrowphase) return -1;
    if (env_kind == TG_ENV_OBJECT_ROLL && topology == 0)
        return launch_reset_wave_t<double, 0, 1>(cone_friction, num_envs, 0, stream, d_robot, d_const, st, d_mask);
    return -1;
}

int launch_step_contact_wave(int env_kind, int physics_dtype, int topology, int control_mode, int cone_friction, int num_envs, int n_tip_verts, hipStream_t stream,
                             const void* d_robot, const void* d_const, const State& st, const float* d_actions, int narrowphase) {
#if TG_WAVE_TIMING
    {
        static int calls = 0;
        if (++calls % 10 == 0) {
            unsigned long long h[16];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h);
            fprintf(stderr, "[wave timing] ticks per phase over %d steps:", calls - 1);
            const char* names[12] = {"dynamics(tail)", "body+table", "tip", "row", "G", "sweeps", "integrate", "dyn:kinematics", "dyn:own", "dyn:composites", "dyn:M", "dyn:inverse"};
            for (int k = 0; k < 12; ++k) fprintf(stderr, " %s %.0f", names[k], (double)h[k] / ((calls - 1) * 24.0));
            fprintf(stderr, " (per tick)\n");
        }
    }
#endif
    if (physics_dtype != TG_PHYSICS_F64) return -1;       // the f32 variant keeps the lane-per-env mapping
    if (env_kind == TG_ENV_OBJECT_PUSH) {
        if (topology == 0) return launch_wave_t<double, 0, 0>(control_mode, cone_friction, num_envs, n_tip_verts, stream, d_robot, d_const, st, d_actions, narrowphase);
        return launch_wave_t<double, 1, 0>(control_mode, cone_friction, num_envs, n_tip_verts, stream, d_robot, d_const, st, d_actions, narrowphase);
    }
    if (narrowphase) return -1;
    if (env_kind == TG_ENV_OBJECT_ROLL && topology == 0)
        return launch_wave_t<double, 0, 1>(control_mode, cone_friction, num_envs, 0, stream, d_robot, d_const, st, d_actions);
    return -1;
}

}  // namespace tg
