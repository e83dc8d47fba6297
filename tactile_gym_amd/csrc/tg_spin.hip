// tg_spin.hip - object_balance, object_mode "spinning_plate" (object_balance_env.py:107-108, 198-239, 267-269, 355-358): the spool
// (plate_buffer.urdf) hangs on the TCP's point-to-point constraint and the dish (spinning_plate.urdf, the env's object) stands on its spindle.
// One stepSimulation() tick = oracle/minibullet.c: mb_step_spin [PARITY_ASSUMPTIONS A41]:
//   arm, spool (Bullet's default velocity damping), dish (the one-tick torque / force of reset_object, no damping): unconstrained velocities;
//   the pair's contacts: AABB test, GJK / EPA of the two convex hulls in the spool's frame (tg_narrowphase.hpp), one new point per tick into the
//   persistent manifold of up to four, refresh;
//   ONE projected Gauss-Seidel loop over the joint motors and the three P2P rows (reverse order on even sweeps), the contacts' normals
//   (lambda >= 0), then their friction pairs under the cone.
//
// Mapping: one wavefront per env.  The arm's dynamics and the bodies' 3 x 3 algebra are wave-uniform (every lane computes the env's values, as
// k_step_body_wave does); the hulls are spread over the lanes (support queries = one arg-max over the wavefront); the solve is the Delassus /
// residual form of the other body kernels with one ROW PER LANE: J, W = Minv_sys J^T and A = J W sit in LDS, lane j carries the residual
// r_j = rhs_j - sum_q A_jq lambda_q with its impulse, limit and 1 / A_jj; a row update is  t = r_i / A_ii  (row i's values by v_readlane), clamp,
// and every lane's  r_j -= A_ji delta  (one LDS read each, stride 21 words: conflict-free).  21 rows at most: 6 motors + 3 P2P + 4 x (normal + 2 friction).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "tg_kernels.hpp"
#include "tg_narrowphase.hpp"
#include "tg_spin.h"

namespace tg {
namespace {

constexpr int kNC = 4;
template <int N> struct SL {   // LDS layout of one env, in doubles
    static constexpr int NU = N + 12, NP = N + 3, NR = NP + 3 * kNC;
    static constexpr int J = 0, W = J + NR * NU, A = W + NU * NR, V = A + NR * NR, LAM = V + NU, DIAG = LAM + NR, LIM = DIAG + NR, TGT = LIM + NR,
                         MI = TGT + NR, IB = MI + N * N, ID = IB + 9, SCR = ID + 9, MANI = SCR + narrow::kScratchWords,
                         HULL = MANI + narrow::kManiWords;   // the dish's hull (3 n_dish words) follows
};
using lds = narrow::lptr<double>;

__device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ void s3_to9(const S3<double>& s, lds o) {
    o[0] = s.xx; o[1] = s.xy; o[2] = s.xz; o[3] = s.xy; o[4] = s.yy; o[5] = s.yz; o[6] = s.xz; o[7] = s.yz; o[8] = s.zz;
}

// One tick.  Everything wave-uniform except `lane`; L = the env's LDS block.  Returns the sweeps it ran.
template <int TOPO, int MOTOR>
__device__ __noinline__ int sim_tick_spin(const DevRobot<double>& m, double (&q)[Topo<TOPO>::N], double (&qd)[Topo<TOPO>::N],
                                          const double (&q_des)[Topo<TOPO>::N], const double (&qd_des)[Topo<TOPO>::N], double kp, double kd,
                                          double max_force, double dt, int iters, V3<double> gravity, FreeBody<double>& b, const BodyConst<double>& bc,
                                          V3<double> pivot_b, FreeBody<double>& dsh, const SpinConst<double>& sc, const narrow::HullB& HB,
                                          V3<double> fext, V3<double> pext, bool pending, lds L, int lane, double& normal_impulse, int& contacts) {
    using T = double;
    constexpr int N = Topo<TOPO>::N;
    using Y = SL<N>;
    constexpr int NU = Y::NU, NP = Y::NP, NR = Y::NR;
#ifdef TG_SPIN_STAMPS
    unsigned long long stamp_[9];
#define TG_SPIN_STAMP(k) stamp_[k] = wall_clock64()
#else
#define TG_SPIN_STAMP(k)
#endif
    TG_SPIN_STAMP(0);
    // ---- arm: unconstrained velocity
    T v[N];
    T Jt[3][N];
    V3<T> pa;
    {
        T hb[N], qdm[N], Minv[N][N], traceM;
        Kin<T, TOPO> kin;
        dynamics_terms<T, TOPO, false>(m, q, qd, hb, qdm, Minv, traceM, gravity, kin);
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
        M3<T> Rl;
        const T ident[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
        const T pav[3] = {bc.pivot_a.x, bc.pivot_a.y, bc.pivot_a.z};
        link_frame<T, TOPO>(kin, bc.link, pav, ident, pa, Rl);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            bool on_path = false;
#pragma unroll
            for (int l = 0; l < N; ++l)
                if (l == bc.link && is_ancestor_or_self<TOPO>(i, l)) on_path = true;
            const V3<T> jt = cross(kin.a[i], pa - kin.o[i]);
            Jt[0][i] = on_path ? jt.x : T(0); Jt[1][i] = on_path ? jt.y : T(0); Jt[2][i] = on_path ? jt.z : T(0);
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) L[Y::MI + i * N + j] = Minv[i][j];
    }
    TG_SPIN_STAMP(1);
    // ---- spool: gravity, Bullet's default damping F = -m v (K + K |v|) [A27], gyroscopic torque
    const S3<T> Iw = rotate(b.R, bc.inertia), Iwi = inverse(Iw);
    V3<T> xc = b.pos + mul(b.R, bc.com);
    const T sv = sc.lin_damp + sc.lin_damp * norm(b.v), sw = sc.ang_damp + sc.ang_damp * norm(b.w);
    const V3<T> Iwv = mul(Iw, b.w);
    const V3<T> vb = b.v + dt * (gravity - sv * b.v);
    const V3<T> wb = b.w + dt * mul(Iwi, (mk<T>(0, 0, 0) - sw * Iwv) - cross(b.w, Iwv));
    // ---- dish: gravity, the one-tick force / torque (reset_object :357-358), gyroscopic torque, no damping (:338-345)
    const S3<T> Dw = rotate(dsh.R, sc.inertia), Dwi = inverse(Dw);
    V3<T> xd = dsh.pos + mul(dsh.R, sc.com);
    V3<T> Fd = sc.mass * gravity, Nd = mk<T>(0, 0, 0);
    if (pending) {
        Fd = Fd + fext;
        Nd = cross(pext - xd, fext) + mul(dsh.R, mk<T>(0, 0, -1));          // apply_random_torque_obj(1.0): LINK_FRAME (0, 0, -1)
    }
    {
        const V3<T> Dv = mul(Dw, dsh.w);
        Nd = Nd - cross(dsh.w, Dv);
    }
    const V3<T> vd = dsh.v + (dt / sc.mass) * Fd, wd = dsh.w + dt * mul(Dwi, Nd);
    s3_to9(Iwi, L + Y::IB);
    s3_to9(Dwi, L + Y::ID);
    TG_SPIN_STAMP(2);
    // ---- narrowphase: the dish's hull in the spool's frame, AABBs, GJK / EPA, the manifold (tg_contact_wave.hip's NT = 4 block, two hulls)
    const lds mf = L + Y::MANI;
    {
        double ol[3] = {dsh.pos.x, dsh.pos.y, dsh.pos.z}, Rl[9], bp[3] = {b.pos.x, b.pos.y, b.pos.z}, bR[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) { Rl[e] = dsh.R.m[e]; bR[e] = b.R.m[e]; }
        const int n_dish = __builtin_amdgcn_readfirstlane(sc.n_dish);
        narrow::Hull H;
        H.n = n_dish;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, lob[3] = {1e300, 1e300, 1e300}, hib[3] = {-1e300, -1e300, -1e300};
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < narrow::kSlots; ++k) {
                if (64 * k >= n_dish) break;
                const int i = 64 * k + lane;
                const int ii = i < n_dish ? i : 0;
                const double v0 = L[Y::HULL + 3 * ii], v1 = L[Y::HULL + 3 * ii + 1], v2 = L[Y::HULL + 3 * ii + 2];
                double w[3], dd[3];
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    w[x] = ol[x] + ((Rl[3 * x] * v0 + Rl[3 * x + 1] * v1) + Rl[3 * x + 2] * v2);
                    dd[x] = w[x] - bp[x];
                    if (i < n_dish) { lo[x] = w[x] < lo[x] ? w[x] : lo[x]; hi[x] = w[x] > hi[x] ? w[x] : hi[x]; }
                }
                H.x[k] = (bR[0] * dd[0] + bR[3] * dd[1]) + bR[6] * dd[2];
                H.y[k] = (bR[1] * dd[0] + bR[4] * dd[1]) + bR[7] * dd[2];
                H.z[k] = (bR[2] * dd[0] + bR[5] * dd[1]) + bR[8] * dd[2];
            }
#pragma unroll
            for (int k = 0; k < narrow::kSlotsB; ++k) {
                const int i = 64 * k + lane;
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const double w = bp[x] + ((bR[3 * x] * HB.x[k] + bR[3 * x + 1] * HB.y[k]) + bR[3 * x + 2] * HB.z[k]);
                    if (i < HB.n) { lob[x] = w < lob[x] ? w : lob[x]; hib[x] = w > hib[x] ? w : hib[x]; }
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) { lo[x] = wave_min(lo[x]); hi[x] = wave_max(hi[x]); lob[x] = wave_min(lob[x]); hib[x] = wave_max(hib[x]); }
        bool overlap = n_dish > 0 && HB.n > 0;
        {
#pragma clang fp contract(off)
            const double pad = 2.0 * sc.margin + sc.breaking;
#pragma unroll
            for (int x = 0; x < 3; ++x)
                if (lo[x] - pad > hib[x] || hi[x] + pad < lob[x]) overlap = false;
        }
        __syncthreads();
        if (__builtin_amdgcn_ballot_w64(overlap) == 0) { if (lane == 0) mf[narrow::kMcount] = 0.0; }
        __syncthreads();
        if (__builtin_amdgcn_ballot_w64(overlap) != 0) {
            double sd = 0.0, nb[3], ab[3], bb[3];
            if (narrow::gjk_epa_hull_hull(H, HB, L + Y::SCR, sd, nb, ab, bb, lane)) {
#pragma clang fp contract(off)
                const double depth = sd - 2.0 * sc.margin;
                double nw[3], aw[3], bw[3], pa_[3], pb_[3];
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    nw[x] = (bR[3 * x] * nb[0] + bR[3 * x + 1] * nb[1]) + bR[3 * x + 2] * nb[2];
                    aw[x] = (bR[3 * x] * ab[0] + bR[3 * x + 1] * ab[1]) + bR[3 * x + 2] * ab[2];
                    bw[x] = (bR[3 * x] * bb[0] + bR[3 * x + 1] * bb[1]) + bR[3 * x + 2] * bb[2];
                }
#pragma unroll
                for (int x = 0; x < 3; ++x) { pa_[x] = (bp[x] + aw[x]) - nw[x] * sc.margin; pb_[x] = (bp[x] + bw[x]) + nw[x] * sc.margin; }
                narrow::manifold_add(mf, sc.breaking, ol, Rl, bp, bR, pa_, pb_, nw, depth, lane);
            }
            narrow::manifold_refresh(mf, sc.breaking, ol, Rl, bp, bR, lane);
        }
        __syncthreads();
    }
    const int nc = __builtin_amdgcn_readfirstlane((int)mf[narrow::kMcount]);
    const int nr = NP + 3 * nc;
    contacts = nc;
    TG_SPIN_STAMP(3);
    // ---- rows into LDS: J [NR][NU] (zeroed, then filled), the unconstrained velocity v [NU]
    for (int e = lane; e < NR * NU; e += 64) L[Y::J + e] = 0.0;
    __syncthreads();
    const V3<T> pb = b.pos + mul(b.R, pivot_b);
    const V3<T> rb = pb - xc;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            L[Y::J + i * NU + i] = 1.0;
#pragma unroll
            for (int x = 0; x < 3; ++x) L[Y::J + (N + x) * NU + i] = Jt[x][i];
            L[Y::V + i] = v[i];
        }
        const V3<T> e3[3] = {mk<T>(1, 0, 0), mk<T>(0, 1, 0), mk<T>(0, 0, 1)};
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const V3<T> rxe = cross(rb, e3[x]);
            L[Y::J + (N + x) * NU + N + x] = -1.0;
            L[Y::J + (N + x) * NU + N + 3] = -rxe.x; L[Y::J + (N + x) * NU + N + 4] = -rxe.y; L[Y::J + (N + x) * NU + N + 5] = -rxe.z;
        }
        const T vbv[12] = {vb.x, vb.y, vb.z, wb.x, wb.y, wb.z, vd.x, vd.y, vd.z, wd.x, wd.y, wd.z};
#pragma unroll
        for (int k = 0; k < 12; ++k) L[Y::V + N + k] = vbv[k];
    }
    if (lane < nc) {                       // lane q: the three rows of contact q (dish +d, spool -d)
        const int qc = lane;
        const V3<T> nrm = mk(mf[narrow::kMn + 3 * qc], mf[narrow::kMn + 3 * qc + 1], mf[narrow::kMn + 3 * qc + 2]);
        const V3<T> cpa = mk(mf[narrow::kMpa + 3 * qc], mf[narrow::kMpa + 3 * qc + 1], mf[narrow::kMpa + 3 * qc + 2]);
        const V3<T> cpb = mk(mf[narrow::kMpb + 3 * qc], mf[narrow::kMpb + 3 * qc + 1], mf[narrow::kMpb + 3 * qc + 2]);
        V3<T> d[3];
        d[0] = nrm;
        plane_space(nrm, d[1], d[2]);
        const V3<T> rda = cpa - xd, rsb = cpb - xc;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const V3<T> rxa = cross(rda, d[r]), rxb = cross(rsb, d[r]);
            const int row = Y::J + (NP + 3 * qc + r) * NU;
            L[row + N + 6] = d[r].x; L[row + N + 7] = d[r].y; L[row + N + 8] = d[r].z;
            L[row + N + 9] = rxa.x; L[row + N + 10] = rxa.y; L[row + N + 11] = rxa.z;
            L[row + N + 0] = -d[r].x; L[row + N + 1] = -d[r].y; L[row + N + 2] = -d[r].z;
            L[row + N + 3] = -rxb.x; L[row + N + 4] = -rxb.y; L[row + N + 5] = -rxb.z;
        }
    }
    __syncthreads();
    TG_SPIN_STAMP(4);
    // ---- W = Minv_sys J^T  [NU][NR]
    const T imb = T(1) / bc.mass, imd = T(1) / sc.mass;
    for (int e = lane; e < NU * NR; e += 64) {
        const int u = e / NR, r = e - u * NR;
        const int row = Y::J + r * NU;
        T acc = T(0);
        if (u < N) {
#pragma unroll
            for (int j = 0; j < N; ++j) acc += L[Y::MI + u * N + j] * L[row + j];
        } else if (u < N + 3) acc = L[row + u] * imb;
        else if (u < N + 6) { const int x = u - N - 3; acc = (L[Y::IB + 3 * x] * L[row + N + 3] + L[Y::IB + 3 * x + 1] * L[row + N + 4]) + L[Y::IB + 3 * x + 2] * L[row + N + 5]; }
        else if (u < N + 9) acc = L[row + u] * imd;
        else { const int x = u - N - 9; acc = (L[Y::ID + 3 * x] * L[row + N + 9] + L[Y::ID + 3 * x + 1] * L[row + N + 10]) + L[Y::ID + 3 * x + 2] * L[row + N + 11]; }
        L[Y::W + e] = acc;
    }
    __syncthreads();
    // ---- A = J W  [NR][NR], its diagonal
    for (int e = lane; e < NR * NR; e += 64) {
        const int j = e / NR, i = e - j * NR;
        T acc = T(0);
        for (int u = 0; u < NU; ++u) acc += L[Y::J + j * NU + u] * L[Y::W + u * NR + i];
        L[Y::A + e] = acc;
        if (i == j) L[Y::DIAG + i] = acc;
    }
    TG_SPIN_STAMP(5);
    // ---- right-hand sides: lane j's row
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const T pos_term = (MOTOR == kMotorPosition) ? kp * (q_des[i] - q[i]) / dt : T(0);
            const T des = pos_term + v[i] + kd * (qd_des[i] - v[i]);
            L[Y::TGT + i] = des - v[i];
            L[Y::LIM + i] = max_force * dt;
        }
        const V3<T> gap = pa - pb;
        L[Y::TGT + N] = -bc.erp * gap.x / dt; L[Y::TGT + N + 1] = -bc.erp * gap.y / dt; L[Y::TGT + N + 2] = -bc.erp * gap.z / dt;
        L[Y::LIM + N] = bc.max_impulse; L[Y::LIM + N + 1] = bc.max_impulse; L[Y::LIM + N + 2] = bc.max_impulse;
    }
    for (int e = lane; e < NR; e += 64) L[Y::LAM + e] = 0.0;
    __syncthreads();
    T rj = T(0);
    if (lane < nr) {
        const int j = lane;
        T cv = T(0);
        for (int u = 0; u < NU; ++u) cv += L[Y::J + j * NU + u] * L[Y::V + u];
        if (j < N) rj = L[Y::TGT + j];
        else if (j < NP) rj = L[Y::TGT + j] - cv;
        else {
            const int qc = (j - NP) / 3, r = (j - NP) - 3 * qc;
            const T depth = mf[narrow::kMdepth + qc];
            rj = (r == 0) ? (depth > T(0) ? (-cv - depth / dt) : (-depth * sc.erp / dt - cv)) : -cv;      // restitution 0
        }
    }
    TG_SPIN_STAMP(6);
    // ---- projected Gauss-Seidel.  Lane j keeps its row's residual, impulse, limit and 1 / A_jj in registers; a row update reads row i's four
    // values through v_readlane and column i of A from LDS (until round 6's last hour the impulses, limits and diagonal sat in LDS too: three
    // dependent LDS round trips and a division per row, 6.6 ms per step at 1024 envs)
    // Every lane evaluates ITS row's candidate update from its own registers (t = r_j / A_jj, the clamp between its bounds, the impulse change);
    // the row whose turn it is hands its change to the wavefront (one v_readlane pair) and commits its impulse.  The dependent chain of a row is
    // FMA - mul - add - max - min - select - readlane - FMA (the first version read four row values through readlanes and clamped with compares:
    // ~350 cycles per row).
    const int n_it = iters < 0 ? -iters : iters;
    const T ajj = lane < nr ? L[Y::DIAG + lane] : T(1);
    const T jdj = lane < nr ? T(1) / ajj : T(0);
    const T hij = lane < NP ? L[Y::LIM + lane] : T(1e300);      // motors, P2P: +-limit; normals: [0, inf); (friction rows: the cone, below)
    const T loj = lane < NP ? -hij : T(0);
    T lamj = T(0);
    T thr = wave_max(lane < nr ? tabs(rj * jdj) : T(0));
    thr = iters < 0 ? T(-1) : thr * T(1.3877787807814457e-17);
    const bool thr_mode = __builtin_amdgcn_ballot_w64(m.res_thr > T(0)) != 0;
    int ran = 0;
    T acol[NR];                                         // lane j's row of A (= column, A is symmetric): the sweeps touch no memory
    {
        const int arow = Y::A + (lane < NR ? lane : 0) * NR;                           // (lanes >= nr carry rows of zeros; lanes >= NR shadow row 0, unread)
#pragma unroll
        for (int i = 0; i < NR; ++i) acol[i] = L[arow + i];
    }
    auto sweeps = [&](auto thr_tag) {
        constexpr bool THR = decltype(thr_tag)::value;
        auto row = [&](const int i, T& res) {           // i is a literal after unrolling
            const T t = rj * jdj;
            const T sum = lamj + t;
            const T sc_ = vmin(vmax(sum, loj), hij);
            const T dj = (sc_ == sum) ? t : sc_ - lamj;
            const T delta = narrow::rdlane(dj, i);
            lamj = lane == i ? sc_ : lamj;
            rj -= acol[i] * delta;
            if (THR) { const T dvel = delta * narrow::rdlane(ajj, i); res = tmax(res, dvel * dvel); }
        };
        for (int it = 0; it < n_it; ++it) {
            if (!THR && (it & 7) == 0 && it > 0) {
                const T mx = wave_max(lane < nr ? tabs(rj * jdj) : T(0));
                if (mx <= thr) break;
            }
            T res = T(0);
            if (it & 1) {                               // motors and P2P rows: reversed on even sweeps
#pragma unroll
                for (int i = 0; i < NP; ++i) row(i, res);
            } else {
#pragma unroll
                for (int i = NP - 1; i >= 0; --i) row(i, res);
            }
#pragma unroll
            for (int qc = 0; qc < kNC; ++qc)            // contact normals
                if (qc < nc) row(NP + 3 * qc, res);
#pragma unroll
            for (int qc = 0; qc < kNC; ++qc) {          // friction pairs, cone (enableConeFriction = 1)
                if (qc >= nc) break;
                const int i1 = NP + 3 * qc + 1, i2 = i1 + 1;
                const T sum = lamj + rj * jdj;
                const T limit = sc.mu * narrow::rdlane(lamj, i1 - 1);
                const T l1 = narrow::rdlane(lamj, i1), l2 = narrow::rdlane(lamj, i2);
                T s1 = narrow::rdlane(sum, i1), s2 = narrow::rdlane(sum, i2);
                friction_clamp(s1, s2, limit, true);    // the cone, branch-free through v_rsq (tg_physics.hpp; as the push / roll kernels)
                const T d1 = s1 - l1, d2 = s2 - l2;
                lamj = lane == i1 ? s1 : (lane == i2 ? s2 : lamj);
                rj -= acol[i1] * d1 + acol[i2] * d2;
                if (THR) { const T dvel = d1 * narrow::rdlane(ajj, i1) + d2 * narrow::rdlane(ajj, i2); res = tmax(res, dvel * dvel); }   // one residual per cone pair [A7c]
            }
            ++ran;
            if (THR && res <= m.res_thr) break;
        }
    };
    if (thr_mode) sweeps(std::true_type{}); else sweeps(std::false_type{});
    if (lane < NR) L[Y::LAM + lane] = lane < nr ? lamj : T(0);
    __syncthreads();
    TG_SPIN_STAMP(7);
    // ---- impulses to velocities: dv = W lambda (lane u), then the env's new state (wave-uniform again)
    if (lane < NU) {
        T acc = T(0);
        for (int r = 0; r < nr; ++r) acc += L[Y::W + lane * NR + r] * L[Y::LAM + r];
        L[Y::V + lane] += acc;
    }
    __syncthreads();
    T imp = T(0);
    for (int qc = 0; qc < nc; ++qc) imp += L[Y::LAM + NP + 3 * qc];
    normal_impulse = imp;
#pragma unroll
    for (int i = 0; i < N; ++i) { qd[i] = L[Y::V + i]; q[i] += dt * qd[i]; }
    b.v = mk(L[Y::V + N], L[Y::V + N + 1], L[Y::V + N + 2]); b.w = mk(L[Y::V + N + 3], L[Y::V + N + 4], L[Y::V + N + 5]);
    dsh.v = mk(L[Y::V + N + 6], L[Y::V + N + 7], L[Y::V + N + 8]); dsh.w = mk(L[Y::V + N + 9], L[Y::V + N + 10], L[Y::V + N + 11]);
    xc = xc + dt * b.v;
    integrate_rotation(b.R, b.w, dt);
    b.pos = xc - mul(b.R, bc.com);
    xd = xd + dt * dsh.v;
    integrate_rotation(dsh.R, dsh.w, dt);
    dsh.pos = xd - mul(dsh.R, sc.com);
    __syncthreads();
    TG_SPIN_STAMP(8);
#ifdef TG_SPIN_STAMPS
    if (blockIdx.x == 0 && lane == 0 && pending) {
        printf("spin tick (env 0, first tick of an episode; 100 MHz ticks): dynamics %llu bodies %llu narrowphase %llu J %llu W+A %llu rhs %llu sweeps(%d) %llu finish %llu\n",
               stamp_[1] - stamp_[0], stamp_[2] - stamp_[1], stamp_[3] - stamp_[2], stamp_[4] - stamp_[3], stamp_[5] - stamp_[4], stamp_[6] - stamp_[5], ran,
               stamp_[7] - stamp_[6], stamp_[8] - stamp_[7]);
    }
#endif
    return ran;
}

// BaseTactileEnv.step for object_balance "spinning_plate": one wavefront per env (k_step_body's prologue and epilogue, wave-uniform)
template <int TOPO, bool POS>
__global__ __launch_bounds__(64) void k_step_spin(const DevRobot<double>* __restrict__ mp, const EnvConst<double>* __restrict__ cp, State st,
                                                  const float* __restrict__ actions) {
    using T = double;
    constexpr int N = Topo<TOPO>::N;
    using Y = SL<N>;
    extern __shared__ double L_[];
    const lds L = (lds)L_;
    const DevRobot<T>& m = *mp;
    const EnvConst<T>& c = *cp;
    const int env = blockIdx.x, lane = threadIdx.x;
    const int n = c.num_envs;
    const SpinConst<T>& sc = c.spin;
    // the dish's hull into LDS, the spool's onto the lanes, the manifold's persistent part (local anchors, normals, count)
    for (int w = lane; w < 3 * sc.n_dish; w += 64) L[Y::HULL + w] = st.spin_hulls[w];
    narrow::HullB HB;
    HB.n = sc.n_spool;
#pragma unroll
    for (int k = 0; k < narrow::kSlotsB; ++k) {
        const int i = 64 * k + lane;
        const bool in = i < sc.n_spool;
        const double* h = st.spin_hulls + (size_t)3 * sc.n_dish + (size_t)3 * (in ? i : 0);
        HB.x[k] = in ? h[0] : 0.0; HB.y[k] = in ? h[1] : 0.0; HB.z[k] = in ? h[2] : 0.0;
    }
    if (lane < 37) L[Y::MANI + (lane < 36 ? lane : narrow::kMcount)] = st.mani[(size_t)lane * n + env];
    __syncthreads();
    T q[N], qd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = st.q[i * n + env]; qd[i] = st.qd[i * n + env]; }
    FreeBody<T> b = load_body<T>(st, n, env);
    FreeBody<T> d = load_dish<T>(st, n, env);
    T enc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};   // encode_actions (object_balance_env.py:398-424)
    const float* a = actions + (size_t)env * c.act_dim;
    if (c.movement_mode == TG_BMOVE_XY) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; }
    else if (c.movement_mode == TG_BMOVE_XYZ) { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[2] = (T)a[2]; }
    else if (c.movement_mode == TG_BMOVE_RXRY) { enc[3] = (T)a[0]; enc[4] = (T)a[1]; }
    else { enc[0] = (T)a[0]; enc[1] = (T)a[1]; enc[3] = (T)a[2]; enc[4] = (T)a[3]; }
    T vels[6];
    scale_actions<T>(c, enc, vels);
    const int step_count = st.step_count[env] + 1;
    T qd_des[N];
    V3<T> tpos; Q4<T> tq;
    if constexpr (POS) tcp_position_target<T, TOPO>(m, c, q, vels, tpos, tq, qd_des);
    else tcp_velocity_control<T, TOPO>(m, c, q, vels, qd_des);
    const T embed = (T)st.embed[env];
    const V3<T> grav = mk(T(0), T(0), (T)st.gravity[env]);
    const V3<T> pivot_b = mk(T(0), T(0), -sc.buffer_height / T(2) + sc.embed0);
    const V3<T> fext = load_v3(c.ext_force);
    const V3<T> pext = mk((T)st.ext_pos[0 * n + env], (T)st.ext_pos[1 * n + env], (T)st.ext_pos[2 * n + env]);
    const bool pending = st.ext_pending[env] != 0;
    T qdummy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) qdummy[i] = T(0);
    int sweeps = 0, contacts = 0;
    T imp = T(0);
    if constexpr (POS) {                                  // blocking_move(max_steps, constant_vel=None), robot.py:188-260
        for (int t = 0; t < c.max_blocking; ++t) {
            const bool stop = pose_reached<T, TOPO>(m, q, qd, tpos, tq);
            sweeps += sim_tick_spin<TOPO, kMotorPosition>(m, q, qd, qd_des, qdummy, m.pos_gain, m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, b,
                                                          c.body, pivot_b, d, sc, HB, fext, pext, pending && t == 0, L, lane, imp, contacts);
            if (__builtin_amdgcn_ballot_w64(stop) != 0) break;
        }
    } else {
        for (int t = 0; t < c.action_repeat; ++t)
            sweeps += sim_tick_spin<TOPO, kMotorVelocity>(m, q, qd, qdummy, qd_des, T(0), m.vel_gain, m.max_force, c.dt, c.solver_iters, grav, b, c.body,
                                                          pivot_b, d, sc, HB, fext, pext, pending && t == 0, L, lane, imp, contacts);
    }
    __syncthreads();
    if (lane < 37) st.mani[(size_t)lane * n + env] = L[Y::MANI + (lane < 36 ? lane : narrow::kMcount)];
    // every lane holds the env's new state: the stores below are the same value to the same address from all of them (as k_step_body_wave's)
    st.step_count[env] = step_count;
#pragma unroll
    for (int i = 0; i < N; ++i) { st.q[i * n + env] = q[i]; st.qd[i * n + env] = qd[i]; st.qd_target[i * n + env] = POS ? 0.0 : qd_des[i]; }
    if (m.res_thr > T(0)) st.sweeps[env] = sweeps;
    st.ext_pending[env] = 0;
    store_body<T>(st, n, env, b);
    store_dish<T>(st, n, env, d, imp, contacts);
    Kin<T, TOPO> k;
    forward_kinematics<T, TOPO>(m, q, k);
    V3<T> ptcp, pbs; M3<T> Rtcp, Rbs;
    link_frame<T, TOPO>(k, m.tcp_link, m.tcp_pos, m.tcp_rot, ptcp, Rtcp);
    link_frame<T, TOPO>(k, m.sensor_link, m.sensor_pos, m.sensor_rot, pbs, Rbs);
    (void)finish_body_frames<T, TOPO>(m, c, st, env, ptcp, Rtcp, pbs, Rbs, b, embed, step_count, true, &d);
}

}  // namespace

int launch_step_spin(int physics_dtype, int topology, int control_mode, int num_envs, int n_dish, hipStream_t stream, const void* d_robot,
                     const void* d_const, const State& st, const float* d_actions) {
    if (physics_dtype != TG_PHYSICS_F64 || topology != 0 || n_dish <= 0 || n_dish > 64 * narrow::kSlots) return -1;
    const size_t lds_bytes = (size_t)(SL<6>::HULL + 3 * n_dish) * sizeof(double);
    if (control_mode == TG_CONTROL_TCP_POSITION)
        hipLaunchKernelGGL((k_step_spin<0, true>), dim3(num_envs), dim3(64), lds_bytes, stream, (const DevRobot<double>*)d_robot,
                           (const EnvConst<double>*)d_const, st, d_actions);
    else
        hipLaunchKernelGGL((k_step_spin<0, false>), dim3(num_envs), dim3(64), lds_bytes, stream, (const DevRobot<double>*)d_robot,
                           (const EnvConst<double>*)d_const, st, d_actions);
    return 0;
}

}  // namespace tg
