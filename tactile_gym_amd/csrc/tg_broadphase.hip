// tg_broadphase.hip - the broadphase guard: what PyBullet's stepSimulation (robots/arms/robot.py:141) does before any contact exists - Bullet's
// broadphase over the world AABBs of every collision object - restated as a CHECK of this library's fixed contact sets (include/tactile_gym_hip.h:
// tg_set_broadphase; oracle: oracle/broadphase.py, same arithmetic in the same order).
//
// One wavefront per env, one lane per box slot (TG_BP_SLOTS = 22: the robot's links, table, plane, stimulus, the free objects):
//   0. forward kinematics of the arm (every lane alike), link frames to LDS;
//   1. lane k: its box's pose source -> world centre, axes, half extents (+ margin), world AABB; all to LDS;
//   2. sort by the AABBs' lower x bound: rank = number of boxes that come before (ties by slot index) - 22 compares per lane, no exchange network;
//   3. sweep: lane k walks the boxes behind it in that order while their lower x bound is not past its upper one; y / z intervals, then the pair
//      rules (different bodies, not both static, not an expected pair) - a pair that passes is what Bullet's broadphase would hand on (stage 1);
//      then the oriented-box separating-axis test (stage 2) and, for a robot link against the table, the link's convex hull against the table top
//      (stage 3: a box bounds a round link loosely - the UR5's upper arm is a 6 cm cylinder about its joint, its box's corners reach 2.5 cm further);
//   4. wave reduction of (pairs, hits, mask of slots in hits) -> tg_state_view.broadphase_*; hits also go to the context's totals.
// HBM traffic: the env's joint angles and object pose in, three int32 out; the scene (8 KB) and the hull vertices (< 100 KB) are L2 resident.
#include "tg_broadphase.h"

#include "tg_kernels.hpp"

namespace tg {

namespace {
constexpr int kS = TG_BP_SLOTS;
constexpr int kW = 21;                                       // words per box in LDS: lo 3, hi 3, centre 3, axes 9 (row major, columns = axes), half 3

__device__ __forceinline__ bool obb_overlap(const double* a, const double* b) {
    // Gottschalk's separating-axis test; A = axes of a (columns), Rm = A^T B, t = A^T (cb - ca); |Rm| + 1e-9 on the cross-product axes
    const double* ca = a + 6; const double* Aa = a + 9; const double* ha = a + 18;
    const double* cb = b + 6; const double* Ab = b + 9; const double* hb = b + 18;
    double Rm[3][3], Ra[3][3], t[3];
    const double d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t[i] = Aa[0 * 3 + i] * d[0] + Aa[1 * 3 + i] * d[1] + Aa[2 * 3 + i] * d[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Rm[i][j] = Aa[0 * 3 + i] * Ab[0 * 3 + j] + Aa[1 * 3 + i] * Ab[1 * 3 + j] + Aa[2 * 3 + i] * Ab[2 * 3 + j];
            Ra[i][j] = fabs(Rm[i][j]) + 1e-9;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (fabs(t[i]) > ha[i] + (Ra[i][0] * hb[0] + Ra[i][1] * hb[1] + Ra[i][2] * hb[2])) return false;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (fabs(t[0] * Rm[0][j] + t[1] * Rm[1][j] + t[2] * Rm[2][j]) > (ha[0] * Ra[0][j] + ha[1] * Ra[1][j] + ha[2] * Ra[2][j]) + hb[j]) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const double ra = ha[i1] * Ra[i2][j] + ha[i2] * Ra[i1][j];
            const double rb = hb[j1] * Ra[i][j2] + hb[j2] * Ra[i][j1];
            if (fabs(t[i2] * Rm[i1][j] - t[i1] * Rm[i2][j]) > ra + rb) return false;
        }
    }
    return true;
}
}  // namespace

template <typename T, int TOPO>
__global__ __launch_bounds__(64) void k_broadphase(const DevRobot<T>* __restrict__ mp, const BpScene* __restrict__ sp, State st, int32_t* __restrict__ out,
                                                   unsigned long long* __restrict__ totals) {
    constexpr int N = Topo<TOPO>::N;
    __shared__ double fr[8][12];                             // link frames: R (row major) 9, origin 3
    __shared__ double bx[kS][kW];
    __shared__ int order[kS];
    __shared__ int acc[3];
    __shared__ int cand[16], n_cand;                         // (robot link, table) pairs whose boxes overlap: stage 3 below
    const int env = blockIdx.x, lane = threadIdx.x, n = (int)gridDim.x;
    const BpScene& sc = *sp;
    {
        T q[N];
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = (T)st.q[i * n + env];
        Kin<T, TOPO> k;
        forward_kinematics<T, TOPO>(*mp, q, k);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
#pragma unroll
                for (int e = 0; e < 9; ++e) fr[i][e] = (double)k.R[i].m[e];
                fr[i][9] = (double)k.o[i].x; fr[i][10] = (double)k.o[i].y; fr[i][11] = (double)k.o[i].z;
            }
            acc[0] = acc[1] = acc[2] = 0; n_cand = 0;
        }
    }
    __syncthreads();
    const bool slot = lane < kS;
    const tg_bp_box& b = sc.box[slot ? lane : 0];
    const int src = slot ? b.src : TG_BP_NONE;
    const bool active = src != TG_BP_NONE;
    if (slot) {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0}, scale = 1.0;
        if (src == TG_BP_LINK && b.link >= 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) R[e] = fr[b.link][e];
            p[0] = fr[b.link][9]; p[1] = fr[b.link][10]; p[2] = fr[b.link][11];
        } else if (src == TG_BP_EDGE) {
            const double ang = st.edge_ang[env], c = cos(ang), s = sin(ang);
            R[0] = c; R[1] = -s; R[3] = s; R[4] = c;
            p[0] = sc.stim_pos[0]; p[1] = sc.stim_pos[1]; p[2] = sc.stim_pos[2];
        } else if (src == TG_BP_BODY || src == TG_BP_SPHERE) {
            p[0] = st.body_pos[0 * n + env]; p[1] = st.body_pos[1 * n + env]; p[2] = st.body_pos[2 * n + env];
            if (src == TG_BP_BODY) {
#pragma unroll
                for (int e = 0; e < 9; ++e) R[e] = st.body_rot[e * n + env];
            } else scale = st.obj_mass[env] / sc.sphere_half;            // object_roll: obj_mass holds the episode's radius
        } else if (src == TG_BP_BALL) {
            p[0] = st.ball[0 * n + env]; p[1] = st.ball[1 * n + env]; p[2] = st.ball[2 * n + env];
            scale = sc.ball_radius / sc.sphere_half;
        }
        double* w = bx[lane];
        double ax[9], h[3], ctr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ctr[i] = (R[3 * i] * (scale * b.center[0]) + R[3 * i + 1] * (scale * b.center[1]) + R[3 * i + 2] * (scale * b.center[2])) + p[i];
            h[i] = scale * b.half[i] + sc.margin;
#pragma unroll
            for (int j = 0; j < 3; ++j) ax[3 * i + j] = R[3 * i] * b.rot[j] + R[3 * i + 1] * b.rot[3 + j] + R[3 * i + 2] * b.rot[6 + j];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double ext = fabs(ax[3 * i]) * h[0] + fabs(ax[3 * i + 1]) * h[1] + fabs(ax[3 * i + 2]) * h[2];
            w[i] = active ? ctr[i] - ext : 1e300;            // an empty slot sorts behind everything and ends every sweep
            w[3 + i] = active ? ctr[i] + ext : -1e300;
            w[6 + i] = ctr[i]; w[18 + i] = h[i];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) w[9 + e] = ax[e];
    }
    __syncthreads();
    int rank = 0;
    if (slot) {
        const double mylo = bx[lane][0];
        for (int j = 0; j < kS; ++j) {
            const double lj = bx[j][0];
            rank += (lj < mylo || (lj == mylo && j < lane)) ? 1 : 0;
        }
        order[rank] = lane;
    }
    __syncthreads();
    int pairs = 0, hits = 0;
    unsigned mask = 0;
    if (active) {
        const double* me = bx[lane];
        for (int r = rank + 1; r < kS; ++r) {
            const int j = order[r];
            const double* ot = bx[j];
            if (ot[0] > me[3]) break;
            const tg_bp_box& o = sc.box[j];
            if (o.body == b.body || (o.is_static && b.is_static)) continue;
            if (me[1] > ot[4] || ot[1] > me[4] || me[2] > ot[5] || ot[2] > me[5]) continue;
            if ((b.expected >> j) & 1u) continue;
            ++pairs;
            bool hit = obb_overlap(me, ot);
            if (hit && o.conj >= 0) hit = obb_overlap(me, bx[o.conj]);            // a shape bounded by two boxes (a disc): both must be reached
            if (hit && b.conj >= 0) hit = obb_overlap(bx[b.conj], ot);
            const int rl = lane == sc.table_slot ? j : (j == sc.table_slot ? lane : -1);     // the robot box of a (robot link, table) pair
            if (hit && rl >= 0 && rl < 16 && sc.box[rl].src == TG_BP_LINK && sc.box[rl].hull_n > 0) {
                cand[atomicAdd(&n_cand, 1)] = rl;            // stage 3 is the whole wavefront's work (below): at most 16 such pairs
                hit = false;
            }
            if (hit) { ++hits; mask |= (1u << lane) | (1u << j); }
        }
    }
    if (pairs) atomicAdd(&acc[0], pairs);
    if (hits) { atomicAdd(&acc[1], hits); atomicOr(&acc[2], (int)mask); }
    __syncthreads();
    // stage 3: a robot link whose BOX reaches the table - its convex hull decides (lowest vertex over the table top, less the hull's collision margin
    // and the guard's).  The lanes share a candidate's vertices (a sensor tip's hull has 1089), the minimum is a wave reduction; min is exact in any order.
    for (int k = 0; k < n_cand; ++k) {
        const int rl = cand[k];
        const tg_bp_box& rb = sc.box[rl];
        const double* tb = bx[sc.table_slot];
        const double* F = fr[rb.link >= 0 ? rb.link : 0];
        double zmin = 1e300;
        for (int v = lane; v < rb.hull_n; v += 64) {
            const double* hv = sc.hull + 3 * (size_t)(rb.hull_off + v);
            double wx = hv[0], wy = hv[1], wz = hv[2];
            if (rb.link >= 0) {
                wx = (F[0] * hv[0] + F[1] * hv[1] + F[2] * hv[2]) + F[9];
                wy = (F[3] * hv[0] + F[4] * hv[1] + F[5] * hv[2]) + F[10];
                wz = (F[6] * hv[0] + F[7] * hv[1] + F[8] * hv[2]) + F[11];
            }
            if (wx >= tb[0] && wx <= tb[3] && wy >= tb[1] && wy <= tb[4] && wz < zmin) zmin = wz;
        }
        for (int o = 1; o < 64; o <<= 1) { const double other = __shfl_xor(zmin, o); zmin = other < zmin ? other : zmin; }
        if (lane == 0 && zmin < 1e299 && (zmin - sc.hull_margin) - sc.margin <= tb[5]) {
            acc[1] += 1; acc[2] |= (1 << rl) | (1 << sc.table_slot);
        }
    }
    __syncthreads();
    if (lane == 0) {
        out[env] = acc[0]; out[n + env] = acc[1]; out[2 * n + env] = acc[2];
        if (acc[0] || acc[1]) { atomicAdd(totals + 1, (unsigned long long)acc[0]); atomicAdd(totals + 2, (unsigned long long)acc[1]); }
        if (env == 0) atomicAdd(totals + 0, (unsigned long long)n);
    }
}

int launch_broadphase(int physics_dtype, int topology, int n, hipStream_t stream, const void* d_robot, const BpScene* d_scene, const State& st,
                      int32_t* out, unsigned long long* totals) {
#define TG_BP_LAUNCH(T, TOPO) hipLaunchKernelGGL((k_broadphase<T, TOPO>), dim3(n), dim3(64), 0, stream, (const DevRobot<T>*)d_robot, d_scene, st, out, totals)
    if (physics_dtype == TG_PHYSICS_F64) {
        if (topology == 0) TG_BP_LAUNCH(double, 0); else if (topology == 1) TG_BP_LAUNCH(double, 1); else return -1;
    } else {
        if (topology == 0) TG_BP_LAUNCH(float, 0); else if (topology == 1) TG_BP_LAUNCH(float, 1); else return -1;
    }
#undef TG_BP_LAUNCH
    return 0;
}

}  // namespace tg
