// tg_narrowphase.hpp - general convex narrowphase of the tip core - cube pair on ONE wavefront (object_push, tg_config.narrowphase):
// support-mapping GJK distance, EPA penetration and btPersistentManifold's cache rules.  stepSimulation's collision detection for this pair
// (robots/arms/robot.py:141; pair set-up object_push_env.py:216-225, sensors/tactile_sensor.py:322-332); Bullet's source is not in
// /root/reference, so the algorithms are the published ones (Gilbert-Johnson-Keerthi 1988, van den Bergen 2001) and the manifold policy is
// stated in PARITY_ASSUMPTIONS A35-A38.  The CPU oracle's restatement is oracle/narrowphase.c; both follow the same operation order without
// FMA contraction, so identical inputs give identical bits.
//
// Mapping.  The hull's vertices (<= 1152, in the box frame) are spread over the lanes, vertex i on lane i & 63 in slot i >> 6: a support query
// is ten FMAs per lane and one (key, index) arg-max over the wavefront (ties: the lower index, what a sequential scan finds), the winner's
// coordinates come back through v_readlane.  Everything else - the simplex solve, the expanding polytope (vertices / faces / horizon edges in
// LDS), the manifold - is wave-uniform: every lane computes the same values, lane 0 writes LDS.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

namespace tg {
namespace narrow {

template <typename T> using lptr = __attribute__((address_space(3))) T*;

constexpr int kSlots = 18;                                    // hull vertices per lane (n <= 1152: the TacTip core has 1089)
constexpr int kMaxV = 48, kMaxF = 96, kMaxE = 48;             // expanding polytope capacities (oracle/narrowphase.c: the same)
// scratch layout in doubles: V [kMaxV][6] (w, a), FN [kMaxF][4] (n, d), then ints: FV [kMaxF][4] (v0 v1 v2 alive), E [kMaxE][2]
constexpr int kOffV = 0, kOffFN = kOffV + 6 * kMaxV, kOffFV = kOffFN + 4 * kMaxF, kOffE = kOffFV + 2 * kMaxF, kScratchWords = kOffE + kMaxE;
// manifold layout in doubles: la [4][3], lb [4][3], nrm [4][3], pa [4][3], pb [4][3], depth [4], n
constexpr int kMla = 0, kMlb = 12, kMn = 24, kMpa = 36, kMpb = 48, kMdepth = 60, kMcount = 64, kManiWords = 65;

struct Hull { double x[kSlots], y[kSlots], z[kSlots]; int n; };
struct SV { double w[3], a[3]; };

__device__ __forceinline__ double dot3(const double* a, const double* b) {
#pragma clang fp contract(off)
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
#pragma clang fp contract(off)
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
__device__ __forceinline__ double rdlane(double v, int src) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
    return u.d;
}

// support of D = hull - box along d:  w = h_i* - b(d),  b(d)_x = d_x > 0 ? -e_x : e_x
__device__ __forceinline__ void support(const Hull& H, const double* e, const double* d, SV& out, int lane) {
#pragma clang fp contract(off)
    double bk = -1.0e300, bx = 0.0, by = 0.0, bz = 0.0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
        if (64 * k >= H.n) break;                             // (H.n is wave-uniform: a scalar branch; the dish's 640 vertices fill 10 of the 18 slots)
        const int i = 64 * k + lane;
        const double key = (H.x[k] * d[0] + H.y[k] * d[1]) + H.z[k] * d[2];
        if (i < H.n && key > bk) { bk = key; bi = i; bx = H.x[k]; by = H.y[k]; bz = H.z[k]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ok = __shfl_xor(bk, off);
        const int oi = __shfl_xor(bi, off);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
    }
    const int src = __builtin_amdgcn_readfirstlane(bi) & 63;
    out.a[0] = rdlane(bx, src); out.a[1] = rdlane(by, src); out.a[2] = rdlane(bz, src);
#pragma unroll
    for (int x = 0; x < 3; ++x) out.w[x] = out.a[x] - (d[x] > 0.0 ? -e[x] : e[x]);
}

// Shape B as a second convex hull (round 6: object_balance's spinning_plate - the dish against the spool, oracle/narrowphase.c:
// mb_gjk_epa_hull_hull): its vertices in B's own frame, lane-spread like Hull's; its support point along -d is the vertex with the largest
// -d . b (ties: the lowest index).  gjk / epa / gjk_epa below take the box's half extents (const double*) or a const HullB*.
constexpr int kSlotsB = 4;                                    // n <= 256 (the spool's hull has 133 vertices)
struct HullB { double x[kSlotsB], y[kSlotsB], z[kSlotsB]; int n; };
__device__ __forceinline__ void support(const Hull& H, const HullB* B, const double* d, SV& out, int lane) {
#pragma clang fp contract(off)
    double bk = -1.0e300, bx = 0.0, by = 0.0, bz = 0.0, ck = -1.0e300, cx = 0.0, cy = 0.0, cz = 0.0;
    int bi = 0x7fffffff, ci = 0x7fffffff;
    const double nd[3] = {-d[0], -d[1], -d[2]};
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
        if (64 * k >= H.n) break;                             // (H.n is wave-uniform: a scalar branch; the dish's 640 vertices fill 10 of the 18 slots)
        const int i = 64 * k + lane;
        const double key = (H.x[k] * d[0] + H.y[k] * d[1]) + H.z[k] * d[2];
        if (i < H.n && key > bk) { bk = key; bi = i; bx = H.x[k]; by = H.y[k]; bz = H.z[k]; }
    }
#pragma unroll
    for (int k = 0; k < kSlotsB; ++k) {
        if (64 * k >= B->n) break;
        const int i = 64 * k + lane;
        const double key = (B->x[k] * nd[0] + B->y[k] * nd[1]) + B->z[k] * nd[2];
        if (i < B->n && key > ck) { ck = key; ci = i; cx = B->x[k]; cy = B->y[k]; cz = B->z[k]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ok = __shfl_xor(bk, off), pk = __shfl_xor(ck, off);
        const int oi = __shfl_xor(bi, off), pi = __shfl_xor(ci, off);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
        if (pk > ck || (pk == ck && pi < ci)) { ck = pk; ci = pi; }
    }
    const int sa = __builtin_amdgcn_readfirstlane(bi) & 63, sb = __builtin_amdgcn_readfirstlane(ci) & 63;
    out.a[0] = rdlane(bx, sa); out.a[1] = rdlane(by, sa); out.a[2] = rdlane(bz, sa);
    out.w[0] = out.a[0] - rdlane(cx, sb); out.w[1] = out.a[1] - rdlane(cy, sb); out.w[2] = out.a[2] - rdlane(cz, sb);
}

// ---- closest point of a simplex to the origin as barycentric weights (Ericson, Real-Time Collision Detection 5.1)
__device__ __forceinline__ void closest_segment(const double* a, const double* b, double* lam) {
#pragma clang fp contract(off)
    double ab[3]; sub3(b, a, ab);
    const double den = dot3(ab, ab);
    double t = den > 0.0 ? -dot3(a, ab) / den : 0.0;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    lam[0] = 1.0 - t; lam[1] = t;
}
__device__ __forceinline__ void closest_triangle(const double* a, const double* b, const double* c, double* lam) {
#pragma clang fp contract(off)
    double ab[3], ac[3]; sub3(b, a, ab); sub3(c, a, ac);
    const double d1 = -dot3(ab, a), d2 = -dot3(ac, a);
    lam[0] = lam[1] = lam[2] = 0.0;
    if (d1 <= 0.0 && d2 <= 0.0) { lam[0] = 1.0; return; }
    const double d3 = -dot3(ab, b), d4 = -dot3(ac, b);
    if (d3 >= 0.0 && d4 <= d3) { lam[1] = 1.0; return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { const double v = d1 / (d1 - d3); lam[0] = 1.0 - v; lam[1] = v; return; }
    const double d5 = -dot3(ab, c), d6 = -dot3(ac, c);
    if (d6 >= 0.0 && d5 <= d6) { lam[2] = 1.0; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { const double w = d2 / (d2 - d6); lam[0] = 1.0 - w; lam[2] = w; return; }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) { const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); lam[1] = 1.0 - w; lam[2] = w; return; }
    const double den = 1.0 / ((va + vb) + vc), v = vb * den, w = vc * den;
    lam[0] = (1.0 - v) - w; lam[1] = v; lam[2] = w;
}
__device__ __forceinline__ bool outside_plane(const double* a, const double* b, const double* c, const double* d) {
#pragma clang fp contract(off)
    double ab[3], ac[3], n[3], ad[3]; sub3(b, a, ab); sub3(c, a, ac); cross3(ab, ac, n); sub3(d, a, ad);
    const double so = -dot3(a, n), sd = dot3(ad, n);
    return sd == 0.0 || so * sd < 0.0;
}
// true: the origin is inside the tetrahedron
__device__ __forceinline__ bool closest_tetra(const SV* S, double* lam) {
#pragma clang fp contract(off)
    double best = 1e300; bool any = false;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int f0 = f == 3 ? 1 : 0, f1 = f == 0 ? 1 : (f == 1 ? 2 : 3), f2 = f == 0 ? 2 : (f == 1 ? 3 : (f == 2 ? 1 : 2)), o = f == 0 ? 3 : (f == 1 ? 1 : (f == 2 ? 2 : 0));
        if (!outside_plane(S[f0].w, S[f1].w, S[f2].w, S[o].w)) continue;
        double l3[3]; closest_triangle(S[f0].w, S[f1].w, S[f2].w, l3);
        double x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = (l3[0] * S[f0].w[k] + l3[1] * S[f1].w[k]) + l3[2] * S[f2].w[k];
        const double d2 = dot3(x, x);
        if (d2 < best) {
            best = d2; any = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) lam[k] = k == f0 ? l3[0] : (k == f1 ? l3[1] : (k == f2 ? l3[2] : 0.0));
        }
    }
    return !any;
}

// GJK: 0 separated (dist, n from the box to the hull, witnesses pa on the hull / pb on the box), 1 overlapping (S = a tetrahedron around the
// origin), 2 touching cores (no depth)
template <class BT> __device__ __forceinline__ int gjk(const Hull& H, BT e, SV* S, double& dist, double* nrm, double* pa, double* pb, int lane) {
#pragma clang fp contract(off)
    const double d0[3] = {1.0, 0.0, 0.0};
    int ns = 1;
    support(H, e, d0, S[0], lane);
    double x[3] = {S[0].w[0], S[0].w[1], S[0].w[2]}, lam[4] = {1.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < 64; ++it) {
        const double xx = dot3(x, x);
        if (xx <= 1e-28) { dist = 0.0; return 2; }
        const double d[3] = {-x[0], -x[1], -x[2]};
        SV w; support(H, e, d, w, lane);
        if (xx - dot3(x, w.w) <= 1e-12 * xx) break;
        bool dup = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < ns && S[k].w[0] == w.w[0] && S[k].w[1] == w.w[1] && S[k].w[2] == w.w[2]) dup = true;
        if (dup) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k == ns) S[k] = w;
        ++ns;
        if (ns == 2) closest_segment(S[0].w, S[1].w, lam);
        else if (ns == 3) closest_triangle(S[0].w, S[1].w, S[2].w, lam);
        else if (closest_tetra(S, lam)) { dist = 0.0; return 1; }
        // keep the vertices that carry the closest point (stable compaction, written without run-time register indexing)
        SV R[4] = {S[0], S[1], S[2], S[3]}; double rl[4] = {0.0, 0.0, 0.0, 0.0}; int m = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool keep = k < ns && lam[k] > 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (keep && j == m) { R[j] = S[k]; rl[j] = lam[k]; }
            m += keep ? 1 : 0;
        }
        ns = m;
#pragma unroll
        for (int k = 0; k < 4; ++k) { S[k] = R[k]; lam[k] = rl[k]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < ns) acc += lam[k] * S[k].w[c];
            x[c] = acc;
        }
    }
    const double len = sqrt(dot3(x, x));
    dist = len;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < ns) acc += lam[k] * S[k].a[c];
        pa[c] = acc; pb[c] = acc - x[c]; nrm[c] = x[c] / len;
    }
    return 0;
}

// ---- EPA on LDS scratch `sc` (kScratchWords doubles)
__device__ __forceinline__ void ld_v(lptr<double> sc, int i, double* w) { w[0] = sc[kOffV + 6 * i]; w[1] = sc[kOffV + 6 * i + 1]; w[2] = sc[kOffV + 6 * i + 2]; }
__device__ __forceinline__ bool make_face(lptr<double> sc, int i0, int i1, int i2, int f, int lane) {
#pragma clang fp contract(off)
    double p0[3], p1[3], p2[3], e1[3], e2[3], nn[3];
    ld_v(sc, i0, p0); ld_v(sc, i1, p1); ld_v(sc, i2, p2);
    sub3(p1, p0, e1); sub3(p2, p0, e2); cross3(e1, e2, nn);
    const double len = sqrt(dot3(nn, nn));
    if (!(len > 0.0)) return false;
    double n[3] = {nn[0] / len, nn[1] / len, nn[2] / len};
    double d = dot3(n, p0);
    int a1 = i1, a2 = i2;
    if (d < 0.0) { a1 = i2; a2 = i1; n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; d = -d; }
    if (lane == 0) {
        lptr<int> fv = (lptr<int>)(sc + kOffFV);
        sc[kOffFN + 4 * f] = n[0]; sc[kOffFN + 4 * f + 1] = n[1]; sc[kOffFN + 4 * f + 2] = n[2]; sc[kOffFN + 4 * f + 3] = d;
        fv[4 * f] = i0; fv[4 * f + 1] = a1; fv[4 * f + 2] = a2; fv[4 * f + 3] = 1;
    }
    __syncthreads();
    return true;
}
template <class BT> __device__ __forceinline__ bool epa(const Hull& H, BT e, const SV* S, lptr<double> sc, double& depth, double* nrm, double* pa, double* pb, int lane) {
#pragma clang fp contract(off)
    lptr<int> fv = (lptr<int>)(sc + kOffFV);
    lptr<int> ed = (lptr<int>)(sc + kOffE);
    int nv = 4, nf = 0;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) { sc[kOffV + 6 * k + c] = S[k].w[c]; sc[kOffV + 6 * k + 3 + c] = S[k].a[c]; }
    }
    __syncthreads();
    if (!make_face(sc, 0, 1, 2, nf++, lane)) return false;
    if (!make_face(sc, 0, 2, 3, nf++, lane)) return false;
    if (!make_face(sc, 0, 3, 1, nf++, lane)) return false;
    if (!make_face(sc, 1, 3, 2, nf++, lane)) return false;
    int best = 0;
    for (int it = 0; it < 64; ++it) {
        best = -1; double bd = 0.0;
        for (int f = 0; f < nf; ++f) {
            const double fd = sc[kOffFN + 4 * f + 3];
            if (fv[4 * f + 3] && (best < 0 || fd < bd)) { best = f; bd = fd; }
        }
        const double bn[3] = {sc[kOffFN + 4 * best], sc[kOffFN + 4 * best + 1], sc[kOffFN + 4 * best + 2]};
        SV w; support(H, e, bn, w, lane);
        if (dot3(bn, w.w) - bd <= 1e-12 || nv == kMaxV) break;
        int ne = 0;
        for (int f = 0; f < nf; ++f) {
            const double fn[3] = {sc[kOffFN + 4 * f], sc[kOffFN + 4 * f + 1], sc[kOffFN + 4 * f + 2]};
            if (!fv[4 * f + 3] || !(dot3(fn, w.w) - sc[kOffFN + 4 * f + 3] > 0.0)) continue;
            const int v0 = fv[4 * f], v1 = fv[4 * f + 1], v2 = fv[4 * f + 2];
            __syncthreads();
            if (lane == 0) fv[4 * f + 3] = 0;
            for (int k = 0; k < 3; ++k) {
                const int ea = k == 0 ? v0 : (k == 1 ? v1 : v2), eb = k == 0 ? v1 : (k == 1 ? v2 : v0);
                int hit = -1;
                for (int q = 0; q < ne; ++q) if (ed[2 * q] == eb && ed[2 * q + 1] == ea) hit = q;
                __syncthreads();
                if (hit >= 0) {
                    if (lane == 0) { ed[2 * hit] = ed[2 * (ne - 1)]; ed[2 * hit + 1] = ed[2 * (ne - 1) + 1]; }
                    --ne;
                } else if (ne < kMaxE) {
                    if (lane == 0) { ed[2 * ne] = ea; ed[2 * ne + 1] = eb; }
                    ++ne;
                }
                __syncthreads();
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { sc[kOffV + 6 * nv + c] = w.w[c]; sc[kOffV + 6 * nv + 3 + c] = w.a[c]; }
        }
        __syncthreads();
        for (int q = 0; q < ne && nf < kMaxF; ++q) if (make_face(sc, ed[2 * q], ed[2 * q + 1], nv, nf, lane)) ++nf;
        ++nv;
    }
    const int i0 = fv[4 * best], i1 = fv[4 * best + 1], i2 = fv[4 * best + 2];
    const double fn[3] = {sc[kOffFN + 4 * best], sc[kOffFN + 4 * best + 1], sc[kOffFN + 4 * best + 2]}, fd = sc[kOffFN + 4 * best + 3];
    double w0[3], w1[3], w2[3], p0[3], p1[3], p2[3], lam[3];
    ld_v(sc, i0, w0); ld_v(sc, i1, w1); ld_v(sc, i2, w2);
#pragma unroll
    for (int c = 0; c < 3; ++c) { p0[c] = w0[c] - fd * fn[c]; p1[c] = w1[c] - fd * fn[c]; p2[c] = w2[c] - fd * fn[c]; }
    closest_triangle(p0, p1, p2, lam);
    depth = fd;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        nrm[c] = -fn[c];
        pa[c] = (lam[0] * sc[kOffV + 6 * i0 + 3 + c] + lam[1] * sc[kOffV + 6 * i1 + 3 + c]) + lam[2] * sc[kOffV + 6 * i2 + 3 + c];
        pb[c] = pa[c] - ((lam[0] * w0[c] + lam[1] * w1[c]) + lam[2] * w2[c]);
    }
    return true;
}

// signed core distance (< 0: overlap depth), normal from the box towards the hull, witnesses; false: touching cores (no contact normal)
template <class BT> __device__ __forceinline__ bool gjk_epa(const Hull& H, BT e, lptr<double> sc, double& sdist, double* nrm, double* pa, double* pb, int lane) {
    SV S[4]; double dist = 0.0;
    const int r = gjk(H, e, S, dist, nrm, pa, pb, lane);
    if (r == 0) { sdist = dist; return true; }
    if (r == 2) return false;
    double depth = 0.0;
    if (!epa(H, e, S, sc, depth, nrm, pa, pb, lane)) return false;
    sdist = -depth;
    return true;
}
__device__ __forceinline__ bool gjk_epa_hull_box(const Hull& H, const double* e, lptr<double> sc, double& sdist, double* nrm, double* pa, double* pb, int lane) {
    return gjk_epa<const double*>(H, e, sc, sdist, nrm, pa, pb, lane);
}
// the same for two hulls: H = body A's hull in body B's frame, B = body B's hull in its own frame; results in B's frame
__device__ __forceinline__ bool gjk_epa_hull_hull(const Hull& H, const HullB& B, lptr<double> sc, double& sdist, double* nrm, double* pa, double* pb, int lane) {
    return gjk_epa<const HullB*>(H, &B, sc, sdist, nrm, pa, pb, lane);
}

// ---- persistent manifold in LDS (`mf`: kManiWords doubles; body A = the tip link (oa, Ra), body B = the cube (ob, Rb); R row-major)
__device__ __forceinline__ void to_world(const double* o, const double* R, const double* l, double* w) {
#pragma clang fp contract(off)
#pragma unroll
    for (int c = 0; c < 3; ++c) w[c] = o[c] + ((R[3 * c] * l[0] + R[3 * c + 1] * l[1]) + R[3 * c + 2] * l[2]);
}
__device__ __forceinline__ void to_local(const double* o, const double* R, const double* w, double* l) {
#pragma clang fp contract(off)
    const double d[3] = {w[0] - o[0], w[1] - o[1], w[2] - o[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) l[c] = (R[c] * d[0] + R[3 + c] * d[1]) + R[6 + c] * d[2];
}
__device__ __forceinline__ void ld3(lptr<double> p, double* o) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
__device__ __forceinline__ double area3(const double* p, const double* a, const double* b, const double* c) {
#pragma clang fp contract(off)
    double u[3], v[3], x[3]; sub3(p, a, u); sub3(c, b, v); cross3(u, v, x);
    return dot3(x, x);
}
__device__ __forceinline__ void manifold_add(lptr<double> mf, double breaking, const double* oa, const double* Ra, const double* ob, const double* Rb,
                                          const double* pa_w, const double* pb_w, const double* n_w, double depth, int lane) {
#pragma clang fp contract(off)
    if (depth > breaking) return;
    double la[3], lb[3]; to_local(oa, Ra, pa_w, la); to_local(ob, Rb, pb_w, lb);
    int n = (int)mf[kMcount];
    int slot = -1; double shortest = breaking * breaking;
    for (int i = 0; i < n; ++i) {
        double c[3], df[3]; ld3(mf + kMla + 3 * i, c); sub3(c, la, df);
        const double d2 = dot3(df, df);
        if (d2 < shortest) { shortest = d2; slot = i; }
    }
    if (slot < 0) {
        if (n < 4) slot = n++;
        else {   // sortCachedPoints: the deepest stays, the rest by largest area
            double c0[3], c1[3], c2[3], c3[3];
            ld3(mf + kMla, c0); ld3(mf + kMla + 3, c1); ld3(mf + kMla + 6, c2); ld3(mf + kMla + 9, c3);
            int deepest = -1; double md = depth;
            for (int i = 0; i < 4; ++i) if (mf[kMdepth + i] < md) { deepest = i; md = mf[kMdepth + i]; }
            double r[4] = {0.0, 0.0, 0.0, 0.0};
            if (deepest != 0) r[0] = area3(la, c1, c2, c3);
            if (deepest != 1) r[1] = area3(la, c0, c2, c3);
            if (deepest != 2) r[2] = area3(la, c0, c1, c3);
            if (deepest != 3) r[3] = area3(la, c0, c1, c2);
            int bestq = 0; double bv = r[0];
            if (r[1] > bv) { bv = r[1]; bestq = 1; }
            if (r[2] > bv) { bv = r[2]; bestq = 2; }
            if (r[3] > bv) { bv = r[3]; bestq = 3; }
            slot = bestq;
        }
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mf[kMla + 3 * slot + c] = la[c]; mf[kMlb + 3 * slot + c] = lb[c]; mf[kMn + 3 * slot + c] = n_w[c];
            mf[kMpa + 3 * slot + c] = pa_w[c]; mf[kMpb + 3 * slot + c] = pb_w[c];
        }
        mf[kMdepth + slot] = depth;
        mf[kMcount] = (double)n;
    }
    __syncthreads();
}
__device__ __forceinline__ void manifold_refresh(lptr<double> mf, double breaking, const double* oa, const double* Ra, const double* ob, const double* Rb, int lane) {
#pragma clang fp contract(off)
    int n = (int)mf[kMcount];
    __syncthreads();
    for (int i = n - 1; i >= 0; --i) {
        double la[3], lb[3], nr[3], pa[3], pb[3], df[3];
        ld3(mf + kMla + 3 * i, la); ld3(mf + kMlb + 3 * i, lb); ld3(mf + kMn + 3 * i, nr);
        to_world(oa, Ra, la, pa); to_world(ob, Rb, lb, pb);
        sub3(pa, pb, df);
        const double dd = dot3(df, nr);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { mf[kMpa + 3 * i + c] = pa[c]; mf[kMpb + 3 * i + c] = pb[c]; }
            mf[kMdepth + i] = dd;
        }
    }
    __syncthreads();
    for (int i = n - 1; i >= 0; --i) {
        double nr[3], pa[3], pb[3];
        ld3(mf + kMn + 3 * i, nr); ld3(mf + kMpa + 3 * i, pa); ld3(mf + kMpb + 3 * i, pb);
        const double dd = mf[kMdepth + i];
        bool drop = !(dd <= breaking);
        if (!drop) {
            double proj[3], dr[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) proj[c] = pa[c] - nr[c] * dd;
            sub3(pb, proj, dr);
            drop = dot3(dr, dr) > breaking * breaking;
        }
        if (drop) {
            const int last = n - 1;
            __syncthreads();
            if (i != last && lane == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    mf[kMla + 3 * i + c] = mf[kMla + 3 * last + c]; mf[kMlb + 3 * i + c] = mf[kMlb + 3 * last + c]; mf[kMn + 3 * i + c] = mf[kMn + 3 * last + c];
                    mf[kMpa + 3 * i + c] = mf[kMpa + 3 * last + c]; mf[kMpb + 3 * i + c] = mf[kMpb + 3 * last + c];
                }
                mf[kMdepth + i] = mf[kMdepth + last];
            }
            n = last;
            __syncthreads();
        }
    }
    if (lane == 0) mf[kMcount] = (double)n;
    __syncthreads();
}

}  // namespace narrow
}  // namespace tg
