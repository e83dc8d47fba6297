/* narrowphase.c - TEST INFRASTRUCTURE (part of the CPU oracle, see minibullet.h): a general convex narrowphase for the tip core - cube pair of
 * object_push and Bullet's persistent-manifold policy on top of it, behind mb_push_scene.narrowphase = 1.
 *
 * What it restates.  north_star names "narrowphase GJK/EPA on the meshes" for pb.stepSimulation() (call site robots/arms/robot.py:141; the pair
 * is set up by object_push_env.py:216-225 and sensors/tactile_sensor.py:322-332).  Bullet's source is not in /root/reference (pybullet is an
 * unpinned wheel, requirements.txt:6), so this is the published algorithms - Gilbert, Johnson & Keerthi 1988 (distance), van den Bergen 2001
 * (expanding polytope) - and btPersistentManifold's cache rules as remembered (PARITY_ASSUMPTIONS A35-A38, all "unverified").  PARITY UNPINNED.
 *
 * Geometry model (unchanged from the closed form, A24): core shapes = the tip core's convex hull (its vertices) and the exact box; a contact
 * exists when  dist(cores) - margin_tip - margin_cube <= breaking;  the points handed to the solver sit on the margin-inflated surfaces.
 * Everything below works in the BOX frame (box axis-aligned at the origin): the Minkowski difference D = hull - box has the support
 *     w(d) = h_i* - b(d),   i* = argmax_i h_i . d (ties: the lowest index),   b(d)_x = d_x > 0 ? -e_x : +e_x   (the box's support along -d).
 * Compiled with -ffp-contract=off; the device restatement (tg_contact_wave.hip, `#pragma clang fp contract(off)` in its narrowphase) follows
 * the same operation order so that the two agree to the last bit on identical inputs. */
#include <math.h>
#include <string.h>

#include "minibullet.h"

typedef struct { double w[3], a[3]; } sv_t;     /* a vertex of D and the hull point it came from (the box point is a - w) */

static double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }

/* shape B: the box (half extents e) or, since round 6, a second convex hull hb [nb][3] in its own frame (object_balance's spinning_plate:
 * the dish on the spool, both btConvexHullShape): its support point along -d is the vertex with the largest -d . b (ties: the lowest index) */
typedef struct { const double* e; const double* hb; int nb; } shape_b;
static void support(const double* hull, int n, const shape_b* B, const double* d, sv_t* out) {
    int best = 0; double bk = dot3(hull, d);
    for (int i = 1; i < n; ++i) { const double k = dot3(hull + 3 * i, d); if (k > bk) { bk = k; best = i; } }
    if (B->hb == NULL) {
        const double* e = B->e;
        for (int x = 0; x < 3; ++x) { out->a[x] = hull[3 * best + x]; out->w[x] = out->a[x] - (d[x] > 0.0 ? -e[x] : e[x]); }
        return;
    }
    const double nd[3] = {-d[0], -d[1], -d[2]};
    int bb = 0; double bkb = dot3(B->hb, nd);
    for (int i = 1; i < B->nb; ++i) { const double k = dot3(B->hb + 3 * i, nd); if (k > bkb) { bkb = k; bb = i; } }
    for (int x = 0; x < 3; ++x) { out->a[x] = hull[3 * best + x]; out->w[x] = out->a[x] - B->hb[3 * bb + x]; }
}

/* ---- closest point of a simplex to the origin, with barycentric weights (Ericson, Real-Time Collision Detection 5.1: Voronoi regions) */
static void closest_segment(const double* a, const double* b, double* lam) {
    double ab[3]; sub3(b, a, ab);
    const double den = dot3(ab, ab);
    double t = den > 0.0 ? -dot3(a, ab) / den : 0.0;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    lam[0] = 1.0 - t; lam[1] = t;
}
static void closest_triangle(const double* a, const double* b, const double* c, double* lam) {
    double ab[3], ac[3]; sub3(b, a, ab); sub3(c, a, ac);
    const double d1 = -dot3(ab, a), d2 = -dot3(ac, a);                       /* ap = -a */
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
/* origin strictly on the other side of plane (a, b, c) from d? */
static int outside_plane(const double* a, const double* b, const double* c, const double* d) {
    double ab[3], ac[3], n[3], ad[3]; sub3(b, a, ab); sub3(c, a, ac); cross3(ab, ac, n); sub3(d, a, ad);
    const double so = -dot3(a, n), sd = dot3(ad, n);
    return sd == 0.0 || so * sd < 0.0;                                         /* a flat tetrahedron: every face is looked at */
}
/* returns 1 when the origin is inside the tetrahedron (lam untouched) */
static int closest_tetra(const double p[4][3], double* lam) {
    static const int F[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 1}, {1, 3, 2}};
    static const int O[4] = {3, 1, 2, 0};
    double best = 1e300; int any = 0;
    for (int f = 0; f < 4; ++f) {
        if (!outside_plane(p[F[f][0]], p[F[f][1]], p[F[f][2]], p[O[f]])) continue;
        double l3[3]; closest_triangle(p[F[f][0]], p[F[f][1]], p[F[f][2]], l3);
        double x[3];
        for (int k = 0; k < 3; ++k) x[k] = (l3[0] * p[F[f][0]][k] + l3[1] * p[F[f][1]][k]) + l3[2] * p[F[f][2]][k];
        const double d2 = dot3(x, x);
        if (d2 < best) { best = d2; any = 1; lam[0] = lam[1] = lam[2] = lam[3] = 0.0; lam[F[f][0]] = l3[0]; lam[F[f][1]] = l3[1]; lam[F[f][2]] = l3[2]; }
    }
    return !any;
}

/* ---- GJK.  Returns 0 separated (dist > 0, n from the box to the hull, a / b the witness points on the hull / the box), 1 overlapping with a
 * tetrahedron around the origin in S (EPA continues from it), 2 touching (dist 0 with a lower-dimensional simplex: reported as no contact depth). */
static int gjk(const double* hull, int n, const shape_b* e, sv_t S[4], int* ns_out, double* dist, double* nrm, double* pa, double* pb) {
    const double d0[3] = {1.0, 0.0, 0.0};
    int ns = 1;
    support(hull, n, e, d0, &S[0]);
    double x[3] = {S[0].w[0], S[0].w[1], S[0].w[2]}, lam[4] = {1.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < 64; ++it) {
        const double xx = dot3(x, x);
        if (xx <= 1e-28) { *ns_out = ns; *dist = 0.0; return 2; }
        const double d[3] = {-x[0], -x[1], -x[2]};
        sv_t w; support(hull, n, e, d, &w);
        if (xx - dot3(x, w.w) <= 1e-12 * xx) break;                           /* no vertex of D is closer along -x: x is the closest point */
        int dup = 0;
        for (int k = 0; k < ns; ++k) if (S[k].w[0] == w.w[0] && S[k].w[1] == w.w[1] && S[k].w[2] == w.w[2]) dup = 1;
        if (dup) break;
        S[ns++] = w;
        if (ns == 2) closest_segment(S[0].w, S[1].w, lam);
        else if (ns == 3) closest_triangle(S[0].w, S[1].w, S[2].w, lam);
        else {
            double p[4][3];
            for (int k = 0; k < 4; ++k) memcpy(p[k], S[k].w, sizeof p[k]);
            if (closest_tetra(p, lam)) { *ns_out = 4; *dist = 0.0; return 1; }
        }
        int m = 0;                                                             /* keep the vertices that carry the closest point */
        for (int k = 0; k < ns; ++k) if (lam[k] > 0.0) { S[m] = S[k]; lam[m] = lam[k]; ++m; }
        ns = m;
        for (int c = 0; c < 3; ++c) { double acc = 0.0; for (int k = 0; k < ns; ++k) acc += lam[k] * S[k].w[c]; x[c] = acc; }
    }
    const double len = sqrt(dot3(x, x));
    *ns_out = ns; *dist = len;
    for (int c = 0; c < 3; ++c) {
        double acc = 0.0; for (int k = 0; k < ns; ++k) acc += lam[k] * S[k].a[c];
        pa[c] = acc; pb[c] = acc - x[c]; nrm[c] = x[c] / len;
    }
    return 0;
}

/* ---- EPA from GJK's tetrahedron.  depth > 0: how far the cores overlap along n (from the box to the hull: moving the hull by depth n separates them). */
enum { EPA_MAXV = 48, EPA_MAXF = 96, EPA_MAXE = 48 };   /* the device's LDS capacities (tg_narrowphase.hpp) */
typedef struct { int v[3]; double n[3], d; int alive; } face_t;
static int make_face(const sv_t* V, int i0, int i1, int i2, face_t* f) {
    double e1[3], e2[3], nn[3]; sub3(V[i1].w, V[i0].w, e1); sub3(V[i2].w, V[i0].w, e2); cross3(e1, e2, nn);
    const double len = sqrt(dot3(nn, nn));
    if (!(len > 0.0)) return 0;
    f->v[0] = i0; f->v[1] = i1; f->v[2] = i2;
    for (int c = 0; c < 3; ++c) f->n[c] = nn[c] / len;
    f->d = dot3(f->n, V[i0].w);
    if (f->d < 0.0) { f->v[1] = i2; f->v[2] = i1; for (int c = 0; c < 3; ++c) f->n[c] = -f->n[c]; f->d = -f->d; }   /* outward: away from the origin inside */
    f->alive = 1;
    return 1;
}
static int epa(const double* hull, int n, const shape_b* e, const sv_t S[4], double* depth, double* nrm, double* pa, double* pb) {
    static sv_t V[EPA_MAXV]; static face_t F[EPA_MAXF];   /* single-threaded test infrastructure */
    int nv = 4, nf = 0;
    for (int k = 0; k < 4; ++k) V[k] = S[k];
    static const int T[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 1}, {1, 3, 2}};
    for (int f = 0; f < 4; ++f) if (!make_face(V, T[f][0], T[f][1], T[f][2], &F[nf++])) return 0;
    int best = 0;
    for (int it = 0; it < 64; ++it) {
        best = -1;
        for (int f = 0; f < nf; ++f) if (F[f].alive && (best < 0 || F[f].d < F[best].d)) best = f;
        sv_t w; support(hull, n, e, F[best].n, &w);
        if (dot3(F[best].n, w.w) - F[best].d <= 1e-12 || nv == EPA_MAXV) break;           /* the face lies on D's boundary */
        /* faces seen from w go; the horizon = their edges whose reverse is not an edge of another removed face */
        int E[EPA_MAXE][2], ne = 0;
        for (int f = 0; f < nf; ++f) {
            if (!F[f].alive || !(dot3(F[f].n, w.w) - F[f].d > 0.0)) continue;
            F[f].alive = 0;
            for (int k = 0; k < 3; ++k) {
                const int ea = F[f].v[k], eb = F[f].v[(k + 1) % 3];
                int hit = -1;
                for (int q = 0; q < ne; ++q) if (E[q][0] == eb && E[q][1] == ea) hit = q;
                if (hit >= 0) { E[hit][0] = E[ne - 1][0]; E[hit][1] = E[ne - 1][1]; --ne; }
                else if (ne < EPA_MAXE) { E[ne][0] = ea; E[ne][1] = eb; ++ne; }
            }
        }
        V[nv] = w;
        for (int q = 0; q < ne && nf < EPA_MAXF; ++q) if (make_face(V, E[q][0], E[q][1], nv, &F[nf])) ++nf;
        ++nv;
    }
    /* witness points: the origin's projection dn lies in the closest face; its barycentric weights carry over to the hull points */
    const face_t* f = &F[best];
    double p0[3], p1[3], p2[3], lam[3];
    for (int c = 0; c < 3; ++c) { p0[c] = V[f->v[0]].w[c] - f->d * f->n[c]; p1[c] = V[f->v[1]].w[c] - f->d * f->n[c]; p2[c] = V[f->v[2]].w[c] - f->d * f->n[c]; }
    closest_triangle(p0, p1, p2, lam);
    *depth = f->d;
    for (int c = 0; c < 3; ++c) {
        nrm[c] = -f->n[c];                                                                 /* D = hull - box: the hull leaves along -n_face */
        pa[c] = (lam[0] * V[f->v[0]].a[c] + lam[1] * V[f->v[1]].a[c]) + lam[2] * V[f->v[2]].a[c];
        pb[c] = pa[c] - ((lam[0] * V[f->v[0]].w[c] + lam[1] * V[f->v[1]].w[c]) + lam[2] * V[f->v[2]].w[c]);
    }
    return 1;
}

/* Signed distance of the cores (negative: overlap depth), the unit normal from the box towards the hull and the witness points on the hull
 * (pa) and on the box (pb), all in the box frame.  hull: [n][3] in the box frame.  Returns 0 when no normal exists (touching cores). */
static int gjk_epa(const double* hull, int n, const shape_b* B, double* sdist, double* nrm, double* pa, double* pb) {
    sv_t S[4]; int ns = 0; double dist = 0.0;
    const int r = gjk(hull, n, B, S, &ns, &dist, nrm, pa, pb);
    if (r == 0) { *sdist = dist; return 1; }
    if (r == 2) return 0;
    double depth = 0.0;
    if (!epa(hull, n, B, S, &depth, nrm, pa, pb)) return 0;
    *sdist = -depth;
    return 1;
}
int mb_gjk_epa_hull_box(const double* hull, int n, const double* half, double* sdist, double* nrm, double* pa, double* pb) {
    const shape_b B = {half, NULL, 0};
    return gjk_epa(hull, n, &B, sdist, nrm, pa, pb);
}
/* The same for two hulls: `hull` [n][3] is body A's, given in body B's frame; `hull_b` [nb][3] is body B's in its own frame.  sdist, the unit
 * normal from B towards A and the witness points (pa on A, pb on B) come back in B's frame. */
int mb_gjk_epa_hull_hull(const double* hull, int n, const double* hull_b, int nb, double* sdist, double* nrm, double* pa, double* pb) {
    const shape_b B = {NULL, hull_b, nb};
    return gjk_epa(hull, n, &B, sdist, nrm, pa, pb);
}

/* ---- btPersistentManifold's cache (A36-A38).  Local points: la in the tip link's frame, lb in the cube's frame (body A = the tip: the
 * normal points from the cube to the tip, as in the closed form). */
static void to_world(const double* o, const double* R, const double* l, double* w) {
    for (int c = 0; c < 3; ++c) w[c] = o[c] + ((R[3 * c] * l[0] + R[3 * c + 1] * l[1]) + R[3 * c + 2] * l[2]);
}
static void to_local(const double* o, const double* R, const double* w, double* l) {
    const double d[3] = {w[0] - o[0], w[1] - o[1], w[2] - o[2]};
    for (int c = 0; c < 3; ++c) l[c] = (R[c] * d[0] + R[3 + c] * d[1]) + R[6 + c] * d[2];
}
static double area3(const double* p, const double* a, const double* b, const double* c) {   /* |(p - a) x (c - b)|^2 */
    double u[3], v[3], x[3]; sub3(p, a, u); sub3(c, b, v); cross3(u, v, x);
    return dot3(x, x);
}
/* sortCachedPoints: which of the four cached points the new one replaces - the deepest stays, the rest by largest area */
static int sort_cached(const mb_manifold* m, const double* la_new, double depth_new) {
    int deepest = -1; double md = depth_new;
    for (int i = 0; i < 4; ++i) if (m->depth[i] < md) { deepest = i; md = m->depth[i]; }
    double r[4] = {0.0, 0.0, 0.0, 0.0};
    if (deepest != 0) r[0] = area3(la_new, m->la[1], m->la[2], m->la[3]);
    if (deepest != 1) r[1] = area3(la_new, m->la[0], m->la[2], m->la[3]);
    if (deepest != 2) r[2] = area3(la_new, m->la[0], m->la[1], m->la[3]);
    if (deepest != 3) r[3] = area3(la_new, m->la[0], m->la[1], m->la[2]);
    int best = 0;                                                                          /* btVector4::closestAxis4 of the absolute values */
    for (int i = 1; i < 4; ++i) if (r[i] > r[best]) best = i;
    return best;
}
void mb_manifold_add(mb_manifold* m, double breaking, const double* oa, const double* Ra, const double* ob, const double* Rb,
                     const double* pa_w, const double* pb_w, const double* n_w, double depth) {
    if (depth > breaking) return;                                                          /* btManifoldResult::addContactPoint */
    double la[3], lb[3]; to_local(oa, Ra, pa_w, la); to_local(ob, Rb, pb_w, lb);
    int slot = -1; double shortest = breaking * breaking;                                  /* getCacheEntry: nearest cached point within the threshold */
    for (int i = 0; i < m->n; ++i) {
        double df[3]; sub3(m->la[i], la, df);
        const double d2 = dot3(df, df);
        if (d2 < shortest) { shortest = d2; slot = i; }
    }
    if (slot < 0) slot = m->n < 4 ? m->n++ : sort_cached(m, la, depth);                    /* addManifoldPoint */
    memcpy(m->la[slot], la, sizeof la); memcpy(m->lb[slot], lb, sizeof lb); memcpy(m->nrm[slot], n_w, sizeof la);
    memcpy(m->pa[slot], pa_w, sizeof la); memcpy(m->pb[slot], pb_w, sizeof la);
    m->depth[slot] = depth;
}
void mb_manifold_refresh(mb_manifold* m, double breaking, const double* oa, const double* Ra, const double* ob, const double* Rb) {
    for (int i = m->n - 1; i >= 0; --i) {                                                  /* refreshContactPoints: world positions, distance ... */
        to_world(oa, Ra, m->la[i], m->pa[i]); to_world(ob, Rb, m->lb[i], m->pb[i]);
        double df[3]; sub3(m->pa[i], m->pb[i], df);
        m->depth[i] = dot3(df, m->nrm[i]);
    }
    for (int i = m->n - 1; i >= 0; --i) {                                                  /* ... then drop what has separated or drifted sideways */
        int drop = !(m->depth[i] <= breaking);
        if (!drop) {
            double proj[3], dr[3];
            for (int c = 0; c < 3; ++c) proj[c] = m->pa[i][c] - m->nrm[i][c] * m->depth[i];
            sub3(m->pb[i], proj, dr);
            drop = dot3(dr, dr) > breaking * breaking;
        }
        if (drop) {                                                                        /* removeContactPoint: the last point takes the slot */
            const int last = m->n - 1;
            if (i != last) {
                memcpy(m->la[i], m->la[last], sizeof m->la[i]); memcpy(m->lb[i], m->lb[last], sizeof m->lb[i]);
                memcpy(m->nrm[i], m->nrm[last], sizeof m->nrm[i]); memcpy(m->pa[i], m->pa[last], sizeof m->pa[i]);
                memcpy(m->pb[i], m->pb[last], sizeof m->pb[i]); m->depth[i] = m->depth[last];
            }
            m->n = last;
        }
    }
}
