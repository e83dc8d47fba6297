"""General convex narrowphase (GJK distance + EPA penetration) in numpy - TEST INFRASTRUCTURE, like the rest of oracle/.

north_star names GJK/EPA as the reference's narrowphase (Bullet's btGjkPairDetector / btGjkEpaPenetrationDepthSolver; no Bullet source is in
/root/reference, this is the published algorithm: Gilbert-Johnson-Keerthi 1988, van den Bergen 2001).  The product and oracle/minibullet.c
generate contacts with closed forms for the four shape pairs the envs can touch (PARITY_ASSUMPTIONS A24, A30); tests/test_oracle_known_answers.py
uses this module to show that on the states the envs visit those closed forms return the distance / depth and normal a general convex
routine returns for the same shapes.  Shapes are support functions: d -> the point of the shape furthest along d."""
import numpy as np


def hull_support(verts):
    verts = np.asarray(verts, dtype=float)
    return lambda d: verts[int(np.argmax(verts @ d))]


def box_support(center, R, half):
    center, R, half = np.asarray(center, float), np.asarray(R, float), np.asarray(half, float)
    return lambda d: center + R @ (np.where(R.T @ d >= 0, 1.0, -1.0) * half)


def sphere_support(center, radius):
    center = np.asarray(center, float)
    return lambda d: center + radius * d / np.linalg.norm(d)


def cylinder_support(center, R, half_len, radius):
    """Solid cylinder, axis = the frame's z."""
    center, R = np.asarray(center, float), np.asarray(R, float)

    def sup(d):
        l = R.T @ d
        r = np.hypot(l[0], l[1])
        p = np.array([radius * l[0] / r if r > 0 else 0.0, radius * l[1] / r if r > 0 else 0.0, half_len if l[2] >= 0 else -half_len])
        return center + R @ p
    return sup


def _closest_on_simplex(pts):
    """Point of conv(pts) (1-4 points) closest to the origin -> (point, barycentric weights) by enumeration of the faces' Voronoi regions."""
    pts = [np.asarray(p, float) for p in pts]
    n = len(pts)
    best = None
    import itertools
    for k in range(1, n + 1):
        for idx in itertools.combinations(range(n), k):
            P = np.array([pts[i] for i in idx])
            if k == 1:
                w = np.array([1.0])
            else:   # minimise |sum w_i p_i|^2 with sum w = 1: solve the affine-hull projection
                A = P[1:] - P[0]
                G = A @ A.T
                try:
                    t = np.linalg.solve(G, -(A @ P[0]))
                except np.linalg.LinAlgError:
                    continue
                w = np.concatenate([[1.0 - t.sum()], t])
                if (w < -1e-14).any():
                    continue            # the projection falls outside this face
            x = w @ P
            d2 = float(x @ x)
            if best is None or d2 < best[0] - 1e-30:
                full = np.zeros(n)
                full[list(idx)] = w
                best = (d2, x, full)
    return best[1], best[2]


def gjk(sup_a, sup_b, iters=64, tol=1e-12):
    """Distance between two convex shapes -> (distance, witness on A, witness on B, simplex of (w, a, b) triples).  distance == 0: they overlap."""
    d = np.array([1.0, 0.0, 0.0])
    a, b = sup_a(d), sup_b(-d)
    simplex = [(a - b, a, b)]
    x = simplex[0][0]
    for _ in range(iters):
        dist = np.linalg.norm(x)
        if dist < tol:
            return 0.0, None, None, simplex
        d = -x / dist
        a, b = sup_a(d), sup_b(-d)
        w = a - b
        if dist - (-(w @ d)) < tol * max(1.0, dist):      # no progress possible along d: x is the closest point of the difference
            break
        simplex.append((w, a, b))
        x, lam = _closest_on_simplex([s[0] for s in simplex])
        simplex = [s for s, l in zip(simplex, lam) if l > 0]
        if len(simplex) == 4:
            return 0.0, None, None, simplex                # the origin is inside the tetrahedron
    x, lam = _closest_on_simplex([s[0] for s in simplex])
    pa = sum(l * s[1] for s, l in zip(simplex, lam))
    pb = sum(l * s[2] for s, l in zip(simplex, lam))
    return float(np.linalg.norm(x)), pa, pb, simplex


def epa(sup_a, sup_b, iters=128, tol=1e-10):
    """Penetration of two overlapping convex shapes -> (depth, unit normal from B to A ... the direction along which moving A by depth separates
    them).  Expanding polytope over the Minkowski difference, started from a tetrahedron of supports around the origin."""
    dirs = [np.array(v, float) for v in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
    pts = []
    for d in dirs:
        w = sup_a(d) - sup_b(-d)
        if not any(np.linalg.norm(w - p) < 1e-14 for p in pts):
            pts.append(w)
    from scipy.spatial import ConvexHull
    for _ in range(iters):
        hull = ConvexHull(np.array(pts))
        eq = hull.equations                                 # n . x + o <= 0 inside, |n| = 1
        k = int(np.argmin(-eq[:, 3]))                       # face closest to the origin (the origin is inside: offsets are negative)
        n, dist = eq[k, :3], -eq[k, 3]
        w = sup_a(n) - sup_b(-n)
        if w @ n - dist < tol:
            return float(dist), n
        pts.append(w)
    return float(dist), n
