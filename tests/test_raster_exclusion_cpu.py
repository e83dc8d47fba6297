"""CPU restatement (numpy float32, the kernels' own expression order) of the renders' edge-function block test - csrc/tg_raster.hip:
edges_exclude_rect, DESIGN 4.2 item 5 - against brute force over every pixel centre of the rectangle.  The device version is checked the
same way on the GPU (tests/test_gpu_parity.py::test_raster_edge_exclusion_never_hides_a_coverable_pixel); this one keeps the argument -
affine edge functions are extremal at the corners; the rounding of the pixel expression stays below m_i - inside the CPU suite."""
import numpy as np

F = np.float32


def _edges(x0, y0, x1, y1, x2, y2, fx, fy):
    a0, a1, a2 = y2 - fy, y1 - fy, y0 - fy
    e0 = (x1 - fx) * a0 - (x2 - fx) * a1
    e1 = (x2 - fx) * a2 - (x0 - fx) * a0
    e2 = (x0 - fx) * a1 - (x1 - fx) * a2
    return e0, e1, e2


def exclude(x0, y0, x1, y1, x2, y2, X0, X1, Y0, Y1):
    ab = np.abs
    B0, B1, B2 = np.maximum(ab(x0 - X0), ab(x0 - X1)), np.maximum(ab(x1 - X0), ab(x1 - X1)), np.maximum(ab(x2 - X0), ab(x2 - X1))
    A0, A1, A2 = np.maximum(ab(y2 - Y0), ab(y2 - Y1)), np.maximum(ab(y1 - Y0), ab(y1 - Y1)), np.maximum(ab(y0 - Y0), ab(y0 - Y1))
    m0, m1, m2 = F(1e-5) * (B1 * A0 + B2 * A1), F(1e-5) * (B2 * A2 + B0 * A0), F(1e-5) * (B0 * A1 + B1 * A2)
    hi = [np.full_like(x0, -3e38) for _ in range(3)]
    lo = [np.full_like(x0, 3e38) for _ in range(3)]
    s_lo, s_hi = np.full_like(x0, 3e38), np.full_like(x0, -3e38)
    for c in range(4):
        fx, fy = (X1 if c & 1 else X0), (Y1 if c & 2 else Y0)
        e = _edges(x0, y0, x1, y1, x2, y2, fx, fy)
        for i in range(3):
            hi[i], lo[i] = np.maximum(hi[i], e[i]), np.minimum(lo[i], e[i])
        sc = (e[0] + e[1]) + e[2]
        s_lo, s_hi = np.minimum(s_lo, sc), np.maximum(s_hi, sc)
    M2 = F(2) * ((m0 + m1) + m2)
    no_pos = (hi[0] < -F(2) * m0) | (hi[1] < -F(2) * m1) | (hi[2] < -F(2) * m2) | (s_hi < -M2)
    no_neg = (lo[0] > F(2) * m0) | (lo[1] > F(2) * m1) | (lo[2] > F(2) * m2) | (s_lo > M2)
    return no_pos & no_neg


def test_edge_exclusion_rule_against_brute_force():
    rng = np.random.default_rng(11)
    n = 60000
    kind = rng.integers(0, 5, n)
    span = np.where(kind == 2, 1e4, 356.0)
    off = np.where(kind == 2, -5e3, -50.0)
    v = [(off + rng.uniform(0, 1, n) * span) for _ in range(6)]
    x0, y0, x1, y1, x2, y2 = v
    t, eps = rng.uniform(-0.5, 1.5, n), rng.uniform(-1e-3, 1e-3, n)
    sl = kind == 1                                                  # slivers
    x2 = np.where(sl, x0 + t * (x1 - x0) + eps, x2); y2 = np.where(sl, y0 + t * (y1 - y0) - eps, y2)
    pc = kind == 3                                                  # vertices on pixel centres
    x0 = np.where(pc, np.floor(x0) + 0.5, x0); y0 = np.where(pc, np.floor(y0) + 0.5, y0); x1 = np.where(pc, np.floor(x1) + 0.5, x1)
    hf = kind == 4                                                  # heightfield-sized
    for arr, base in ((x1, x0), (y1, y0), (x2, x0), (y2, y0)):
        arr[hf] = base[hf] + rng.uniform(-24, 24, int(hf.sum()))
    x0, y0, x1, y1, x2, y2 = (a.astype(F) for a in (x0, y0, x1, y1, x2, y2))
    cell = rng.integers(0, 2, n).astype(bool)
    w, h = np.where(cell, 32, 16), np.where(cell, 8, 16)
    X0 = (rng.integers(0, 1 << 20, n) % (256 // w) * w + 0.5).astype(F); Y0 = (rng.integers(0, 1 << 20, n) % (256 // h) * h + 0.5).astype(F)
    X1, Y1 = (X0 + (w - 1)).astype(F), (Y0 + (h - 1)).astype(F)
    ex = exclude(x0, y0, x1, y1, x2, y2, X0, X1, Y0, Y1)
    covered = np.zeros(n, bool)
    for py in range(16):
        for px in range(32):
            inside = (px < w) & (py < h)
            e0, e1, e2 = _edges(x0, y0, x1, y1, x2, y2, (X0 + F(px)).astype(F), (Y0 + F(py)).astype(F))
            covered |= inside & (((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0)))
    assert not (ex & covered).any(), int((ex & covered).sum())
    empty = ~covered
    assert empty.sum() > n // 2 and ex.sum() > 0.9 * empty.sum(), (int(ex.sum()), int(empty.sum()))
