"""numpy statement of the area enclosure the 2D NMS decides most of its pairs with (stardist_amd/csrc/area_bounds.h): exact area of the
intersection of two integer polygons by boundary integration with exact integer predicates and a symbolic perturbation of the second
polygon by (eps, eps^2); `plain` = the polygon is simple once zero-length edges are dropped.  Test infrastructure: the GPU probe
(sd_area_bounds_pairs_device) is compared with it, and both with Clipper's area (oracle/_ref on the CPU, sd_clip_pairs_device on the GPU)."""
import numpy as np

# band weights (stardist_amd/csrc/area_bounds.h NEAR_W, STRIP_W): of an edge pair within one lattice step (round 5: 0.125) and of a STRIP
# = an edge with at least one near partner (round 6: along nearly coincident boundaries every edge is near about three edges of the other
# polygon, so 0.15 T already charges each strip 0.45; a pair with FEW near pairs -- each its own strip -- was charged a third of that)
NEAR_W = 0.15
STRIP_W = 0.45


def sgn(v):
    return np.sign(v).astype(np.int8)


def exact_area(PX, PY, QX, QY):
    """PX.. (N, n) int64 vertices of polygon P (clip) and Q (subject), Q perturbed by (eps, eps^2).
    Returns (area (N,) float64, crossings (N,), ok (N,) bool: same orientation)."""
    N, n = PX.shape
    ax, ay = PX[:, :, None], PY[:, :, None]                     # P vertex index on axis 1
    bx, by = np.roll(PX, -1, 1)[:, :, None], np.roll(PY, -1, 1)[:, :, None]
    cx, cy = QX[:, None, :], QY[:, None, :]                     # Q vertex index on axis 2
    dx, dy = np.roll(QX, -1, 1)[:, None, :], np.roll(QY, -1, 1)[:, None, :]
    ex, ey = bx - ax, by - ay                                   # edge e of P
    fx, fy = dx - cx, dy - cy                                   # edge f of Q
    # side of Q's vertices c (and d) against edge e, perturbed
    o_ec = ex * (cy - ay) - ey * (cx - ax)
    tie_e = np.where(ey != 0, -sgn(ey), sgn(ex))                # sign of cross(e, (eps, eps^2))
    s_ec = np.where(o_ec != 0, sgn(o_ec), tie_e)
    o_ed = ex * (dy - ay) - ey * (dx - ax)
    s_ed = np.where(o_ed != 0, sgn(o_ed), tie_e)
    # side of P's vertices a (and b) against edge f (both ends perturbed): orient(c, d, a - delta)
    o_fa = fx * (ay - cy) - fy * (ax - cx)
    tie_f = np.where(fy != 0, sgn(fy), -sgn(fx))
    s_fa = np.where(o_fa != 0, sgn(o_fa), tie_f)
    o_fb = fx * (by - cy) - fy * (bx - cx)
    s_fb = np.where(o_fb != 0, sgn(o_fb), tie_f)
    e_ok = (ex != 0) | (ey != 0)
    f_ok = (fx != 0) | (fy != 0)
    cross = e_ok & f_ok & (s_ec != s_ed) & (s_fa != s_fb)
    # signed areas (orientation)
    aP = (PX * np.roll(PY, -1, 1) - PY * np.roll(PX, -1, 1)).sum(1)
    aQ = (QX * np.roll(QY, -1, 1) - QY * np.roll(QX, -1, 1)).sum(1)
    sP, sQ = np.sign(aP), np.sign(aQ)
    ok = (sP == sQ) & (sP != 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(cross, o_fa / (o_fa - o_fb).astype(np.float64), 0.0)        # parameter on e
        u = np.where(cross, o_ec / (o_ec - o_ed).astype(np.float64), 0.0)        # parameter on f
    # point in polygon for every vertex (ray +x), perturbed
    # a in Q: straddle (cy < ay) != (dy < ay); right iff (s_fa > 0) == (dy > cy)
    st = f_ok & ((cy < ay) != (dy < ay))
    inQ = (st & ((s_fa > 0) == (dy > cy))).sum(2) & 1                               # (N, n) for each a
    # c in P: straddle (ay <= cy) != (by <= cy); right iff (s_ec > 0) == (by > ay)
    st2 = e_ok & ((ay <= cy) != (by <= cy))
    inP = (st2 & ((s_ec > 0) == (by > ay))).sum(1) & 1                              # (N, n) for each c
    # consistency of the propagation rule: in(b) = in(a) xor parity of crossings on e
    parE = cross.sum(2) & 1
    assert np.array_equal(np.roll(inQ, -1, 1), inQ ^ parE), "propagation P"
    parF = cross.sum(1) & 1
    assert np.array_equal(np.roll(inP, -1, 1), inP ^ parF), "propagation Q"
    lam = inQ + (np.where(cross, (sQ[:, None, None] * s_fb) * (1.0 - t), 0.0)).sum(2)
    mu = inP + (np.where(cross, (sP[:, None, None] * s_ed) * (1.0 - u), 0.0)).sum(1)
    cab = (PX * np.roll(PY, -1, 1) - PY * np.roll(PX, -1, 1)).astype(np.float64)
    ccd = (QX * np.roll(QY, -1, 1) - QY * np.roll(QX, -1, 1)).astype(np.float64)
    area = 0.5 * np.abs((cab * lam).sum(1) + (ccd * mu).sum(1))
    return area, cross.sum((1, 2)), ok, lam, mu


def plain(X, Y):
    """(N, n) -> bool: the polygon is simple after dropping zero-length edges (no two edges share a point except neighbours at their
    common vertex, no fold-back)."""
    N, n = X.shape
    out = np.ones(N, bool)
    for i in range(N):
        x, y = X[i], Y[i]
        keep = np.ones(n, bool)
        for k in range(n):
            if x[k] == x[k - 1] and y[k] == y[k - 1]: keep[k] = False
        x, y = x[keep], y[keep]
        m = len(x)
        if m < 3: out[i] = False; continue
        ax, ay = x[:, None], y[:, None]; bx, by = np.roll(x, -1)[:, None], np.roll(y, -1)[:, None]
        cx, cy = x[None, :], y[None, :]; dx, dy = np.roll(x, -1)[None, :], np.roll(y, -1)[None, :]
        o1 = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax); o2 = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax)
        o3 = (dx - cx) * (ay - cy) - (dy - cy) * (ax - cx); o4 = (dx - cx) * (by - cy) - (dy - cy) * (bx - cx)
        inter = (np.sign(o1) * np.sign(o2) <= 0) & (np.sign(o3) * np.sign(o4) <= 0)
        # collinear: need interval overlap
        col = (o1 == 0) & (o2 == 0)
        ov = (np.maximum(np.minimum(ax, bx), np.minimum(cx, dx)) <= np.minimum(np.maximum(ax, bx), np.maximum(cx, dx))) & \
             (np.maximum(np.minimum(ay, by), np.minimum(cy, dy)) <= np.minimum(np.maximum(ay, by), np.maximum(cy, dy)))
        inter = np.where(col, ov, inter)
        k = np.arange(m)
        adj = (k[:, None] == k[None, :]) | ((k[:, None] + 1) % m == k[None, :]) | ((k[None, :] + 1) % m == k[:, None])
        if (inter & ~adj).any(): out[i] = False; continue
        # neighbours: fold-back
        ex, ey = np.roll(x, -1) - x, np.roll(y, -1) - y
        fx, fy = np.roll(ex, -1), np.roll(ey, -1)
        if ((ex * fy - ey * fx == 0) & (ex * fx + ey * fy < 0)).any(): out[i] = False; continue
        # robustly simple (round 5): no vertex (of the ORIGINAL list, duplicates included) within HALF a lattice step, along its scan line, of an
        # edge it is not an end point of
        vx, vy = X[i][:, None], Y[i][:, None]                               # vertices on axis 0
        ax, ay = x[None, :], y[None, :]; bx, by = np.roll(x, -1)[None, :], np.roll(y, -1)[None, :]   # edges (non-degenerate by construction) on axis 1
        endp = ((ax == vx) & (ay == vy)) | ((bx == vx) & (by == vy))
        inr = (vy >= np.minimum(ay, by)) & (vy <= np.maximum(ay, by))
        hor = ay == by
        risk_h = hor & (vx >= np.minimum(ax, bx)) & (vx <= np.maximum(ax, bx))
        num = (ax - vx) * (by - ay) + (vy - ay) * (bx - ax)
        risk_s = (~hor) & (2 * np.abs(num) <= np.abs(by - ay))                       # |x_edge(vy) - vx| <= 1/2: the rounded abscissae tie
        if (~endp & inr & (risk_h | risk_s)).any(): out[i] = False
    return out


def band(K, T, lmaxP, lmaxQ, ext, perimP, perimQ, S=None):
    """half-width of the enclosure of Clipper's area around the exact one (area_bounds.h); T = near edge pairs, S = strips (near_strips;
    None: the T term alone)"""
    near = NEAR_W * T if S is None else np.maximum(NEAR_W * T, STRIP_W * S)
    return (0.5 * K + near) * (lmaxP + lmaxQ) + 0.75 + 2e-6 * ext * (perimP + perimQ)


def _near_matrix(PX, PY, QX, QY):
    ax, ay = PX[:, :, None], PY[:, :, None]; bx, by = np.roll(PX, -1, 1)[:, :, None], np.roll(PY, -1, 1)[:, :, None]
    cx, cy = QX[:, None, :], QY[:, None, :]; dx, dy = np.roll(QX, -1, 1)[:, None, :], np.roll(QY, -1, 1)[:, None, :]
    ox = np.maximum(np.minimum(ax, bx), np.minimum(cx, dx)) - np.minimum(np.maximum(ax, bx), np.maximum(cx, dx))
    oy = np.maximum(np.minimum(ay, by), np.minimum(cy, dy)) - np.minimum(np.maximum(ay, by), np.maximum(cy, dy))
    ok = ((ax != bx) | (ay != by)) & ((cx != dx) | (cy != dy))
    return ok & (ox <= 1) & (oy <= 1)


def near_strips(PX, PY, QX, QY):
    """number of strips: max over the two polygons of the number of edges with at least one near partner (near_pairs' criterion)"""
    m = _near_matrix(PX, PY, QX, QY)
    return np.maximum(m.any(2).sum(1), m.any(1).sum(1))


def near_pairs(PX, PY, QX, QY):
    """number of edge pairs (e of P, f of Q), both of non-zero length, whose bounding boxes come within one lattice step"""
    ax, ay = PX[:, :, None], PY[:, :, None]; bx, by = np.roll(PX, -1, 1)[:, :, None], np.roll(PY, -1, 1)[:, :, None]
    cx, cy = QX[:, None, :], QY[:, None, :]; dx, dy = np.roll(QX, -1, 1)[:, None, :], np.roll(QY, -1, 1)[:, None, :]
    ox = np.maximum(np.minimum(ax, bx), np.minimum(cx, dx)) - np.minimum(np.maximum(ax, bx), np.maximum(cx, dx))
    oy = np.maximum(np.minimum(ay, by), np.minimum(cy, dy)) - np.minimum(np.maximum(ay, by), np.maximum(cy, dy))
    ok = ((ax != bx) | (ay != by)) & ((cx != dx) | (cy != dy))
    return (ok & (ox <= 1) & (oy <= 1)).sum((1, 2))


def edge_stats(X, Y):
    """(longest edge, L1 perimeter) per polygon"""
    ex, ey = np.roll(X, -1, 1) - X, np.roll(Y, -1, 1) - Y
    return np.sqrt((ex * ex + ey * ey).astype(np.float64)).max(1), (np.abs(ex) + np.abs(ey)).sum(1).astype(np.float64)
