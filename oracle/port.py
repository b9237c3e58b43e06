"""numpy restatement of the reference's Python glue on the hot path (TEST INFRASTRUCTURE ONLY).

The reference package cannot be imported here (csbdeep / scikit-image / TensorFlow are not
installed), so the pure-Python pieces between the network heads and the
natives are restated, each function citing the reference lines it follows.  The natives
themselves are NOT restated: they are the compiled reference in oracle/_ref (oracle.ref).

Parity status: `polygon` restates scikit-image's rule (skimage/draw/_draw.pyx `_polygon` +
skimage/_shared/geometry.pyx `point_in_polygon`, the OUTSIDE/INSIDE/VERTEX/EDGE version).
scikit-image is absent from /root/reference and from the default interpreter, but the image's
Anaconda python has scikit-image 0.18.3: tests/golden/make_raster2d_golden.py runs the
reference's geom2d functions on it, and `polygon` / `polygons_to_label*` reproduce those golden
label images bit for bit (tests/test_cpu_oracle.py) -- pinned.
"""
import numpy as np

from . import ref


# ----------------------------------------------------------------------------- nms.py
def ind_prob_thresh(prob, prob_thresh, b=2):
    """stardist/nms.py:6-17"""
    if b is not None and np.isscalar(b):
        b = ((b, b),) * prob.ndim
    ind_thresh = prob > prob_thresh
    if b is not None:
        _ind = np.zeros_like(ind_thresh)
        ss = tuple(slice(_bs[0] if _bs[0] > 0 else None, -_bs[1] if _bs[1] > 0 else None) for _bs in b)
        _ind[ss] = True
        ind_thresh &= _ind
    return ind_thresh


def _prep(x, dtype):
    return np.ascontiguousarray(x.astype(dtype, copy=False))


def non_maximum_suppression_inds(dist, points, scores=None, thresh=0.5, use_bbox=True, use_kdtree=True, verbose=0):
    """stardist/nms.py:186-227 -> compiled reference c_non_max_suppression_inds"""
    return ref.stardist2d().c_non_max_suppression_inds(_prep(dist, np.float32), _prep(points, np.float32),
                                                       int(use_kdtree), int(use_bbox), int(verbose), np.float32(thresh))


def non_maximum_suppression_sparse(dist, prob, points, nms_thresh=0.5, use_bbox=True, use_kdtree=True):
    """stardist/nms.py:135-183"""
    dist = np.asarray(dist); prob = np.asarray(prob); points = np.asarray(points)
    inds_original = np.arange(len(prob))
    _sorted = np.argsort(prob)[::-1]
    probi, disti, pointsi = prob[_sorted], dist[_sorted], points[_sorted]
    inds_original = inds_original[_sorted]
    inds = non_maximum_suppression_inds(disti, pointsi, scores=probi, thresh=nms_thresh, use_kdtree=use_kdtree)
    return pointsi[inds], probi[inds], disti[inds], inds_original[inds]


def non_maximum_suppression_3d_inds(dist, points, rays_vertices, rays_faces, scores, thresh=0.5, use_bbox=True,
                                    use_kdtree=True, verbose=0):
    """stardist/nms.py:327-384 -> compiled reference c_non_max_suppression_inds (3D)"""
    n_poly = dist.shape[0]
    if scores is None:
        scores = np.ones(n_poly)
    ind = np.argsort(scores)[::-1]
    survivors = np.ones(n_poly, bool)
    dist, points, scores = dist[ind], points[ind], scores[ind]
    survivors[ind] = ref.stardist3d().c_non_max_suppression_inds(
        _prep(dist, np.float32), _prep(points, np.float32), _prep(rays_vertices, np.float32),
        _prep(rays_faces, np.int32), _prep(scores, np.float32), int(use_bbox), int(use_kdtree), int(verbose),
        np.float32(thresh))
    return survivors


# ----------------------------------------------------------------------------- geom2d.py
def ray_angles(n_rays=32):
    """stardist/geometry/geom2d.py:214-215"""
    return np.linspace(0, 2 * np.pi, n_rays, endpoint=False)


def dist_to_coord(dist, points, scale_dist=(1, 1)):
    """stardist/geometry/geom2d.py:130-146"""
    dist = np.asarray(dist); points = np.asarray(points)
    n_rays = dist.shape[1]
    phis = ray_angles(n_rays)
    coord = (dist[:, np.newaxis] * np.array([np.sin(phis), np.cos(phis)])).astype(np.float32)
    coord *= np.asarray(scale_dist).reshape(1, 2, 1)
    coord += points[..., np.newaxis]
    return coord


def polygon(r, c, shape):
    """scikit-image skimage.draw.polygon(r, c, shape) restated (see module docstring).

    _polygon: bbox = [int(max(0,min)), min(shape-1, int(ceil(max)))], float64 vertices,
    point_in_polygon(cptr, rptr, c_i, r_i) != 0."""
    r = np.asanyarray(r); c = np.asanyarray(c)
    minr = int(max(0, r.min())); maxr = int(np.ceil(r.max()))
    minc = int(max(0, c.min())); maxc = int(np.ceil(c.max()))
    if shape is not None:
        maxr = min(shape[0] - 1, maxr); maxc = min(shape[1] - 1, maxc)
    if maxr < minr or maxc < minc:
        return np.zeros(0, np.intp), np.zeros(0, np.intp)
    yp = np.ascontiguousarray(r, "float64"); xp = np.ascontiguousarray(c, "float64")
    rr, cc = np.mgrid[minr:maxr + 1, minc:maxc + 1]
    y = rr.ravel().astype(np.float64); x = cc.ravel().astype(np.float64)
    eps = float(np.float32(1e-12))
    n = len(xp)
    l_cross = np.zeros(len(x), np.int64); r_cross = np.zeros(len(x), np.int64)
    vertex = np.zeros(len(x), bool)
    x1 = xp[n - 1] - x; y1 = yp[n - 1] - y
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(n):
            x0 = xp[i] - x; y0 = yp[i] - y
            vertex |= (-eps < x0) & (x0 < eps) & (-eps < y0) & (y0 < eps)
            q = (x0 * y1 - x1 * y0) / (y1 - y0)
            s = (y0 > 0) != (y1 > 0)
            r_cross += (s & (q > 0))
            s2 = (y0 < 0) != (y1 < 0)
            l_cross += (s2 & (q < 0))
            x1, y1 = x0, y0
    # the reference returns at the FIRST vertex hit; crossings counted before it do not matter
    inside = vertex | ((r_cross & 1) != (l_cross & 1)) | ((r_cross & 1) == 1)
    return rr.ravel()[inside], cc.ravel()[inside]


def polygons_to_label_coord(coord, shape, labels=None):
    """stardist/geometry/geom2d.py:149-166"""
    coord = np.asarray(coord)
    if labels is None:
        labels = np.arange(len(coord))
    lbl = np.zeros(shape, np.int32)
    for i, c in zip(labels, coord):
        rr, cc = polygon(*c, shape)
        lbl[rr, cc] = i + 1
    return lbl


def polygons_to_label(dist, points, shape, prob=None, thr=-np.inf, scale_dist=(1, 1)):
    """stardist/geometry/geom2d.py:169-197"""
    dist = np.asarray(dist); points = np.asarray(points)
    prob = np.inf * np.ones(len(points)) if prob is None else np.asarray(prob)
    ind = prob > thr
    points, dist, prob = points[ind], dist[ind], prob[ind]
    ind = np.argsort(prob, kind="stable")
    points, dist = points[ind], dist[ind]
    coord = dist_to_coord(dist, points, scale_dist=scale_dist)
    return polygons_to_label_coord(coord, shape=shape, labels=ind)


def star_dist(lbl, n_rays=32, grid=(1, 1)):
    """stardist/geometry/geom2d.py:29-31 -> compiled reference c_star_dist"""
    return ref.stardist2d().c_star_dist(lbl.astype(np.uint16, copy=False), np.int32(n_rays), np.int32(grid[0]), np.int32(grid[1]))


# ----------------------------------------------------------------------------- geom3d.py
def star_dist3D(lbl, rays_vertices, grid=(1, 1, 1)):
    """stardist/geometry/geom3d.py:16-24 -> compiled reference c_star_dist3d"""
    dz, dy, dx = np.asarray(rays_vertices).T
    return ref.stardist3d().c_star_dist3d(lbl.astype(np.uint16, copy=False), dz.astype(np.float32, copy=False),
                                          dy.astype(np.float32, copy=False), dx.astype(np.float32, copy=False),
                                          int(len(rays_vertices)), *tuple(int(a) for a in grid))


def polyhedron_to_label(dist, points, rays_vertices, rays_faces, shape, prob=None, thr=-np.inf, labels=None,
                        mode="full", verbose=False, overlap_label=None):
    """stardist/geometry/geom3d.py:100-198 -> compiled reference c_polyhedron_to_label"""
    if len(points) == 0:
        return np.zeros(shape, np.uint16)
    dist = np.asanyarray(dist); points = np.asanyarray(points)
    if dist.ndim == 1: dist = dist.reshape(1, -1)
    if points.ndim == 1: points = points.reshape(1, -1)
    if labels is None: labels = np.arange(1, len(points) + 1)
    if np.amin(dist) <= 0: raise ValueError("distance array should be positive!")
    prob = np.ones(len(points)) if prob is None else np.asanyarray(prob)
    modes = {"full": 0, "kernel": 1, "hull": 2, "bbox": 3, "debug": 4}
    lbl = np.zeros(shape, np.uint16)
    ind = np.where(prob >= thr)[0]
    if len(ind) == 0: return lbl
    prob, points, dist, labels = prob[ind], points[ind], dist[ind], np.asarray(labels)[ind]
    ind = np.argsort(prob)[::-1]
    points, dist, labels = points[ind], dist[ind], labels[ind]
    return ref.stardist3d().c_polyhedron_to_label(_prep(dist, np.float32), _prep(points, np.float32),
                                                  _prep(np.asarray(rays_vertices), np.float32), _prep(np.asarray(rays_faces), np.int32),
                                                  _prep(labels, np.int32), np.int32(modes[mode]), np.int32(verbose),
                                                  np.int32(overlap_label is not None),
                                                  np.int32(0 if overlap_label is None else overlap_label), tuple(int(s) for s in shape))


# ----------------------------------------------------------------------------- matching.py
def relabel_sequential(label_field, offset=1):
    """stardist/matching.py:319-408 (skimage.segmentation.relabel_sequential variant)."""
    offset = int(offset)
    if offset <= 0: raise ValueError("Offset must be strictly positive.")
    if np.min(label_field) < 0: raise ValueError("Cannot relabel array that contains negative values.")
    max_label = int(label_field.max())
    if not np.issubdtype(label_field.dtype, np.integer):
        new_type = np.min_scalar_type(max_label)
        label_field = label_field.astype(new_type)
    labels = np.unique(label_field)
    labels0 = labels[labels != 0]
    new_max_label = offset - 1 + len(labels0)
    new_labels0 = np.arange(offset, new_max_label + 1)
    output_type = label_field.dtype
    required_type = np.min_scalar_type(new_max_label)
    if np.dtype(required_type).itemsize > np.dtype(label_field.dtype).itemsize:
        output_type = required_type
    forward_map = np.zeros(max_label + 1, dtype=output_type)
    forward_map[labels0] = new_labels0
    inverse_map = np.zeros(new_max_label + 1, dtype=output_type)
    inverse_map[offset:] = labels0
    relabeled = forward_map[label_field]
    return relabeled, forward_map, inverse_map


# ----------------------------------------------------------------------------- utils.py
def edt_prob(lbl_img, anisotropy=None):
    """stardist/utils.py:71-125 (`_edt_prob_scipy`: per object, scipy's exact EDT of the object mask on its bounding box grown by one
    pixel where it does not touch the image border, divided by the object's maximum + 1e-10).  Restated without scipy as an
    exhaustive search: the grown box always contains an object pixel's nearest non-object pixel, so the value is the float64
    distance to the nearest pixel INSIDE THE IMAGE with another label.  Pinned to goldens made by the reference function
    (tests/test_cpu_oracle.py).  O(object pixels x image pixels): small test images only."""
    import warnings
    lbl = np.asarray(lbl_img)
    nd = lbl.ndim
    samp = np.ones(nd) if anisotropy is None else np.asarray(anisotropy, np.float64)
    constant = lbl.min() == lbl.max() and lbl.flat[0] > 0
    if constant:
        lbl = np.pad(lbl, ((1, 1),) * nd, mode="constant")
        warnings.warn("EDT of constant label image is ill-defined. (Assuming background around it.)")
    coords = np.stack(np.meshgrid(*[np.arange(s) for s in lbl.shape], indexing="ij"), -1).reshape(-1, nd).astype(np.float64)
    flat = lbl.reshape(-1)
    prob = np.zeros(flat.shape, np.float32)
    for l in np.unique(flat):
        if l <= 0:
            continue
        inside = np.flatnonzero(flat == l)
        other = coords[flat != l]
        d = np.empty(len(inside))
        for a in range(0, len(inside), 256):
            diff = (coords[inside[a:a + 256], None, :] - other[None]) * samp
            acc = np.zeros(diff.shape[:2])
            for k in range(nd):                                  # scipy sums the squared axis terms in axis order
                acc += diff[..., k] * diff[..., k]
            d[a:a + 256] = np.sqrt(acc.min(axis=1))
        prob[inside] = d / (d.max() + 1e-10)
    prob = prob.reshape(lbl.shape)
    if constant:
        prob = prob[(slice(1, -1),) * nd].copy()
    return prob
