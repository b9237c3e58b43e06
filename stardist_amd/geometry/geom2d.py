"""Mirror of stardist/geometry/geom2d.py (hot-path functions) on top of the HIP natives."""
import numpy as np

from ..lib import _native as N
from ..utils import _normalize_grid


_SC_CACHE = {}


def ray_angles(n_rays=32):
    """geom2d.py:214-215"""
    return np.linspace(0, 2 * np.pi, n_rays, endpoint=False)


def star_dist(a, n_rays=32, grid=(1, 1), mode="hip"):
    """geom2d.py:73-85. 'a' is a label image (0 = background). mode 'hip' replaces 'cpp'/'opencl'."""
    from ..lib.stardist2d import c_star_dist
    n_rays >= 3 or (_ for _ in ()).throw(ValueError("need 'n_rays' >= 3"))
    if mode not in ("hip", "cpp", "opencl"):
        raise ValueError("Unknown mode %s" % mode)
    grid = _normalize_grid(grid, 2)
    if N.is_torch(a):
        import torch
        return c_star_dist(a.to(torch.uint16) if a.dtype != torch.uint16 else a, np.int32(n_rays), np.int32(grid[0]), np.int32(grid[1]))
    return c_star_dist(a.astype(np.uint16, copy=False), np.int32(n_rays), np.int32(grid[0]), np.int32(grid[1]))


def _dist_to_coord_old(rhos, grid=(1, 1)):
    """geom2d.py:88-109: dense polar -> cartesian for one image (3-D) or several (4-D): coord (..., h, w, 2, n_rays) in the dtype of rhos."""
    grid = _normalize_grid(grid, 2)
    rhos = np.asarray(rhos)
    is_single_image = rhos.ndim == 3
    if is_single_image:
        rhos = np.expand_dims(rhos, 0)
    assert rhos.ndim == 4
    n_images, h, w, n_rays = rhos.shape
    coord = np.empty((n_images, h, w, 2, n_rays), dtype=rhos.dtype)
    start = np.indices((h, w))
    for i in range(2):
        coord[..., i, :] = grid[i] * np.broadcast_to(start[i].reshape(1, h, w, 1), (n_images, h, w, n_rays))
    phis = ray_angles(n_rays).reshape(1, 1, 1, n_rays)
    coord[..., 0, :] += rhos * np.sin(phis)   # row coordinate
    coord[..., 1, :] += rhos * np.cos(phis)   # col coordinate
    return coord[0] if is_single_image else coord


def _polygons_to_label_old(coord, prob, points, shape=None, thr=-np.inf):
    """geom2d.py:112-127: paint the polygons of the dense coordinate map at `points` with increasing probability (ids 1, 2, ... in
    that order); the per-polygon skimage.draw.polygon loop runs as one call of the HIP rasteriser."""
    coord = np.asarray(coord); prob = np.asarray(prob); points = np.asarray(points)
    sh = coord.shape[:2] if shape is None else shape
    if len(points) == 0:
        return np.zeros(sh, np.int32)
    pr = prob[points[:, 0], points[:, 1]]
    ind = np.argsort(pr)
    points = points[ind]
    points = points[prob[points[:, 0], points[:, 1]] >= thr]
    c = np.ascontiguousarray(coord[points[:, 0], points[:, 1]], np.float32)
    return polygons_to_label_coord(c, sh, labels=np.arange(len(points)))


def dist_to_coord(dist, points, scale_dist=(1, 1)):
    """geom2d.py:130-146: polar -> cartesian, coord (n_polys, 2, n_rays) float32."""
    assert dist.ndim == 2 and points.ndim == 2 and len(dist) == len(points) and points.shape[1] == 2 and len(scale_dist) == 2
    n_rays = dist.shape[1]
    phis = ray_angles(n_rays)
    sc = np.array([np.sin(phis), np.cos(phis)])           # float64 (2, n_rays)
    if N.is_torch(dist) and dist.is_cuda and dist.dtype == __import__("torch").float32 and N.is_torch(points):
        # one launch in numpy's arithmetic (sd_dist_to_coord_device): f32 * f64 products rounded to f32, the centre added in f64 and
        # rounded once; the (sin, cos) table is the host libm's, uploaded once per (n_rays, device)
        import torch
        key = (n_rays, str(dist.device))
        sct = _SC_CACHE.get(key)
        if sct is None:
            sct = _SC_CACHE[key] = torch.as_tensor(np.ascontiguousarray(sc), device=dist.device)
        d = dist.contiguous()
        p = points.to(torch.float64).contiguous()
        coord = torch.empty((d.shape[0], 2, n_rays), dtype=torch.float32, device=dist.device)
        if d.shape[0]:
            import ctypes
            N.dcall(d, "sd_dist_to_coord_device", N.tptr(d), N.tptr(p), N.tptr(sct), int(d.shape[0]), int(n_rays),
                    ctypes.c_double(float(scale_dist[0])), ctypes.c_double(float(scale_dist[1])), N.tptr(coord))
        return coord
    if N.is_torch(dist):
        import torch
        sct = torch.as_tensor(sc, device=dist.device)                      # float64
        coord = (dist[:, None].to(torch.float64) * sct).to(torch.float32)  # f32*f64 product -> f32, as numpy does
        sd = torch.as_tensor(np.asarray(scale_dist, np.float64).reshape(1, 2, 1), device=dist.device)
        coord = (coord.to(torch.float64) * sd).to(torch.float32) if tuple(scale_dist) != (1, 1) else coord
        # numpy's in-place `coord += points[..., None]` adds in the common type of both operands and rounds ONCE to float32
        # (points may be int64 / float64): add in float64, cast once
        coord = (coord.to(torch.float64) + points[..., None].to(torch.float64)).to(torch.float32)
        return coord
    dist = np.asarray(dist); points = np.asarray(points)
    coord = (dist[:, np.newaxis] * sc).astype(np.float32)
    coord *= np.asarray(scale_dist).reshape(1, 2, 1)
    coord += points[..., np.newaxis]
    return coord


def polygons_to_label_coord(coord, shape, labels=None, window=None):
    """geom2d.py:149-166: paint polygons in the given order (later overwrite earlier), value labels[i]+1.
    window = ((y0, x0), (h, w)) (device tensors): only that part of the image is rendered and returned."""
    from ..lib.stardist2d import c_polygons_to_label
    assert coord.ndim == 3 and coord.shape[1] == 2
    n = len(coord)
    if N.is_torch(coord):
        import torch
        if labels is None:
            labels = torch.arange(n, device=coord.device)
        assert len(labels) == n
        if window is not None and n:                               # polygons whose bounding box misses the window are dropped up front
            (y0, x0), (h, w) = window
            lo, hi = coord.amin(dim=2), coord.amax(dim=2)
            hit = (hi[:, 0] >= y0 - 1) & (lo[:, 0] <= y0 + h) & (hi[:, 1] >= x0 - 1) & (lo[:, 1] <= x0 + w)
            coord, labels = coord[hit], labels[hit]
        return c_polygons_to_label(coord, labels.to(torch.int32), shape, window=window)
    assert window is None, "window rendering takes device tensors"
    coord = np.asarray(coord)
    if labels is None:
        labels = np.arange(n)
    labels = np.asarray(labels)
    if not (np.issubdtype(labels.dtype, np.integer) or labels.dtype == bool):
        raise ValueError("labels must be an array of integers")
    assert len(labels) == n
    return c_polygons_to_label(coord, labels.astype(np.int32), shape)


def polygons_to_label(dist, points, shape, prob=None, thr=-np.inf, scale_dist=(1, 1), window=None):
    """geom2d.py:169-197: label ids are consecutive and adhere to the order given."""
    assert dist.ndim == 2 and points.ndim == 2 and len(dist) == len(points) and points.shape[1] == 2
    if N.is_torch(dist):
        import torch
        prob = torch.full((len(points),), float("inf"), device=dist.device) if prob is None else prob
        ind = prob > thr
        points, dist, prob = points[ind], dist[ind], prob[ind]
        ind = torch.sort(prob, stable=True)[1]
        points, dist = points[ind], dist[ind]
        coord = dist_to_coord(dist, points, scale_dist=scale_dist)
        return polygons_to_label_coord(coord, shape=shape, labels=ind, window=window)
    assert window is None, "window rendering takes device tensors"
    dist = np.asarray(dist); points = np.asarray(points)
    prob = np.inf * np.ones(len(points)) if prob is None else np.asarray(prob)
    assert len(points) == len(prob) and prob.ndim == 1
    ind = prob > thr
    points, dist, prob = points[ind], dist[ind], prob[ind]
    ind = np.argsort(prob, kind="stable")
    points, dist = points[ind], dist[ind]
    coord = dist_to_coord(dist, points, scale_dist=scale_dist)
    return polygons_to_label_coord(coord, shape=shape, labels=ind)


def _region_centroids(lbl):
    """(labels, centroids): what `skimage.measure.regionprops(lbl)` yields as (r.label, r.centroid) -- the labels present in ascending
    order, the centroid = the mean of the region's pixel coordinates (skimage: `coords.mean(axis=0)`, a float64 quotient of an exactly
    representable integer sum and the pixel count; reproduced here as that quotient, so the truncation `astype(int)` the callers apply
    sees the very same float64).  Per object on its bounding box (scipy.ndimage.find_objects), like regionprops: no image-sized temporaries."""
    from scipy.ndimage import find_objects
    lbl = np.asarray(lbl)
    boxes = [(i, sl) for i, sl in enumerate(find_objects(lbl), 1) if sl is not None]
    labs = np.array([i for i, _ in boxes], np.int64)
    cen = np.empty((len(boxes), lbl.ndim), np.float64)
    for k, (lab, sl) in enumerate(boxes):
        m = lbl[sl] == lab
        n = int(m.sum())
        for d in range(lbl.ndim):
            along = m.sum(axis=tuple(a for a in range(lbl.ndim) if a != d), dtype=np.int64)           # pixels of the object per coordinate
            cen[k, d] = int((along * np.arange(sl[d].start, sl[d].stop, dtype=np.int64)).sum()) / n    # exact integer sum, one rounding
    return labs, cen


def _check_label_array(y, name=None):
    """matching.py:23-35 (the non-sequential form the geometry functions use)"""
    y = np.asarray(y)
    if not np.issubdtype(y.dtype, np.integer) or (y.size and y.min() < 0):
        raise ValueError("%s must be an array of non-negative integers." % ("labels" if name is None else name))
    return True


def relabel_image_stardist(lbl, n_rays, **kwargs):
    """geom2d.py:200-211: relabel each label region in `lbl` with its star representation (star_dist at the truncated region
    centroid -> polygons_to_label).  Label ids of the result are 1..n in the order of the regions' ids, as in the reference."""
    lbl = np.asarray(lbl)
    _check_label_array(lbl, "lbl")
    if not lbl.ndim == 2:
        raise ValueError("lbl image should be 2 dimensional")
    dist = star_dist(lbl, n_rays, **kwargs)
    points = _region_centroids(lbl)[1].astype(int)
    if len(points) == 0:
        dist, points = np.zeros((0, n_rays), np.float32), np.zeros((0, 2), int)
    else:
        dist = dist[tuple(points.T)]
    return polygons_to_label(dist, points, shape=lbl.shape)
