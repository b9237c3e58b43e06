"""Mirror of stardist/geometry/geom3d.py (hot-path functions) on top of the HIP natives."""
import numpy as np

from ..lib import _native as N
from ..utils import _normalize_grid


def star_dist3D(lbl, rays, grid=(1, 1, 1), mode="hip"):
    """geom3d.py:86-97 ('hip' replaces 'cpp' / 'opencl'); lbl: label volume, 0 = background."""
    from ..lib.stardist3d import c_star_dist3d
    grid = _normalize_grid(grid, 3)
    if mode not in ("hip", "cpp", "opencl"):
        raise ValueError("Unknown mode %s" % mode)
    dz, dy, dx = rays.vertices.T
    if N.is_torch(lbl):
        import torch
        src = lbl if lbl.dtype == torch.uint16 else lbl.to(torch.uint16)
    else:
        src = lbl.astype(np.uint16, copy=False)
    return c_star_dist3d(src, dz.astype(np.float32, copy=False), dy.astype(np.float32, copy=False),
                         dx.astype(np.float32, copy=False), int(len(rays)), *tuple(int(a) for a in grid))


def polyhedron_to_label(dist, points, rays, shape, prob=None, thr=-np.inf, labels=None, mode="full", verbose=True,
                        overlap_label=None, window=None):
    """geom3d.py:100-198: filters prob >= thr, sorts by descending prob, paints first-writer-wins.
    window = ((z0, y0, x0), (nz, ny, nx)) (device tensors): only that part of the volume is rendered and returned."""
    from ..lib.stardist3d import c_polyhedron_to_label

    def _empty():
        """geom3d.py:128-131: a background-only uint16 image -- of the WINDOW (device tensor, like every windowed result) when one is asked for"""
        if window is None:
            return np.zeros(shape, np.uint16)
        import torch
        dev = dist.device if N.is_torch(dist) else (points.device if N.is_torch(points) else "cpu")
        return torch.zeros(tuple(int(v) for v in window[1]), dtype=torch.int32, device=dev)
    if len(points) == 0:
        if verbose:
            print("warning: empty list of points (returning background-only image)")
        return _empty()
    modes = {"full": 0, "kernel": 1, "hull": 2, "bbox": 3, "debug": 4}
    if mode not in modes:
        raise KeyError("Unknown render mode '%s' , allowed:  %s" % (mode, tuple(modes.keys())))
    if mode in ("full", "hull"):
        from ..rays3d import warn_if_degenerate
        warn_if_degenerate(rays)
    if N.is_torch(dist):
        import torch
        if dist.dim() == 1: dist = dist.reshape(1, -1)
        if points.dim() == 1: points = points.reshape(1, -1)
        dev = dist.device
        if labels is None: labels = torch.arange(1, len(points) + 1, device=dev)
        if float(dist.min()) <= 0: raise ValueError("distance array should be positive!")
        prob = torch.ones(len(points), device=dev) if prob is None else prob
        if dist.dim() != 2: raise ValueError("dist should be 2 dimensional but has shape %s" % str(tuple(dist.shape)))
        if dist.shape[1] != len(rays): raise ValueError("inconsistent number of rays!")
        if len(prob) != len(points): raise ValueError("len(prob) != len(points)")
        if len(labels) != len(points): raise ValueError("len(labels) != len(points)")
        if thr != -np.inf:                                          # (prob >= -inf holds for every score: no compaction, no read-back)
            ind = torch.where(prob >= thr)[0]
            if len(ind) == 0:
                if verbose: print("warning: no points found with probability>= {thr:.4f} (returning background-only image)".format(thr=thr))
                return _empty()
            prob, points, dist, labels = prob[ind], points[ind], dist[ind], labels[ind]
        ind = torch.flip(torch.sort(prob, stable=True)[1], dims=(0,))
        points, dist, labels = points.index_select(0, ind), dist.index_select(0, ind), labels.index_select(0, ind)
        from ..rays3d import rays_device_tensors
        verts, faces = rays_device_tensors(rays, dev)
        if window is not None:                                     # polyhedra whose bounding box misses the window are dropped up front
            o = torch.tensor(window[0], device=dev, dtype=torch.float32); e = o + torch.tensor(window[1], device=dev, dtype=torch.float32)
            reach = dist.float().amax(dim=1, keepdim=True) * verts.abs().amax(dim=0, keepdim=True) + 2.0
            pf = points.float()
            hit = torch.all((pf + reach >= o) & (pf - reach <= e), dim=1)
            points, dist, labels = points[hit], dist[hit], labels[hit]
        return c_polyhedron_to_label(dist.float().contiguous(), points.float().contiguous(), verts, faces, labels.to(torch.int32).contiguous(),
                                     np.int32(modes[mode]), np.int32(verbose), np.int32(overlap_label is not None),
                                     np.int32(0 if overlap_label is None else overlap_label), shape, window=window)
    assert window is None, "window rendering takes device tensors"
    dist = np.asanyarray(dist); points = np.asanyarray(points)
    if dist.ndim == 1: dist = dist.reshape(1, -1)
    if points.ndim == 1: points = points.reshape(1, -1)
    if labels is None: labels = np.arange(1, len(points) + 1)
    if np.amin(dist) <= 0: raise ValueError("distance array should be positive!")
    prob = np.ones(len(points)) if prob is None else np.asanyarray(prob)
    if dist.ndim != 2: raise ValueError("dist should be 2 dimensional but has shape %s" % str(dist.shape))
    if dist.shape[1] != len(rays): raise ValueError("inconsistent number of rays!")
    if len(prob) != len(points): raise ValueError("len(prob) != len(points)")
    if len(labels) != len(points): raise ValueError("len(labels) != len(points)")
    lbl = np.zeros(shape, np.uint16)
    ind = np.where(prob >= thr)[0]
    if len(ind) == 0:
        if verbose: print("warning: no points found with probability>= {thr:.4f} (returning background-only image)".format(thr=thr))
        return lbl
    prob, points, dist, labels = prob[ind], points[ind], dist[ind], np.asarray(labels)[ind]
    ind = np.argsort(prob, kind="stable")[::-1]
    points, dist, labels = points[ind], dist[ind], labels[ind]

    def _prep(x, dtype):
        return np.ascontiguousarray(x.astype(dtype, copy=False))
    return c_polyhedron_to_label(_prep(dist, np.float32), _prep(points, np.float32), _prep(rays.vertices, np.float32),
                                 _prep(rays.faces, np.int32), _prep(labels, np.int32), np.int32(modes[mode]), np.int32(verbose),
                                 np.int32(overlap_label is not None), np.int32(0 if overlap_label is None else overlap_label), shape)


def dist_to_coord3D(dist, points, rays_vertices):
    """geom3d.py:261-274"""
    dist = np.asarray(dist); points = np.asarray(points); rays_vertices = np.asarray(rays_vertices)
    if not all((len(dist) == len(points), dist.ndim == 2, points.ndim == 2, points.shape[-1] == 3,
                rays_vertices.shape[-1] == 3, dist.shape[-1] == len(rays_vertices))):
        raise ValueError("Wrong shapes! dist -> (m,n) points -> (m,3) rays_vertices -> (m,)")
    return points[:, np.newaxis] + dist[..., np.newaxis] * rays_vertices


def relabel_image_stardist3D(lbl, rays, verbose=False, **kwargs):
    """geom3d.py:201-217: relabel each label region in `lbl` with its star representation (star_dist3D at the truncated region
    centroid, clamped below at 1e-3 -> polyhedron_to_label with the regions' own label ids)."""
    from .geom2d import _check_label_array, _region_centroids
    lbl = np.asarray(lbl)
    _check_label_array(lbl, "lbl")
    if not lbl.ndim == 3:
        raise ValueError("lbl image should be 3 dimensional")
    dist_all = star_dist3D(lbl, rays, **kwargs)
    labs, cen = _region_centroids(lbl)
    points = cen.astype(int)
    dist = np.maximum(np.asarray(dist_all)[tuple(points.T)].reshape(len(points), len(rays)), 1e-3)
    return polyhedron_to_label(dist, points, rays, shape=lbl.shape, labels=labs, verbose=verbose)


def dist_to_volume(dist, rays):
    """volumes of the polyhedra, dist.shape = (nz, ny, nx, n_rays) (geom3d.py:220-235)"""
    from ..lib.stardist3d import c_dist_to_volume
    if dist.ndim != 4:
        raise ValueError("dist.ndim = %d but should be 4" % dist.ndim)
    if dist.shape[-1] != len(rays):
        raise ValueError("dist.shape[-1] = %d but should be %d" % (dist.shape[-1], len(rays)))
    return c_dist_to_volume(dist, rays.vertices.astype(np.float32), rays.faces.astype(np.int32))


def dist_to_centroid(dist, rays, mode="absolute"):
    """centroids of the polyhedra, mode = 'absolute' or 'relative' (geom3d.py:238-257)"""
    from ..lib.stardist3d import c_dist_to_centroid
    if dist.ndim != 4:
        raise ValueError("dist.ndim = %d but should be 4" % dist.ndim)
    if dist.shape[-1] != len(rays):
        raise ValueError("dist.shape[-1] = %d but should be %d" % (dist.shape[-1], len(rays)))
    if mode not in ("absolute", "relative"):
        raise ValueError("mode should be either 'absolute' or 'relative'")
    return c_dist_to_centroid(dist, rays.vertices.astype(np.float32), rays.faces.astype(np.int32), int(mode == "absolute"))


def export_to_obj_file3D(polys, fname=None, scale=1, single_mesh=True, uv_map=False, name="poly"):
    """Wavefront .obj text of the predicted polyhedra (geom3d.py:277-347): one `o` block (or one per polyhedron), vertices as
    `v x y z` with the reference's precision rule, optional spherical `vt` coordinates of the rays, faces `f a/a b/b c/c` with
    running 1-based vertex indices.  polys = the dict returned by StarDist3D.predict_instances."""
    try:
        dist, points = np.asarray(polys["dist"]), np.asarray(polys["points"])
        rays_vertices, rays_faces = np.asarray(polys["rays_vertices"]), np.asarray(polys["rays_faces"])
    except KeyError as e:
        raise ValueError("polys should be a dict with keys 'dist', 'points', 'rays_vertices', 'rays_faces' "
                         "(such as generated by StarDist3D.predict_instances)") from e
    coord = dist_to_coord3D(dist, points, rays_vertices)
    if not (coord.ndim == 3 and coord.shape[-1] == 3 and rays_faces.shape[-1] == 3):
        raise ValueError("Wrong shapes! coord -> (m,n,3) rays_faces -> (k,3)")
    scale = np.asarray((scale,) * 3 if np.isscalar(scale) else scale)
    assert len(scale) == 3
    coord = coord * scale
    decimals = int(max(1, 1 - np.log10(np.min(scale))))
    vfmt = "v %%.%df %%.%df %%.%df\n" % (decimals, decimals, decimals)
    uv_lines = ""
    if uv_map:
        sv = scale * rays_vertices
        sv = sv / np.linalg.norm(sv, axis=1, keepdims=True)
        u = 1 - (.5 + .5 * np.arctan2(sv[:, 0], sv[:, 2]) / np.pi)
        v = 1 - (.5 - np.arcsin(sv[:, 1]) / np.pi)
        uv_lines = "".join("vt %.4f %.4f\n" % (a, b) for a, b in zip(u, v))
    faces1 = rays_faces.astype(np.int64) + 1
    parts = []
    for i, xs in enumerate(coord):
        if i == 0 or not single_mesh:
            parts.append("o %s_%d\n" % (name, i))
        parts.append("".join(vfmt % (p[2], p[1], p[0]) for p in xs))           # (z, y, x) -> x y z
        parts.append(uv_lines)
        f = faces1 + i * len(xs)
        parts.append("".join("f %d/%d %d/%d %d/%d\n" % (a, a, b, b, c, c) for a, b, c in f))
    obj_str = "".join(parts)
    if fname is not None:
        with open(fname, "w") as fh:
            fh.write(obj_str)
    return obj_str
