"""Host-side mirror of the reference's stardist/nms.py on top of the HIP natives.

Same function names, arguments and return values as the reference (file:line cited per
function).  Arrays may be numpy (host entry points of the C ABI) or torch CUDA tensors
(device entry points, nothing leaves HBM); the result type follows the input type.
"""
import numpy as np

from .lib import _native as N
from .utils import _normalize_grid


def _is_t(x):
    return N.is_torch(x)


def _as_arrays(*xs):
    """np.asarray of every argument, as the reference's entry points do first (lists of rows are legal input there: nms.py:156-158,
    303-305) -- unless the caller works with device tensors, which pass through untouched"""
    if any(_is_t(x) for x in xs):
        return xs
    return tuple(np.asarray(x) for x in xs)


def _ind_prob_thresh(prob, prob_thresh, b=2):
    """stardist/nms.py:6-17"""
    if b is not None and np.isscalar(b):
        b = ((b, b),) * prob.ndim
    if _is_t(prob):
        import torch
        ind_thresh = prob > torch.tensor(prob_thresh, dtype=prob.dtype, device=prob.device)
        if b is not None:
            _ind = torch.zeros_like(ind_thresh)
            ss = tuple(slice(_bs[0] if _bs[0] > 0 else None, -_bs[1] if _bs[1] > 0 else None) for _bs in b)
            _ind[ss] = True
            ind_thresh &= _ind
        return ind_thresh
    ind_thresh = prob > prob_thresh
    if b is not None:
        _ind = np.zeros_like(ind_thresh)
        ss = tuple(slice(_bs[0] if _bs[0] > 0 else None, -_bs[1] if _bs[1] > 0 else None) for _bs in b)
        _ind[ss] = True
        ind_thresh &= _ind
    return ind_thresh


def _argsort_desc(x):
    """np.argsort(x)[::-1] (nms.py:114,167): ascending order, reversed. Ties: numpy's default sort is
    unstable, so the reference's tie order is implementation-defined; we use 'stable ascending,
    then reversed' on both backends (documented in DESIGN.md)."""
    if _is_t(x):
        return _sort_desc(x)[1]
    return np.argsort(x, kind="stable")[::-1]


def _sort_desc(x):
    """(sorted scores, order) of a score tensor in the order _argsort_desc states.  float32 scores on the GPU: the library's radix sort of
    (score, position) pairs + one reversing write (csrc/select.hip sd_sort_scores_desc_device, include/stardist_hip.h); anything else:
    the framework's stable sort, flipped."""
    import torch
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 1 and 0 < x.numel() < 2 ** 31:
        from .lib import _native as N
        x = x.contiguous()
        sp = torch.empty_like(x)
        order = torch.empty(x.numel(), dtype=torch.int64, device=x.device)
        N.dcall(x, "sd_sort_scores_desc_device", N.tptr(x), int(x.numel()), N.tptr(sp), N.tptr(order))
        return sp, order
    sp, order = torch.sort(x, stable=True)
    return torch.flip(sp, dims=(0,)), torch.flip(order, dims=(0,))


def non_maximum_suppression_inds(dist, points, scores, thresh=0.5, use_bbox=True, use_kdtree=True, verbose=1):
    """stardist/nms.py:186-227. Polygons must be sorted by score (descending). Returns bool survivors."""
    from .lib.stardist2d import c_non_max_suppression_inds
    assert dist.ndim == 2 and points.ndim == 2
    n_poly = dist.shape[0]
    assert points.shape[0] == n_poly and (scores is None or len(scores) == n_poly)
    if _is_t(dist):
        import torch
        d = dist.to(torch.float32).contiguous()
        p = points.to(torch.float32).contiguous()
    else:
        d = np.ascontiguousarray(dist.astype(np.float32, copy=False))
        p = np.ascontiguousarray(points.astype(np.float32, copy=False))
    return c_non_max_suppression_inds(d, p, int(use_kdtree), int(use_bbox), int(verbose), np.float32(thresh))


def _non_maximum_suppression_old(coord, prob, grid=(1, 1), b=2, nms_thresh=0.5, prob_thresh=0.5, verbose=False, max_bbox_search=True):
    """stardist/nms.py:20-74: the legacy NMS on dense polygon coordinates (Ny,Nx,2,n_rays) and prob (Ny,Nx); returns the retained
    grid points (np.nonzero order).  Kept because the reference keeps it as a second statement of the same NMS (tests/test_nms2D.py:78-110)."""
    from .lib.stardist2d import c_non_max_suppression_inds_old
    coord = np.asarray(coord); prob = np.asarray(prob)
    assert prob.ndim == 2
    assert coord.ndim == 4
    grid = _normalize_grid(grid, 2)
    mask = _ind_prob_thresh(prob, prob_thresh, b)
    polygons = coord[mask]
    scores = prob[mask]
    ind = _argsort_desc(scores)
    survivors = np.zeros(len(ind), bool)
    polygons = polygons[ind]
    scores = scores[ind]
    if max_bbox_search:
        # pixel -> id of the score-sorted polygon there, -1: no candidate (nms.py:56-58)
        mapping = -np.ones(mask.shape, np.int32)
        mapping.flat[np.flatnonzero(mask)[ind]] = range(len(ind))
    else:
        mapping = np.empty((0, 0), np.int32)
    survivors[ind] = c_non_max_suppression_inds_old(np.ascontiguousarray(polygons.astype(np.int32)), mapping, np.float32(nms_thresh),
                                                    np.int32(max_bbox_search), np.int32(grid[0]), np.int32(grid[1]), np.int32(verbose))
    if verbose:
        print("keeping %s/%s polygons" % (np.count_nonzero(survivors), len(polygons)))
    points = np.stack([ii[survivors] for ii in np.nonzero(mask)], axis=-1)
    return points


def non_maximum_suppression(dist, prob, grid=(1, 1), b=2, nms_thresh=0.5, prob_thresh=0.5,
                            use_bbox=True, use_kdtree=True, verbose=False):
    """stardist/nms.py:77-132: dense (Ny,Nx,n_rays)/(Ny,Nx) maps -> (points, prob, dist) of survivors."""
    assert prob.ndim == 2 and dist.ndim == 3 and tuple(prob.shape) == tuple(dist.shape[:2])
    grid = _normalize_grid(grid, 2)
    mask = _ind_prob_thresh(prob, prob_thresh, b)
    if _is_t(prob):
        import torch
        points = torch.stack(torch.where(mask), dim=1)
        dist = dist[mask]; scores = prob[mask]
        ind = _argsort_desc(scores)
        dist, scores, points = dist[ind], scores[ind], points[ind]
        points = points * torch.tensor(grid, device=points.device).reshape(1, 2)
        inds = non_maximum_suppression_inds(dist, points.to(torch.int32), scores=scores, use_bbox=use_bbox,
                                            use_kdtree=use_kdtree, thresh=nms_thresh, verbose=verbose)
        return points[inds], scores[inds], dist[inds]
    dist = np.asarray(dist); prob = np.asarray(prob)
    points = np.stack(np.where(mask), axis=1)
    dist = dist[mask]; scores = prob[mask]
    ind = _argsort_desc(scores)
    dist, scores, points = dist[ind], scores[ind], points[ind]
    points = (points * np.array(grid).reshape((1, 2)))
    inds = non_maximum_suppression_inds(dist, points.astype(np.int32, copy=False), scores=scores, use_bbox=use_bbox,
                                        use_kdtree=use_kdtree, thresh=nms_thresh, verbose=verbose)
    return points[inds], scores[inds], dist[inds]


def non_maximum_suppression_sparse(dist, prob, points, b=2, nms_thresh=0.5, use_bbox=True, use_kdtree=True, verbose=False):
    """stardist/nms.py:135-183: candidate lists -> (points, prob, dist, inds) of survivors."""
    dist, prob, points = _as_arrays(dist, prob, points)
    assert dist.ndim == 2 and prob.ndim == 1 and points.ndim == 2 and points.shape[-1] == 2 and \
        len(prob) == len(dist) == len(points)
    _sorted = _argsort_desc(prob)
    if _is_t(prob):
        import torch
        inds_original = torch.arange(len(prob), device=prob.device)[_sorted]
    else:
        dist = np.asarray(dist); prob = np.asarray(prob); points = np.asarray(points)
        inds_original = np.arange(len(prob))[_sorted]
    probi, disti, pointsi = prob[_sorted], dist[_sorted], points[_sorted]
    inds = non_maximum_suppression_inds(disti, pointsi, scores=probi, thresh=nms_thresh, use_kdtree=use_kdtree, verbose=verbose)
    if _is_t(inds):
        inds = _survivor_positions(inds)          # positions once (one read-back) instead of one boolean-mask indexing per array
    return pointsi[inds], probi[inds], disti[inds], inds_original[inds]


def _survivor_positions(keep):
    """positions of the True entries of a device bool tensor, ascending (what boolean-mask indexing computes internally, once)"""
    import torch
    return torch.nonzero(keep).reshape(-1)


def non_maximum_suppression_sparse_sorted(dist, prob, points, b=2, nms_thresh=0.5, use_bbox=True, use_kdtree=True, verbose=False):
    """non_maximum_suppression_sparse (stardist/nms.py:135-183) for candidates that are ALREADY in score order (descending, the order
    `np.argsort(prob)[::-1]` gives them) as device tensors: positions (int64 tensor) of the survivors, best score first."""
    return _survivor_positions(nms_keep_sorted(dist, prob, points, b=b, nms_thresh=nms_thresh, use_bbox=use_bbox, use_kdtree=use_kdtree, verbose=verbose))


def nms_keep_sorted(dist, prob, points, b=2, nms_thresh=0.5, use_bbox=True, use_kdtree=True, verbose=False):
    """the keep flags (uint8 device tensor) behind non_maximum_suppression_sparse_sorted: same arguments, same candidates in score order"""
    from .lib.stardist2d import c_non_max_suppression_inds
    import torch
    assert dist.ndim == 2 and prob.ndim == 1 and points.ndim == 2 and points.shape[-1] == 2 and len(prob) == len(dist) == len(points)
    return c_non_max_suppression_inds(dist.to(torch.float32).contiguous(), points.to(torch.float32).contiguous(), int(use_kdtree), 1, int(verbose),
                                      np.float32(nms_thresh), _as_uint8=True)          # (use_bbox: nms.py:175-176 passes its default)


def non_maximum_suppression_3d_sparse_sorted(dist, prob, points, rays, b=2, nms_thresh=0.5, use_kdtree=True, verbose=False):
    """non_maximum_suppression_3d_sparse (stardist/nms.py:285-324) for candidates already in score order (descending) as device tensors:
    positions (int64 tensor) of the survivors, best score first.  (non_maximum_suppression_3d_inds re-sorts by score, nms.py:352: a
    stable re-sort of a sorted list is the identity, so it is skipped.)"""
    from .lib.stardist3d import c_non_max_suppression_inds
    from .rays3d import rays_device_tensors, warn_if_degenerate
    assert dist.ndim == 2 and prob.ndim == 1 and points.ndim == 2 and dist.shape[-1] == len(rays) and points.shape[-1] == 3 and \
        len(prob) == len(dist) == len(points)
    warn_if_degenerate(rays)
    verts, faces = rays_device_tensors(rays, dist.device)
    keep = c_non_max_suppression_inds(dist, points, verts, faces, prob, 1, int(use_kdtree), int(verbose), np.float32(nms_thresh), _as_uint8=True)
    return _survivor_positions(keep)


# ----------------------------------------------------------------------------- 3D
def non_maximum_suppression_3d_inds(dist, points, rays, scores, thresh=0.5, use_bbox=True, use_kdtree=True, verbose=1):
    """stardist/nms.py:327-384 (re-sorts by score itself, returns survivors in input order)."""
    from .lib.stardist3d import c_non_max_suppression_inds
    from .rays3d import warn_if_degenerate
    assert dist.ndim == 2 and points.ndim == 2 and dist.shape[1] == len(rays)
    warn_if_degenerate(rays)
    n_poly = dist.shape[0]
    if _is_t(dist):
        import torch
        if scores is None:
            scores = torch.ones(n_poly, device=dist.device)
        ind = _argsort_desc(scores)
        survivors = torch.ones(n_poly, dtype=torch.bool, device=dist.device)
        from .rays3d import rays_device_tensors
        verts, faces = rays_device_tensors(rays, dist.device)
        survivors[ind] = c_non_max_suppression_inds(dist[ind].float().contiguous(), points[ind].float().contiguous(),
                                                    verts, faces, scores[ind].float().contiguous(),
                                                    int(use_bbox), int(use_kdtree), int(verbose), np.float32(thresh))
        return survivors
    if scores is None:
        scores = np.ones(n_poly)
    ind = _argsort_desc(scores)
    survivors = np.ones(n_poly, bool)
    dist, points, scores = dist[ind], points[ind], scores[ind]

    def _prep(x, dtype):
        return np.ascontiguousarray(x.astype(dtype, copy=False))
    survivors[ind] = c_non_max_suppression_inds(_prep(dist, np.float32), _prep(points, np.float32),
                                                _prep(rays.vertices, np.float32), _prep(rays.faces, np.int32),
                                                _prep(scores, np.float32), int(use_bbox), int(use_kdtree), int(verbose),
                                                np.float32(thresh))
    return survivors


def non_maximum_suppression_3d(dist, prob, rays, grid=(1, 1, 1), b=2, nms_thresh=0.5, prob_thresh=0.5, use_bbox=True,
                               use_kdtree=True, verbose=False):
    """stardist/nms.py:233-282"""
    dist = np.asarray(dist); prob = np.asarray(prob)
    assert prob.ndim == 3 and dist.ndim == 4 and dist.shape[-1] == len(rays) and prob.shape == dist.shape[:3]
    grid = _normalize_grid(grid, 3)
    ind_thresh = _ind_prob_thresh(prob, prob_thresh, b)
    points = np.stack(np.where(ind_thresh), axis=1)
    probi = prob[ind_thresh]; disti = dist[ind_thresh]
    _sorted = _argsort_desc(probi)
    probi, disti, points = probi[_sorted], disti[_sorted], points[_sorted]
    points = (points * np.array(grid).reshape((1, 3)))
    inds = non_maximum_suppression_3d_inds(disti, points, rays=rays, scores=probi, thresh=nms_thresh, use_bbox=use_bbox,
                                           use_kdtree=use_kdtree, verbose=verbose)
    return points[inds], probi[inds], disti[inds]


def non_maximum_suppression_3d_sparse(dist, prob, points, rays, b=2, nms_thresh=0.5, use_kdtree=True, verbose=False):
    """stardist/nms.py:285-324"""
    dist, prob, points = _as_arrays(dist, prob, points)
    assert dist.ndim == 2 and prob.ndim == 1 and points.ndim == 2 and dist.shape[-1] == len(rays) and \
        points.shape[-1] == 3 and len(prob) == len(dist) == len(points)
    _sorted = _argsort_desc(prob)
    if _is_t(prob):
        import torch
        inds_original = torch.arange(len(prob), device=prob.device)[_sorted]
    else:
        dist = np.asarray(dist); prob = np.asarray(prob); points = np.asarray(points)
        inds_original = np.arange(len(prob))[_sorted]
    probi, disti, pointsi = prob[_sorted], dist[_sorted], points[_sorted]
    inds = non_maximum_suppression_3d_inds(disti, pointsi, rays=rays, scores=probi, thresh=nms_thresh, use_kdtree=use_kdtree, verbose=verbose)
    if _is_t(inds):
        inds = _survivor_positions(inds)
    return pointsi[inds], probi[inds], disti[inds], inds_original[inds]
