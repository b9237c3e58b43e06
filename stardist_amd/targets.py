"""Training targets of a batch of label images, computed on the GPU (the natives the reference's data generators
call per sample -- `edt_prob` and `star_dist` / `star_dist3D` -- as HIP kernels behind one call).

Mirror of the target part of StarDistData2D.__getitem__ / StarDistData3D.__getitem__ (stardist/models/model2d.py:63-104,
model3d.py:66-104) without shape completion:
    negative labels -> background, remembered in `mask_neg_labels` (loss disabled there: prob = -1)
    prob          = edt_prob(lbl[b][::grid])  in 2D                (model2d.py:86: subsampled first, then the distance transform)
                    edt_prob(lbl, anisotropy)[b][::grid]  in 3D    (model3d.py:88: the distance transform at FULL resolution, then subsampled)
    dist          = star_dist(lbl, n_rays, grid=grid)              (radial distances, ..., n_rays)
    dist_and_mask = [dist | prob]                                  (..., n_rays + 1: the mask channel weights the distance loss)
The training loop itself (losses, optimiser, augmentation, patch sampling) is out of scope."""
import numpy as np


def stardist_targets(labels, n_rays=32, grid=None, rays=None, b=None, anisotropy=None):
    """labels: sequence of integer label images of one shape (2D: (H, W); 3D: (Z, Y, X), then `rays` must be a Rays_* object).
    grid: subsampling per axis (powers of two).  b: optional tuple of slices cropping the border before subsampling (the generators'
    `self.b`).  Returns (prob (n, ..., 1), dist_and_mask (n, ..., n_rays + 1)) as float32 numpy arrays."""
    from .geometry.geom2d import star_dist
    from .geometry.geom3d import star_dist3D
    from .utils import _normalize_grid, edt_prob
    Y = [np.asarray(y) for y in labels]
    nd = Y[0].ndim
    if nd not in (2, 3) or any(y.shape != Y[0].shape for y in Y):
        raise ValueError("labels must be 2D or 3D images of one shape")
    if nd == 3 and rays is None:
        raise ValueError("3D targets need `rays`")
    grid = _normalize_grid((1,) * nd if grid is None else grid, nd)
    ss = tuple(slice(0, None, g) for g in grid)
    b = (slice(None),) * nd if b is None else tuple(b)
    neg = np.stack([y[b][ss] < 0 for y in Y])
    if neg.any():
        Y = [np.maximum(y, 0) for y in Y]
    if nd == 2:
        prob = np.stack([edt_prob(y[b][ss], anisotropy=anisotropy) for y in Y])
    else:                       # StarDistData3D: EDT of the whole volume first, border crop and grid subsampling second (model3d.py:88)
        prob = np.stack([edt_prob(y, anisotropy=anisotropy)[b][ss] for y in Y])
    if nd == 2:
        dist = np.stack([star_dist(y, n_rays, grid=grid, mode="hip")[b] for y in Y]) if b == (slice(None),) * nd else \
            np.stack([star_dist(y, n_rays, mode="hip")[b + (slice(None),)][ss] for y in Y])
        R = n_rays
    else:
        dist = np.stack([star_dist3D(y, rays, grid=grid, mode="hip") for y in Y]) if b == (slice(None),) * nd else \
            np.stack([star_dist3D(y, rays, mode="hip")[b + (slice(None),)][ss] for y in Y])
        R = len(rays)
    prob = prob[..., None].astype(np.float32)
    out = np.empty(dist.shape[:-1] + (R + 1,), np.float32)
    out[..., :-1] = dist
    out[..., -1:] = prob
    if neg.any():
        prob[neg] = -1
    return prob, out
