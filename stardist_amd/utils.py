"""Small host-side helpers mirrored from the reference (stardist/utils.py, csbdeep.utils)."""
import numpy as np


def _normalize_grid(grid, n):
    """stardist/utils.py:60-68"""
    try:
        grid = tuple(grid)
        (len(grid) == n and all(map(np.isscalar, grid)) and all(map(_is_power_of_2, grid))) or _raise(TypeError())
        return tuple(int(g) for g in grid)
    except (TypeError, AssertionError):
        raise ValueError("grid = {grid} must be a list/tuple of length {n} with values that are power of 2".format(grid=grid, n=n))


def _is_power_of_2(i):
    assert i > 0
    e = np.log2(i)
    return e == int(e)


def _raise(e):
    raise e


def normalize_mi_ma(x, mi, ma, clip=False, eps=1e-20, dtype=np.float32):
    """csbdeep.utils.normalize_mi_ma (csbdeep>=0.8.0, caller side of predict_instances)."""
    if dtype is not None:
        x = x.astype(dtype, copy=False)
        mi = dtype(mi) if np.isscalar(mi) else mi.astype(dtype, copy=False)
        ma = dtype(ma) if np.isscalar(ma) else ma.astype(dtype, copy=False)
        eps = dtype(eps)
    x = (x - mi) / (ma - mi + eps)
    if clip:
        x = np.clip(x, 0, 1)
    return x


def normalize(x, pmin=3, pmax=99.8, axis=None, clip=False, eps=1e-20, dtype=np.float32):
    """csbdeep.utils.normalize: percentile-based image normalization."""
    mi = np.percentile(x, pmin, axis=axis, keepdims=True)
    ma = np.percentile(x, pmax, axis=axis, keepdims=True)
    return normalize_mi_ma(x, mi, ma, clip=clip, eps=eps, dtype=dtype)


def to_host(t):
    """device tensor -> numpy array through a page-locked staging tensor (PyTorch's caching host allocator re-uses the
    pinned block once the returned array is dropped): ~50 GB/s over PCIe instead of the pageable path's ~5 GB/s."""
    import torch
    if not t.is_cuda:
        return t.numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy()
