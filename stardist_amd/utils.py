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


# ----------------------------------------------------------------------------- training-side target: edt_prob
def edt_prob(lbl_img, anisotropy=None):
    """Per-object normalised Euclidean distance transform (stardist/utils.py:71-125; the scipy variant `_edt_prob_scipy`,
    which the reference uses when the optional `edt` package is absent): for every label id the EDT of its mask (computed
    on the object's bounding box grown by one pixel where it does not touch the image border), divided by its maximum."""
    import warnings
    from scipy.ndimage import distance_transform_edt, find_objects

    def grow(sl, interior):
        return tuple(slice(s.start - int(w[0]), s.stop + int(w[1])) for s, w in zip(sl, interior))

    def shrink(interior):
        return tuple(slice(int(w[0]), (-1 if w[1] else None)) for w in interior)
    lbl_img = np.asarray(lbl_img)
    constant_img = lbl_img.min() == lbl_img.max() and lbl_img.flat[0] > 0
    if constant_img:
        lbl_img = np.pad(lbl_img, ((1, 1),) * lbl_img.ndim, mode="constant")
        warnings.warn("EDT of constant label image is ill-defined. (Assuming background around it.)")
    objects = find_objects(lbl_img)
    prob = np.zeros(lbl_img.shape, np.float32)
    for i, sl in enumerate(objects, 1):
        if sl is None:
            continue
        interior = [(s.start > 0, s.stop < sz) for s, sz in zip(sl, lbl_img.shape)]
        shrink_slice = shrink(interior)
        grown_mask = lbl_img[grow(sl, interior)] == i
        mask = grown_mask[shrink_slice]
        edt = distance_transform_edt(grown_mask, sampling=anisotropy)[shrink_slice][mask]
        prob[sl][mask] = edt / (np.max(edt) + 1e-10)
    if constant_img:
        prob = prob[(slice(1, -1),) * lbl_img.ndim].copy()
    return prob


# ----------------------------------------------------------------------------- ImageJ ROI export (stardist/utils.py:195-268)
def polyroi_bytearray(x, y, pos=None, subpixel=True):
    """Byte array of an ImageJ polygon ROI (RoiDecoder.java layout): 64-byte big-endian header ('Iout', version 227, type 0,
    bbox, n, subpixel flag 128 at byte 50, position at 56), int16 coordinates relative to the bbox, then float32 sub-pixel
    coordinates.  ImageJ's pixel centre is (0.5, 0.5), hence the +0.5."""
    import struct

    def _int16(v): return int(v).to_bytes(2, byteorder="big", signed=True)

    def _uint16(v): return int(v).to_bytes(2, byteorder="big", signed=False)

    def _int32(v): return int(v).to_bytes(4, byteorder="big", signed=True)
    subpixel = bool(subpixel)
    x_raw = np.asarray(x).ravel() + 0.5
    y_raw = np.asarray(y).ravel() + 0.5
    x = np.round(x_raw); y = np.round(y_raw)
    assert len(x) == len(y)
    top, left, bottom, right = y.min(), x.min(), y.max(), x.max()
    n = len(x)
    header = 64
    B = bytearray(header + n * 2 * 2 + subpixel * n * 2 * 4)
    B[0:4] = b"Iout"
    B[4:6] = _int16(227); B[6:8] = _int16(0)
    B[8:10] = _int16(top); B[10:12] = _int16(left); B[12:14] = _int16(bottom); B[14:16] = _int16(right)
    B[16:18] = _uint16(n)
    if subpixel:
        B[50:52] = _int16(128)
    if pos is not None:
        B[56:60] = _int32(pos)
    for i, (_x, _y) in enumerate(zip(x, y)):
        xs = header + 2 * i
        ys = xs + 2 * n
        B[xs:xs + 2] = _int16(_x - left)
        B[ys:ys + 2] = _int16(_y - top)
    if subpixel:
        base1 = header + n * 2 * 2
        base2 = base1 + n * 4
        for i, (_x, _y) in enumerate(zip(x_raw, y_raw)):
            B[base1 + 4 * i:base1 + 4 * i + 4] = struct.pack(">f", _x)
            B[base2 + 4 * i:base2 + 4 * i + 4] = struct.pack(">f", _y)
    return B


def export_imagej_rois(fname, polygons, set_position=True, subpixel=True, compression=None):
    """polygons: array (n, 2, n_rays) (the `coord` entry of predict_instances' dict: [y, x]) or a list of such arrays, one
    per image position; writes <fname>.zip with one '<pos>_<i>.roi' per polygon (stardist/utils.py:254-268)"""
    from pathlib import Path
    from zipfile import ZIP_DEFLATED, ZipFile
    if compression is None:
        compression = ZIP_DEFLATED
    if isinstance(polygons, np.ndarray):
        polygons = (polygons,)
    fname = Path(fname)
    if fname.suffix == ".zip":
        fname = fname.with_suffix("")
    with ZipFile(str(fname) + ".zip", mode="w", compression=compression) as roizip:
        for pos, polygroup in enumerate(polygons, start=1):
            for i, poly in enumerate(polygroup, start=1):
                roi = polyroi_bytearray(poly[1], poly[0], pos=(pos if set_position else None), subpixel=subpixel)
                roizip.writestr("{pos:03d}_{i:03d}.roi".format(pos=pos, i=i), bytes(roi))
