"""Small host-side helpers mirrored from the reference (stardist/utils.py, csbdeep.utils)."""
from zipfile import ZIP_DEFLATED

import numpy as np


def gputools_available():
    """stardist/utils.py:40-45 asks whether the OpenCL helpers can be imported (they serve the reference's mode='opencl' of the training targets).
    There is no OpenCL path here -- 'hip' is the one device mode -- so the answer is always False."""
    return False


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


def to_host_many(tensors):
    """several device tensors -> numpy arrays with ONE stream synchronisation: every copy goes through its own page-locked staging
    tensor (as to_host) and is enqueued non-blocking; None entries and host tensors pass through."""
    import torch
    outs, dev = [], None
    for t in tensors:
        if t is None or not t.is_cuda:
            outs.append(None if t is None else t.numpy())
            continue
        dev = t.device
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        outs.append(h)
    if dev is not None:
        torch.cuda.current_stream(dev).synchronize()
    return [o.numpy() if (o is not None and not isinstance(o, np.ndarray)) else o for o in outs]


def to_device(a, device):
    """host numpy array -> device tensor.  Measured on the MI355X box (tools/probe_h2d.py, 2048^2 float32 = 16.8 MB): the plain
    pageable copy 0.32 ms (52 GB/s), a persistent page-locked staging tensor 0.38 ms -- and filling that staging tensor with torch's
    multi-threaded copy_ left 128 worker threads spinning, which slowed every later host-synchronised stage of the step (the whole
    predict_instances went from 20.5 to 30 ms).  So: the runtime's own pageable path, nothing on top."""
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device=torch.device(device))


# ----------------------------------------------------------------------------- training-side target: edt_prob
def edt_prob(lbl_img, anisotropy=None):
    """Per-object normalised Euclidean distance transform of a label image (the object-probability training target; what
    stardist/utils.py:71-125 `edt_prob` returns) computed on the GPU by csrc/edt.hip: for every pixel of an object the distance
    (float64, axis spacing `anisotropy`) to the nearest pixel inside the image that carries another label, divided by the object's
    maximum; background 0.  numpy in -> numpy float32 out; a torch device tensor in -> a device tensor out.
    A constant non-zero image is ill-defined; like the reference it is treated as surrounded by background (with a warning)."""
    import ctypes
    import warnings
    import torch
    from .lib import _native as N
    N.require_device()
    as_np = not N.is_torch(lbl_img)
    # label ids: the kernel keeps one table entry per id up to the largest, in int32.  Ids beyond int32, or sparse huge ids (hash-like
    # labels), are first compacted to 1..n -- the result depends on which pixels share a label, not on the ids
    if as_np:
        a = np.ascontiguousarray(lbl_img)
        if a.size and (int(a.max()) >= 2 ** 31 or int(a.max()) > 4 * a.size + 1024):
            u, inv = np.unique(a, return_inverse=True)
            a = inv.reshape(a.shape).astype(np.int64) + (0 if (len(u) and u[0] == 0) else 1)
            if len(u) and u[0] < 0:
                raise ValueError("edt_prob: negative labels")
        lab = torch.from_numpy(a.astype(np.int32, copy=False)).cuda()
    else:
        t = lbl_img
        if t.numel() and (int(t.max()) >= 2 ** 31 or int(t.max()) > 4 * t.numel() + 1024):
            u, inv = torch.unique(t, return_inverse=True)
            t = inv.reshape(t.shape) + (0 if int(u[0]) == 0 else 1)
        lab = t.to(torch.int32).contiguous()
    nd = lab.dim()
    if nd not in (2, 3):
        raise ValueError("edt_prob: label image must be 2D or 3D")
    samp = (1.0,) * nd if anisotropy is None else tuple(float(a) for a in anisotropy)
    if len(samp) != nd:
        raise ValueError("edt_prob: anisotropy must have one entry per axis")
    if lab.numel() == 0:
        out = torch.zeros(lab.shape, dtype=torch.float32, device=lab.device)
        return out.cpu().numpy() if as_np else out
    lo, hi = int(lab.min()), int(lab.max())
    padded = lo == hi and lo > 0
    if padded:
        warnings.warn("EDT of constant label image is ill-defined. (Assuming background around it.)")
        lab = torch.nn.functional.pad(lab, (1, 1) * nd).contiguous()
    Z, Y, X = ((1,) + tuple(lab.shape)) if nd == 2 else tuple(lab.shape)
    sz, sy, sx = ((1.0,) + samp) if nd == 2 else samp
    out = torch.empty(lab.shape, dtype=torch.float32, device=lab.device)
    N.dcall(lab, "sd_edt_prob_device", ctypes.c_void_p(lab.data_ptr()), Z, Y, X, sz, sy, sx, max(hi, 0), ctypes.c_void_p(out.data_ptr()))
    if padded:
        out = out[(slice(1, -1),) * nd].contiguous()
    return out.cpu().numpy() if as_np else out


# ----------------------------------------------------------------------------- ImageJ ROI export
# File layout from ImageJ's ij/io/RoiDecoder.java (all big-endian): 64-byte header -- "Iout", int16 version (>= 217; 227 written),
# byte roi type (0 = polygon) + 1 unused byte, int16 top / left / bottom / right, uint16 n coordinates, ..., int16 options at 50
# (128 = SUB_PIXEL_RESOLUTION), ..., int32 position at 56 -- then n int16 x offsets from `left`, n int16 y offsets from `top` and,
# with the sub-pixel option, n float32 x and n float32 y absolute coordinates.  ImageJ puts the centre of pixel (0, 0) at (0.5, 0.5).
_ROI_HEADER = 64


def polyroi_bytearray(x, y, pos=None, subpixel=True):
    """one ImageJ polygon ROI as bytes (same arguments and bytes as stardist/utils.py:195-251)"""
    import struct
    fx = np.asarray(x, np.float64).ravel() + 0.5
    fy = np.asarray(y, np.float64).ravel() + 0.5
    if fx.shape != fy.shape:
        raise ValueError("x and y must have the same length")
    ix, iy = np.round(fx), np.round(fy)
    n = fx.size
    top, left, bottom, right = int(iy.min()), int(ix.min()), int(iy.max()), int(ix.max())
    head = bytearray(_ROI_HEADER)
    struct.pack_into(">4shBx4hH", head, 0, b"Iout", 227, 0, top, left, bottom, right, n)
    if subpixel:
        struct.pack_into(">h", head, 50, 128)
    if pos is not None:
        struct.pack_into(">i", head, 56, int(pos))
    body = (ix - left).astype(">i2").tobytes() + (iy - top).astype(">i2").tobytes()
    if subpixel:
        body += fx.astype(">f4").tobytes() + fy.astype(">f4").tobytes()
    return head + body


def export_imagej_rois(fname, polygons, set_position=True, subpixel=True, compression=ZIP_DEFLATED):
    """ImageJ ROI archive of predicted polygons (what stardist/utils.py:254-268 writes): `polygons` is the `coord` array of
    predict_instances' dict -- (n, 2, n_rays), rows (y, x) -- or a sequence of such arrays, one per stack position; entry
    'PPP_III.roi' holds polygon III of position PPP (both 1-based)."""
    import zipfile
    groups = (polygons,) if isinstance(polygons, np.ndarray) else polygons
    target = str(fname)
    if not target.endswith(".zip"):
        target += ".zip"
    with zipfile.ZipFile(target, mode="w", compression=ZIP_DEFLATED if compression is None else compression) as zf:
        for pos, group in enumerate(groups, start=1):
            for k, (ys, xs) in enumerate(group, start=1):
                zf.writestr("%03d_%03d.roi" % (pos, k), bytes(polyroi_bytearray(xs, ys, pos=pos if set_position else None, subpixel=subpixel)))


# ----------------------------------------------------------------------------- label-image helpers the reference package exports
# (stardist/__init__.py:13: fill_label_holes, sample_points, calculate_extents).  Host-side preparation / inspection of label images,
# not part of the prediction path; numpy + scipy like the reference's.

def _object_boxes(lbl_img):
    """[(label id, bounding-box slices)] of the labels present, ascending (scipy.ndimage.find_objects: entry i - 1 belongs to label i)"""
    from scipy.ndimage import find_objects
    return [(i, sl) for i, sl in enumerate(find_objects(lbl_img), 1) if sl is not None]


def fill_label_holes(lbl_img, **kwargs):
    """Fill the holes of every labelled object (stardist/utils.py:137-153): per object, `binary_fill_holes` of its mask on its bounding
    box grown by one pixel on every side that is not an image border -- a cavity that reaches the image border stays open, as there.
    **kwargs go to scipy.ndimage.binary_fill_holes."""
    from scipy.ndimage import binary_fill_holes
    lbl_img = np.asarray(lbl_img)
    out = np.zeros_like(lbl_img)
    for lab, box in _object_boxes(lbl_img):
        room = [(s.start > 0, s.stop < n) for s, n in zip(box, lbl_img.shape)]
        grown = tuple(slice(s.start - int(lo), s.stop + int(hi)) for s, (lo, hi) in zip(box, room))
        inner = tuple(slice(int(lo), -1 if hi else None) for lo, hi in room)
        filled = binary_fill_holes(lbl_img[grown] == lab, **kwargs)[inner]
        out[box][filled] = lab
    return out


def sample_points(n_samples, mask, prob=None, b=2):
    """Draw `n_samples` pixel positions (with replacement) from a 2D mask, at least `b` pixels from the border, uniformly or weighted by
    `prob` (stardist/utils.py:156-177; numpy's global random state, the same draws as there)."""
    mask = np.asarray(mask)
    if b is not None and b > 0:
        inner = np.zeros_like(mask)
        inner[b:-b, b:-b] = True
    else:
        inner = True
    rows, cols = np.nonzero(mask & inner)
    if prob is not None:
        w = np.asarray(prob)[rows, cols].astype(np.float64)
        w /= np.sum(w)
        pick = np.random.choice(len(rows), n_samples, replace=True, p=w)
    else:
        pick = np.random.choice(len(rows), n_samples, replace=True)
    return np.stack((rows[pick], cols[pick]), axis=-1)


def calculate_extents(lbl, func=np.median):
    """Aggregate (median by default) of the objects' bounding-box sizes per axis, for one 2D / 3D label image, or over a sequence / 4D
    stack of them (stardist/utils.py:180-193)."""
    from collections.abc import Iterable
    if (isinstance(lbl, np.ndarray) and lbl.ndim == 4) or (not isinstance(lbl, np.ndarray) and isinstance(lbl, Iterable)):
        return func(np.stack([calculate_extents(one, func) for one in lbl], axis=0), axis=0)
    n = lbl.ndim
    if n not in (2, 3):
        raise ValueError("label image should be 2- or 3-dimensional (or pass a list of these)")
    boxes = _object_boxes(lbl)
    if not boxes:
        return np.zeros(n)
    return func(np.array([[s.stop - s.start for s in box] for _, box in boxes]), axis=0)


def optimize_threshold(Y, Yhat, model, nms_thresh, measure="accuracy", iou_threshs=[0.3, 0.5, 0.7], bracket=None, tol=1e-2, maxiter=20, verbose=1):
    """Tune prob_thresh for a fixed nms_thresh so that `measure` of stardist.matching (averaged over iou_threshs) between the label images Y
    and the instances of the predictions Yhat = [(prob, dist), ...] is largest: golden-section search over [max prob / 2, max prob]
    (stardist/utils.py:271-307).  Returns (prob_thresh, value).  Host-side tool: every evaluation is one `_instances_from_prediction` per
    image (NMS + rasteriser natives).  verbose > 1 prints every evaluation; the reference's progress bar is not drawn."""
    import datetime
    from scipy.optimize import minimize_scalar
    from .matching import matching_dataset
    if not np.isscalar(nms_thresh):
        raise ValueError("nms_thresh must be a scalar")
    iou_threshs = [iou_threshs] if np.isscalar(iou_threshs) else iou_threshs
    if bracket is None:
        top = max([np.max(prob) for prob, dist in Yhat])
        bracket = top / 2, top
    seen = {}

    def negative_score(thr):
        prob_thresh = np.clip(thr, *bracket)
        value = seen.get(prob_thresh)
        if value is None:
            instances = [model._instances_from_prediction(y.shape, *prob_dist, prob_thresh=prob_thresh, nms_thresh=nms_thresh)[0] for y, prob_dist in zip(Y, Yhat)]
            stats = matching_dataset(Y, instances, thresh=iou_threshs, show_progress=False, parallel=True)
            seen[prob_thresh] = value = np.mean([s._asdict()[measure] for s in stats])
        if verbose > 1:
            print("%s   thresh: %f   %s: %f" % (datetime.datetime.now().strftime("%H:%M:%S"), prob_thresh, measure, value), flush=True)
        return -value
    opt = minimize_scalar(negative_score, method="golden", bracket=bracket, tol=tol, options={"maxiter": maxiter})
    if verbose > 1:
        print("\n", opt, flush=True)
    return opt.x, -opt.fun
