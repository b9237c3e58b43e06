"""Mirror of the reference module `stardist.lib.stardist2d` (stardist/lib/stardist2d.cpp:621-646).

Same function names, positional signatures, dtypes and return values; the work runs in
libstardist_hip.so on the GPU.  numpy in -> numpy out (host entry points);
torch CUDA tensors in -> torch CUDA tensors out (device entry points, current stream).
"""
import numpy as np

from . import _native as N


def c_non_max_suppression_inds(dist, points, use_kdtree, use_bbox, verbose, threshold, return_stats=False, _as_uint8=False):
    """stardist2d.cpp:390-615. dist (n,R) f32, points (n,2) f32 sorted by score desc -> bool (n,).
    _as_uint8 (device tensors, internal): the flags as the native wrote them (uint8 0 / 1) without the cast to bool."""
    N.require_device()
    stats = np.zeros(16, np.int64)
    if N.is_torch(dist):
        import torch
        assert dist.dtype == torch.float32 and points.dtype == torch.float32
        dist = dist.contiguous(); points = points.contiguous()
        n, R = dist.shape
        keep = torch.empty(n, dtype=torch.uint8, device=dist.device)
        N.dcall(dist, "sd_nms2d_device", N.tptr(dist), N.tptr(points), n, R, int(use_kdtree), int(use_bbox),
                                        int(verbose), float(threshold), N.tptr(keep), N.ptr(stats))
        if not _as_uint8:
            keep = keep.bool()
    else:
        dist = np.ascontiguousarray(dist, np.float32)
        points = np.ascontiguousarray(points, np.float32)
        if dist.ndim != 2 or points.ndim != 2 or points.shape[1] != 2 or len(points) != len(dist):
            raise ValueError("dist must be (n,n_rays) and points (n,2)")
        n, R = dist.shape
        keep = np.zeros(n, np.uint8)
        N.check(N.lib().sd_nms2d_host(N.ptr(dist), N.ptr(points), n, R, int(use_kdtree), int(use_bbox),
                                      int(verbose), float(threshold), N.ptr(keep), N.ptr(stats)))
        keep = keep.astype(bool)
    N.last_stats["nms2d"] = stats
    return (keep, stats) if return_stats else keep


def c_non_max_suppression_inds_old(polys, mapping, threshold, max_bbox_search, grid_y, grid_x, verbose):
    """stardist2d.cpp:173-386 ("O!O!fiiii"). polys (n,2,R) int32 (row 0 = y, row 1 = x) sorted by score desc, mapping (H,W) int32
    (pixel -> polygon id, -1 = none; an empty array without max_bbox_search, nms.py:56-60) -> bool (n,)."""
    N.require_device()
    if N.is_torch(polys):
        import torch
        assert polys.dtype == torch.int32 and polys.dim() == 3 and polys.shape[1] == 2
        polys = polys.contiguous()
        mapping = mapping.to(torch.int32).contiguous()
        n, _, R = polys.shape
        H, W = (int(mapping.shape[0]), int(mapping.shape[1])) if mapping.dim() == 2 else (0, 0)
        keep = torch.empty(n, dtype=torch.uint8, device=polys.device)
        if n:
            N.dcall(polys, "sd_nms2d_old_device", N.tptr(polys), n, R, N.tptr(mapping) if mapping.numel() else None, H, W, float(threshold),
                    int(max_bbox_search), int(grid_y), int(grid_x), int(verbose), N.tptr(keep))
        return keep.bool()
    polys = np.ascontiguousarray(polys)
    mapping = np.ascontiguousarray(mapping)
    if polys.dtype != np.int32 or mapping.dtype != np.int32:
        # the reference reads both buffers as int whatever their dtype (PyArray_GETPTR + cast, :227-228, :303); its caller always
        # passes int32 (nms.py:56-66) -- refuse anything else instead of reinterpreting bytes
        raise TypeError("polys and mapping must be int32")
    if polys.ndim != 3 or polys.shape[1] != 2:
        raise ValueError("polys must be (n, 2, n_rays)")
    n, _, R = polys.shape
    H, W = (mapping.shape[0], mapping.shape[1]) if mapping.ndim == 2 else (0, 0)
    keep = np.zeros(n, np.uint8)
    if n:
        N.check(N.lib().sd_nms2d_old_host(N.ptr(polys), n, R, N.ptr(mapping) if mapping.size else None, int(H), int(W), float(threshold),
                                          int(max_bbox_search), int(grid_y), int(grid_x), int(verbose), N.ptr(keep)))
    return keep.astype(bool)


def c_star_dist(src, n_rays, grid_y, grid_x):
    """stardist2d.cpp:55-124. src (H,W) uint16 -> (ceil(H/gy), ceil(W/gx), n_rays) f32."""
    N.require_device()
    n_rays, grid_y, grid_x = int(n_rays), int(grid_y), int(grid_x)
    if N.is_torch(src):
        import torch
        assert src.dtype in (torch.uint16, torch.int16) and src.dim() == 2
        src = src.contiguous()
        H, W = src.shape
        dst = torch.empty(((H - 1) // grid_y + 1, (W - 1) // grid_x + 1, n_rays), dtype=torch.float32, device=src.device)
        N.dcall(src, "sd_star_dist2d_device", N.tptr(src), H, W, n_rays, grid_y, grid_x, N.tptr(dst))
        return dst
    src = np.ascontiguousarray(src)
    if src.dtype != np.uint16:
        # the reference reads the buffer as unsigned short whatever the dtype (stardist2d.cpp:83);
        # its Python wrapper always casts (geom2d.py:31) -- do the cast here.
        src = src.astype(np.uint16)
    H, W = src.shape
    dst = np.empty(((H - 1) // grid_y + 1, (W - 1) // grid_x + 1, n_rays), np.float32)
    N.check(N.lib().sd_star_dist2d_host(N.ptr(src), H, W, n_rays, grid_y, grid_x, N.ptr(dst)))
    return dst


def c_polygons_to_label(coord, labels, shape, window=None):
    """New native for the reference's Python rasteriser loop (geom2d.py:149-166).
    coord (n,2,R) f32 painted in order, value labels[i]+1; returns int32 (H,W).
    window = ((y0, x0), (h, w)) (device tensors only): just that part of the `shape` image is rendered and returned."""
    N.require_device()
    H, W = int(shape[0]), int(shape[1])
    if N.is_torch(coord):
        import torch
        coord = coord.contiguous().float()
        labels = labels.contiguous().to(torch.int32)
        n, _, R = coord.shape
        (y0, x0), (h, w) = ((0, 0), (H, W)) if window is None else window
        out = torch.empty((int(h), int(w)), dtype=torch.int32, device=coord.device)
        N.dcall(coord, "sd_polygons_to_label_window_device", N.tptr(coord), N.tptr(labels), n, R, H, W, int(y0), int(x0), int(h), int(w), N.tptr(out))
        return out
    if window is not None:
        raise ValueError("window rendering takes device tensors")
    coord = np.ascontiguousarray(coord, np.float32)
    labels = np.ascontiguousarray(labels, np.int32)
    n = coord.shape[0]
    R = coord.shape[2] if coord.ndim == 3 else 0
    out = np.zeros((H, W), np.int32)
    if n:
        N.check(N.lib().sd_polygons_to_label_host(N.ptr(coord), N.ptr(labels), n, R, H, W, N.ptr(out)))
    return out


def survivors_of_sorted(keep, prob, points, dist, want_paint=True):
    """What model2d.py:536-561 does with the keep flags of candidates that are in SCORE order, on the device and in two native calls
    (csrc/survivors.hip): returns (prob (m,), points (m, 2) int64, coord (m, 2, R) float32, coord_paint, labels_paint) -- coord_paint /
    labels_paint (int32, label id - 1) are the polygons in the order polygons_to_label paints them (ascending score, stable:
    geom2d.py:186-197), ready for c_polygons_to_label; None without want_paint.  keep: uint8 / bool device tensor (n,); prob (n,)
    float32, points (n, 2) int64, dist (n, R) float32, all on keep's device.  One read-back (the survivor count)."""
    import torch
    from ..geometry.geom2d import ray_angles, _SC_CACHE
    N.require_device()
    assert keep.is_cuda and prob.dtype == torch.float32 and dist.dtype == torch.float32 and points.dtype == torch.int64 and points.shape[1] == 2
    n, R = int(dist.shape[0]), int(dist.shape[1])
    assert keep.numel() == n == prob.numel() == points.shape[0]
    dev = keep.device
    keep = keep.contiguous().view(torch.uint8) if keep.dtype == torch.bool else keep.contiguous()
    prob, points, dist = prob.contiguous(), points.contiguous(), dist.contiguous()
    pos = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    N.dcall(keep, "sd_survivor_positions_device", N.tptr(keep), n, N.tptr(pos), N.tptr(cnt))
    m = int(cnt.item())
    key = (R, str(dev))
    sct = _SC_CACHE.get(key)
    if sct is None:
        phis = ray_angles(R)
        sct = _SC_CACHE[key] = torch.as_tensor(np.ascontiguousarray(np.array([np.sin(phis), np.cos(phis)])), device=dev)
    oprob = torch.empty(m, dtype=torch.float32, device=dev)
    opts = torch.empty((m, 2), dtype=torch.int64, device=dev)
    coord = torch.empty((m, 2, R), dtype=torch.float32, device=dev)
    cpaint = torch.empty((m, 2, R), dtype=torch.float32, device=dev) if want_paint else None
    lpaint = torch.empty(m, dtype=torch.int32, device=dev) if want_paint else None
    if m:
        N.dcall(keep, "sd_survivors2d_device", N.tptr(pos), m, N.tptr(prob), N.tptr(points), N.tptr(dist), R, N.tptr(sct), N.tptr(oprob), N.tptr(opts),
                N.tptr(coord), N.tptr(cpaint) if want_paint else None, N.tptr(lpaint) if want_paint else None)
    return oprob, opts, coord, cpaint, lpaint


def clip_pairs(xa, ya, xb, yb):
    """Pair-level probe (tests): 2*area of A∩B per pair as the reference's Clipper call sums it.
    xa..yb int32 (n_pairs, n_verts) numpy arrays. Returns (twice_area int64, flags int32)."""
    import torch
    N.require_device()
    dev = torch.device("cuda")
    t = [torch.from_numpy(np.ascontiguousarray(v, np.int32)).to(dev) for v in (xa, ya, xb, yb)]
    n, R = t[0].shape
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    fl = torch.zeros(n, dtype=torch.int32, device=dev)
    N.dcall(t[0], "sd_clip_pairs_device", N.tptr(t[0]), N.tptr(t[1]), N.tptr(t[2]), N.tptr(t[3]), n, R,
                                         N.tptr(out), N.tptr(fl))
    torch.cuda.synchronize()
    return out.cpu().numpy(), fl.cpu().numpy()


def area_bounds_pairs(xa, ya, xb, yb):
    """Pair-level probe (tests) of the 2D NMS's decision shortcut (csrc/area_bounds.h): (exact intersection area float32, half-width of the
    band enclosing Clipper's area, usable bool, number of boundary crossings, number of near edge pairs) per pair.  xa..yb int32 (n_pairs, n_verts <= 32)."""
    import torch
    N.require_device()
    dev = torch.device("cuda")
    t = [torch.from_numpy(np.ascontiguousarray(v, np.int32)).to(dev) for v in (xa, ya, xb, yb)]
    n, R = t[0].shape
    area = torch.zeros(n, dtype=torch.float32, device=dev); band = torch.zeros(n, dtype=torch.float32, device=dev)
    info = torch.zeros(n, dtype=torch.int32, device=dev)
    N.dcall(t[0], "sd_area_bounds_pairs_device", N.tptr(t[0]), N.tptr(t[1]), N.tptr(t[2]), N.tptr(t[3]), n, R, N.tptr(area), N.tptr(band), N.tptr(info))
    torch.cuda.synchronize()
    info = info.cpu().numpy()
    return area.cpu().numpy(), band.cpu().numpy(), (info & 1).astype(bool), (info >> 8) & 0x3FF, (info >> 18) & 0x7FF
