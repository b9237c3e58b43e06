"""Mirror of the reference module `stardist.lib.stardist3d` (stardist/lib/stardist3d.cpp:351-392).

Same function names and positional signatures (note: use_bbox comes BEFORE use_kdtree here,
unlike the 2D module -- stardist3d.cpp:23).  numpy in -> numpy out through the reference's own
C-ABI names (_LIB_*), torch CUDA tensors in -> torch CUDA tensors out through the *_device
entry points on the current stream.
"""
import ctypes

import numpy as np

from . import _native as N


def c_non_max_suppression_inds(dist, points, verts, faces, scores, use_bbox, use_kdtree, verbose, threshold, return_stats=False):
    """stardist3d.cpp:13-62 -> stardist3d_impl.cpp:956-1385. Inputs sorted by score descending. Returns bool (n,)."""
    N.require_device()
    stats = np.zeros(16, np.int64)
    if N.is_torch(dist):
        import torch
        dist = dist.contiguous().float(); points = points.contiguous().float()
        verts = verts.contiguous().float(); faces = faces.contiguous().to(torch.int32); scores = scores.contiguous().float()
        n, R = dist.shape
        keep = torch.empty(n, dtype=torch.uint8, device=dist.device)
        if n:
            N.check(N.lib().sd_nms3d_device(N.tptr(scores), N.tptr(dist), N.tptr(points), n, R, faces.shape[0], N.tptr(verts),
                                            N.tptr(faces), float(threshold), int(use_bbox), int(use_kdtree), int(verbose),
                                            N.tptr(keep), N.ptr(stats), N.current_stream()))
        keep = keep.bool()
        N.last_stats["nms3d"] = stats
        return (keep, stats) if return_stats else keep
    dist = np.ascontiguousarray(dist, np.float32); points = np.ascontiguousarray(points, np.float32)
    verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    scores = np.ascontiguousarray(scores, np.float32)
    n, R = dist.shape
    keep = np.zeros(n, np.bool_)
    if n:
        N.lib()._LIB_non_maximum_suppression_sparse(N.ptr(scores), N.ptr(dist), N.ptr(points), n, R, faces.shape[0], N.ptr(verts),
                                                    N.ptr(faces), float(threshold), int(use_bbox), int(use_kdtree), int(verbose),
                                                    keep.ctypes.data_as(ctypes.c_void_p))
    return (keep, stats) if return_stats else keep


def c_polyhedron_to_label(dist, points, verts, faces, labels, render_mode, verbose, use_overlap_label, overlap_label, shape):
    """stardist3d.cpp:82-143 -> stardist3d_impl.cpp:1404-1525. Returns a new zero-initialised int32 (nz,ny,nx) volume."""
    N.require_device()
    nz, ny, nx = (int(v) for v in shape)
    if N.is_torch(dist):
        import torch
        dist = dist.contiguous().float(); points = points.contiguous().float()
        verts = verts.contiguous().float(); faces = faces.contiguous().to(torch.int32); labels = labels.contiguous().to(torch.int32)
        out = torch.zeros((nz, ny, nx), dtype=torch.int32, device=dist.device)
        if dist.shape[0]:
            N.check(N.lib().sd_polyhedron_to_label_device(N.tptr(dist), N.tptr(points), N.tptr(verts), N.tptr(faces), dist.shape[0],
                                                          dist.shape[1], faces.shape[0], N.tptr(labels), nz, ny, nx, int(render_mode),
                                                          int(verbose), int(use_overlap_label), int(overlap_label), N.tptr(out),
                                                          N.current_stream()))
        return out
    dist = np.ascontiguousarray(dist, np.float32); points = np.ascontiguousarray(points, np.float32)
    verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    labels = np.ascontiguousarray(labels, np.int32)
    out = np.zeros((nz, ny, nx), np.int32)
    if dist.shape[0]:
        if int(render_mode) == 2:
            raise N.NativeError("render mode 'hull' needs the convex hull (Qhull) and is not implemented")
        N.lib()._LIB_polyhedron_to_label(N.ptr(dist), N.ptr(points), N.ptr(verts), N.ptr(faces), dist.shape[0], dist.shape[1],
                                         faces.shape[0], N.ptr(labels), nz, ny, nx, int(render_mode), int(verbose),
                                         int(use_overlap_label), int(overlap_label), N.ptr(out))
    return out


def c_star_dist3d(src, pdz, pdy, pdx, n_rays, grid_z, grid_y, grid_x):
    """stardist3d.cpp:245-346. src (Z,Y,X) uint16 -> (ceil(Z/gz), ceil(Y/gy), ceil(X/gx), n_rays) float32."""
    N.require_device()
    n_rays, gz, gy, gx = int(n_rays), int(grid_z), int(grid_y), int(grid_x)
    if N.is_torch(src):
        import torch
        src = src.contiguous()
        Z, Y, X = src.shape
        dev = src.device
        pdz, pdy, pdx = (torch.as_tensor(v, dtype=torch.float32, device=dev).contiguous() for v in (pdz, pdy, pdx))
        dst = torch.empty(((Z - 1) // gz + 1, (Y - 1) // gy + 1, (X - 1) // gx + 1, n_rays), dtype=torch.float32, device=dev)
        N.check(N.lib().sd_star_dist3d_device(N.tptr(src), Z, Y, X, N.tptr(pdz), N.tptr(pdy), N.tptr(pdx), n_rays, gz, gy, gx,
                                              N.tptr(dst), N.current_stream()))
        return dst
    src = np.ascontiguousarray(src)
    if src.dtype != np.uint16:
        src = src.astype(np.uint16)
    pdz, pdy, pdx = (np.ascontiguousarray(v, np.float32) for v in (pdz, pdy, pdx))
    Z, Y, X = src.shape
    dst = np.empty(((Z - 1) // gz + 1, (Y - 1) // gy + 1, (X - 1) // gx + 1, n_rays), np.float32)
    N.check(N.lib().sd_star_dist3d_host(N.ptr(src), Z, Y, X, N.ptr(pdz), N.ptr(pdy), N.ptr(pdx), n_rays, gz, gy, gx, N.ptr(dst)))
    return dst
