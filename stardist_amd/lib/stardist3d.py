"""Mirror of the reference module `stardist.lib.stardist3d` (stardist/lib/stardist3d.cpp:351-392).

Same function names and positional signatures (note: use_bbox comes BEFORE use_kdtree here,
unlike the 2D module -- stardist3d.cpp:23).  numpy in -> numpy out through the reference's own
C-ABI names (_LIB_*), torch CUDA tensors in -> torch CUDA tensors out through the *_device
entry points on the current stream.
"""
import ctypes

import numpy as np

from . import _native as N


def c_non_max_suppression_inds(dist, points, verts, faces, scores, use_bbox, use_kdtree, verbose, threshold, return_stats=False, _as_uint8=False):
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
            N.dcall(scores, "sd_nms3d_device", N.tptr(scores), N.tptr(dist), N.tptr(points), n, R, faces.shape[0], N.tptr(verts),
                                            N.tptr(faces), float(threshold), int(use_bbox), int(use_kdtree), int(verbose),
                                            N.tptr(keep), N.ptr(stats))
        if not _as_uint8:
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


def c_polyhedron_to_label(dist, points, verts, faces, labels, render_mode, verbose, use_overlap_label, overlap_label, shape, window=None):
    """stardist3d.cpp:82-143 -> stardist3d_impl.cpp:1404-1525. Returns a new zero-initialised int32 (nz,ny,nx) volume.
    window = ((z0, y0, x0), (nz, ny, nx)) (device tensors only): just that part of the `shape` volume is rendered and returned."""
    N.require_device()
    nz, ny, nx = (int(v) for v in shape)
    if N.is_torch(dist):
        import torch
        dist = dist.contiguous().float(); points = points.contiguous().float()
        verts = verts.contiguous().float(); faces = faces.contiguous().to(torch.int32); labels = labels.contiguous().to(torch.int32)
        (z0, y0, x0), (wz, wy, wx) = ((0, 0, 0), (nz, ny, nx)) if window is None else window
        out = torch.zeros((wz, wy, wx), dtype=torch.int32, device=dist.device)
        if dist.shape[0]:
            N.dcall(dist, "sd_polyhedron_to_label_window_device", N.tptr(dist), N.tptr(points), N.tptr(verts), N.tptr(faces), dist.shape[0],
                    dist.shape[1], faces.shape[0], N.tptr(labels), nz, ny, nx, int(z0), int(y0), int(x0), int(wz), int(wy), int(wx),
                    int(render_mode), int(verbose), int(use_overlap_label), int(overlap_label), N.tptr(out))
        return out
    if window is not None:
        raise ValueError("window rendering takes device tensors")
    dist = np.ascontiguousarray(dist, np.float32); points = np.ascontiguousarray(points, np.float32)
    verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    labels = np.ascontiguousarray(labels, np.int32)
    out = np.zeros((nz, ny, nx), np.int32)
    if dist.shape[0]:
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
        N.dcall(src, "sd_star_dist3d_device", N.tptr(src), Z, Y, X, N.tptr(pdz), N.tptr(pdy), N.tptr(pdx), n_rays, gz, gy, gx,
                                              N.tptr(dst))
        return dst
    src = np.ascontiguousarray(src)
    if src.dtype != np.uint16:
        src = src.astype(np.uint16)
    pdz, pdy, pdx = (np.ascontiguousarray(v, np.float32) for v in (pdz, pdy, pdx))
    Z, Y, X = src.shape
    dst = np.empty(((Z - 1) // gz + 1, (Y - 1) // gy + 1, (X - 1) // gx + 1, n_rays), np.float32)
    N.check(N.lib().sd_star_dist3d_host(N.ptr(src), Z, Y, X, N.ptr(pdz), N.ptr(pdy), N.ptr(pdx), n_rays, gz, gy, gx, N.ptr(dst)))
    return dst


# ---- analysis helpers of the same native module (not on the prediction path): tensor formulation on the device
def _tet_terms(dist, verts, faces):
    """signed volumes of the tetrahedra (origin, A, B, C) of every face and the vertex sums A+B+C; dist (..., R) float32 tensor.
    tetrahedron_volume / polyhedron_volume, stardist3d_impl.cpp:234-291 (fp32; the face sum is a tree sum here, sequential there)."""
    P = dist[..., None] * verts                                   # (..., R, 3) in (z, y, x)
    A, B, C = P[..., faces[:, 0], :], P[..., faces[:, 1], :], P[..., faces[:, 2], :]
    M0, M1, M2 = B - A, C - A, -A
    det = (M0[..., 0] * (M1[..., 1] * M2[..., 2] - M2[..., 1] * M1[..., 2])
           - M0[..., 1] * (M1[..., 0] * M2[..., 2] - M1[..., 2] * M2[..., 0])
           + M0[..., 2] * (M1[..., 0] * M2[..., 1] - M1[..., 1] * M2[..., 0]))
    return det / 6.0, A + B + C


def _dist_to_volume_t(dist, verts, faces, chunk=8):
    import torch
    out = torch.empty(dist.shape[:-1], dtype=torch.float32, device=dist.device)
    for z in range(0, dist.shape[0], chunk):
        out[z:z + chunk] = _tet_terms(dist[z:z + chunk], verts, faces)[0].sum(-1)
    return out


def _dist_to_centroid_t(dist, verts, faces, absolute, chunk=8):
    import torch
    out = torch.empty(dist.shape[:-1] + (3,), dtype=torch.float32, device=dist.device)
    for z in range(0, dist.shape[0], chunk):
        vol_f, s = _tet_terms(dist[z:z + chunk], verts, faces)
        vol = vol_f.sum(-1)
        R = (0.25 * s * vol_f[..., None]).sum(-2)                  # stardist3d_impl.cpp:327-329
        c = torch.where((vol > 1e-10)[..., None], R / vol[..., None].clamp_min(1e-30), torch.zeros_like(R))   # :335-337
        out[z:z + chunk] = c
    if absolute:                                                   # :1581-1583
        Z, Y, X = dist.shape[:3]
        grid = torch.stack(torch.meshgrid(torch.arange(Z, device=dist.device), torch.arange(Y, device=dist.device),
                                          torch.arange(X, device=dist.device), indexing="ij"), -1).to(torch.float32)
        out = out + grid
    return out


def _analysis_call(fn, dist, verts, faces, *extra):
    import torch
    N.require_device()
    as_np = not N.is_torch(dist)
    dev = torch.device("cuda") if as_np else dist.device
    d = (torch.from_numpy(np.ascontiguousarray(dist, np.float32)) if as_np else dist.float()).to(dev)
    if d.dim() != 4:
        raise ValueError("dist.ndim = %d but should be 4" % d.dim())
    v = torch.as_tensor(np.asarray(verts.cpu() if N.is_torch(verts) else verts), dtype=torch.float32, device=dev)
    f = torch.as_tensor(np.asarray(faces.cpu() if N.is_torch(faces) else faces), dtype=torch.int64, device=dev)
    out = fn(d, v, f, *extra)
    return out.cpu().numpy() if as_np else out


def c_dist_to_volume(dist, verts, faces):
    """stardist3d.cpp:148-194 -> _COMMON_dist_to_volume (stardist3d_impl.cpp:1529-1556): dist (Z,Y,X,R) -> volumes (Z,Y,X) float32."""
    return _analysis_call(_dist_to_volume_t, dist, verts, faces)


def c_dist_to_centroid(dist, verts, faces, absolute):
    """stardist3d.cpp:196-243 -> _COMMON_dist_to_centroid (:1558-1589): dist (Z,Y,X,R) -> centroids (Z,Y,X,3) float32 (z,y,x),
    relative to the voxel or (absolute=1) in volume coordinates."""
    return _analysis_call(_dist_to_centroid_t, dist, verts, faces, int(absolute))


def hiv_pair_volumes(dist, points, verts, faces, pairs, kernel=True, hull=True):
    """Pair-level probe (tests): intersection volume of the kernels / of the convex hulls of polyhedra pairs[:,0], pairs[:,1]
    as the 3D NMS cascade computes them (float64).  Returns (vol_kernel | None, vol_hull | None)."""
    import torch
    N.require_device()
    dev = torch.device("cuda")
    d = torch.from_numpy(np.ascontiguousarray(dist, np.float32)).to(dev); p = torch.from_numpy(np.ascontiguousarray(points, np.float32)).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(verts, np.float32)).to(dev); f = torch.from_numpy(np.ascontiguousarray(faces, np.int32)).to(dev)
    pr = torch.from_numpy(np.ascontiguousarray(pairs, np.int32)).to(dev)
    vk = torch.zeros(len(pairs), dtype=torch.float64, device=dev) if kernel else None
    vh = torch.zeros(len(pairs), dtype=torch.float64, device=dev) if hull else None
    N.dcall(d, "sd_hiv_pairs_device", N.tptr(d), N.tptr(p), d.shape[0], d.shape[1], f.shape[0], N.tptr(v), N.tptr(f), N.tptr(pr), len(pairs),
            N.tptr(vk) if kernel else None, N.tptr(vh) if hull else None)
    torch.cuda.synchronize()
    return (vk.cpu().numpy() if kernel else None), (vh.cpu().numpy() if hull else None)
