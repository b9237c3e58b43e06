"""Loader for the compiled reference natives in oracle/_ref (TEST INFRASTRUCTURE ONLY).

``stardist2d`` / ``stardist3d`` are the reference's CPython extension modules
(stardist/lib/stardist2d.cpp:621-646, stardist/lib/stardist3d.cpp:351-392) built by
oracle/Makefile from /root/reference; ``clipper`` wraps oracle/clipper_shim.cpp, a
pair-level probe around the vendored Clipper (call pattern of stardist2d.cpp:152-165).
"""
import ctypes
import importlib.machinery
import importlib.util
import os
import sysconfig

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_EXT = sysconfig.get_config_var("EXT_SUFFIX")


def available():
    return os.path.exists(os.path.join(_REF, "stardist2d" + _EXT)) and \
        os.path.exists(os.path.join(_REF, "stardist3d" + _EXT))


def _load_ext(name):
    path = os.path.join(_REF, name + _EXT)
    if not os.path.exists(path):
        raise ImportError("oracle/_ref/%s%s missing: run `make -C oracle ref` where /root/reference exists" % (name, _EXT))
    loader = importlib.machinery.ExtensionFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


_cache = {}


def set_threads(n):
    """OpenMP threads used by the compiled reference (same libgomp instance as the oracle .so files).

    The reference's 3D NMS accumulates the anisotropy inside an `omp parallel for` without
    synchronisation (stardist3d_impl.cpp:995-1011), so its result is only DEFINED for one thread;
    parity tests therefore run it with n=1.  Timing legs (bench.py cpu_baseline) set n=cores."""
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def stardist2d():
    if "2d" not in _cache:
        _cache["2d"] = _load_ext("stardist2d")
    return _cache["2d"]


def stardist3d():
    if "3d" not in _cache:
        _cache["3d"] = _load_ext("stardist3d")
        set_threads(int(os.environ.get("ORACLE_OMP_THREADS", "1")))
    return _cache["3d"]


def _clipper():
    if "clip" not in _cache:
        lib = ctypes.CDLL(os.path.join(_REF, "libclipper_ref.so"))
        i64p = ctypes.POINTER(ctypes.c_int64)
        lib.clipper_ref_area.restype = ctypes.c_float
        lib.clipper_ref_area.argtypes = [i64p, i64p, ctypes.c_int, i64p, i64p, ctypes.c_int]
        lib.clipper_ref_intersect.restype = ctypes.c_int
        lib.clipper_ref_intersect.argtypes = [i64p, i64p, ctypes.c_int, i64p, i64p, ctypes.c_int,
                                              i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        _cache["clip"] = lib
    return _cache["clip"]


def _p64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def clipper_area(xa, ya, xb, yb):
    """float32 intersection area exactly as stardist2d.cpp:152-165 (A = clip, B = subject)."""
    xa, ya, xb, yb = (np.ascontiguousarray(v, np.int64) for v in (xa, ya, xb, yb))
    return float(_clipper().clipper_ref_area(_p64(xa), _p64(ya), len(xa), _p64(xb), _p64(yb), len(xb)))


def clipper_paths(xa, ya, xb, yb):
    """Raw output paths of the vendored Clipper for the same call."""
    xa, ya, xb, yb = (np.ascontiguousarray(v, np.int64) for v in (xa, ya, xb, yb))
    out = np.zeros(2 * 1024, np.int64)
    lens = np.zeros(64, np.int32)
    n = _clipper().clipper_ref_intersect(_p64(xa), _p64(ya), len(xa), _p64(xb), _p64(yb), len(xb),
                                         _p64(out), 1024, lens.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 64)
    assert n >= 0
    paths, k = [], 0
    for r in range(n):
        paths.append(out[2 * k:2 * (k + lens[r])].reshape(-1, 2).copy())
        k += lens[r]
    return paths


def pair_volumes(dist, points, verts, faces, pairs, kernel=True, hull=True):
    """reference qhull_overlap_kernel / qhull_overlap_convex_hulls (stardist3d_impl.cpp:830-939) per pair, float32 as they return"""
    if "qh" not in _cache:
        lib = ctypes.CDLL(os.path.join(_REF, "libqhull_ref.so"))
        lib.ref_pair_volumes.restype = None
        _cache["qh"] = lib
    dist = np.ascontiguousarray(dist, np.float32); points = np.ascontiguousarray(points, np.float32)
    verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    pairs = np.ascontiguousarray(pairs, np.int32)
    vk = np.zeros(len(pairs), np.float32) if kernel else None
    vh = np.zeros(len(pairs), np.float32) if hull else None
    vp = ctypes.c_void_p
    _cache["qh"].ref_pair_volumes(vp(dist.ctypes.data), vp(points.ctypes.data), vp(verts.ctypes.data), vp(faces.ctypes.data), vp(pairs.ctypes.data),
                                  ctypes.c_int(len(pairs)), ctypes.c_int(dist.shape[1]), ctypes.c_int(len(faces)),
                                  vp(vk.ctypes.data) if kernel else None, vp(vh.ctypes.data) if hull else None)
    return vk, vh


def pair_cascade(dist, points, verts, faces, pairs, anisotropy=(1.0, 1.0, 1.0)):
    """every quantity the reference's cascade (stardist3d_impl.cpp:1207-1318) looks at for the given pairs (i, j), by the reference's own
    functions: rows of (volume i, volume j, upper bound, lower bound, kernel stage, hull stage, rendered overlap -- the full count --, 0)"""
    if "qh" not in _cache:
        lib = ctypes.CDLL(os.path.join(_REF, "libqhull_ref.so"))
        lib.ref_pair_volumes.restype = None
        _cache["qh"] = lib
    dist = np.ascontiguousarray(dist, np.float32); points = np.ascontiguousarray(points, np.float32)
    verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.int32)
    pairs = np.ascontiguousarray(pairs, np.int32); an = np.ascontiguousarray(anisotropy, np.float32)
    out = np.zeros((len(pairs), 8), np.float32)
    vp = ctypes.c_void_p
    _cache["qh"].ref_pair_cascade.restype = None
    _cache["qh"].ref_pair_cascade(vp(dist.ctypes.data), vp(points.ctypes.data), vp(verts.ctypes.data), vp(faces.ctypes.data), vp(pairs.ctypes.data),
                                  ctypes.c_int(len(pairs)), ctypes.c_int(dist.shape[1]), ctypes.c_int(len(faces)), vp(an.ctypes.data), vp(out.ctypes.data))
    return out
