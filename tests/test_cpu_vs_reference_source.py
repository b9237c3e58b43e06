"""CPU: the pure-host functions of the mirror against the REFERENCE'S OWN functions, taken from the reference files at run time
(nothing is copied into this repo; the goldens under tests/golden pin a handful of cases each for the GPU box, this file sweeps many
parameter values where the reference sources are at hand -- the build container; skipped elsewhere).

  rays3d.py        loaded as a module (numpy + scipy only): every Rays_* class over parameter sweeps, to_json / rays_from_json,
                   dist_loss_weights, volume, copy(scale)
  big.py           loaded as a module with stub packages for what it imports but the block algebra does not use (skimage, csbdeep,
                   .geometry): Block.cover / BlockND.cover over random sizes -- start, end, read / write / crop slices, context,
                   responsibility rule incl. its exceptions; _grid_divisible
  matching.py      relabel_sequential                      (function bodies via ast -> exec)
  nms.py           _ind_prob_thresh
  geometry/geom2d  ray_angles, dist_to_coord (numpy path), _dist_to_coord_old
  geometry/geom3d  dist_to_coord3D, export_to_obj_file3D
  utils.py         _normalize_grid, _is_power_of_2, polyroi_bytearray, export_imagej_rois (bytes of the zip members)"""
import ast
import importlib.util
import io
import os
import sys
import types
import zipfile

import numpy as np
import pytest

REF = "/root/reference/stardist"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference sources (build container only)")


def _raise(e):
    raise e


def ref_functions(relpath, names, ns):
    path = os.path.join(REF, relpath)
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    assert set(names) <= set(ns), sorted(set(names) - set(ns))
    return ns


@pytest.fixture(scope="module")
def ref_rays():
    spec = importlib.util.spec_from_file_location("_ref_rays3d", os.path.join(REF, "rays3d.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _same_rays(a, b, tag):
    assert len(a) == len(b), tag
    assert a.vertices.dtype == b.vertices.dtype and np.array_equal(a.vertices, b.vertices), tag
    assert np.array_equal(a.faces, b.faces), tag
    assert repr(a) == repr(b) and a.to_json() == b.to_json(), tag


def test_ray_sets_equal_the_reference_classes(ref_rays):
    from stardist_amd import rays3d as M
    for n in (8, 13, 32, 64, 65, 96, 128, 187):
        for an in (None, (2, 1, 1), (1.0, 0.5, 1.5)):
            a, b = M.Rays_GoldenSpiral(n, anisotropy=an), ref_rays.Rays_GoldenSpiral(n, anisotropy=an)
            _same_rays(a, b, ("golden", n, an))
            d = np.random.RandomState(n).uniform(1, 9, (5, n))
            assert np.allclose(a.volume(d), b.volume(d), rtol=1e-12, atol=0)      # float64 helpers, other summation order
            assert np.allclose(a.surface(d), b.surface(d), rtol=1e-12, atol=0) and a.surface(d).shape == b.surface(d).shape      # (volume(None): the reference's default raises for n != 3)
            assert np.array_equal(a.dist_loss_weights((1, 2, 0.5)), b.dist_loss_weights((1, 2, 0.5)))
            _same_rays(a.copy(scale=(0.5, 2, 1.25)), b.copy(scale=(0.5, 2, 1.25)), ("golden copy", n, an))
            _same_rays(M.rays_from_json(a.to_json()), ref_rays.rays_from_json(b.to_json()), ("golden json", n, an))
    for nx, nz in ((11, 5), (8, 5), (4, 3), (16, 9), (5, 4)):
        _same_rays(M.Rays_Cartesian(nx, nz), ref_rays.Rays_Cartesian(nx, nz), ("cartesian", nx, nz))
    for lvl in (1, 2, 3, 4):
        _same_rays(M.Rays_Tetra(lvl), ref_rays.Rays_Tetra(lvl), ("tetra", lvl))
        _same_rays(M.Rays_Octo(lvl), ref_rays.Rays_Octo(lvl), ("octo", lvl))
    g = ref_rays.Rays_GoldenSpiral(10)
    a, b = M.Rays_Explicit(g.vertices.tolist(), g.faces.tolist()), ref_rays.Rays_Explicit(g.vertices.tolist(), g.faces.tolist())
    _same_rays(a, b, "explicit")
    for bad in (dict(n=3),):                                              # refusals
        with pytest.raises(Exception) as e1:
            ref_rays.Rays_GoldenSpiral(**bad)
        with pytest.raises(type(e1.value)):
            M.Rays_GoldenSpiral(**bad)


@pytest.fixture(scope="module")
def ref_big():
    """the reference's big.py as a module; the packages it imports for rendering / measuring (not used by the block algebra) are stubs"""
    saved = {k: sys.modules.get(k) for k in ("skimage", "skimage.measure", "skimage.draw", "csbdeep", "csbdeep.utils", "_ref_sd", "_ref_sd.geometry")}

    def axes_check_and_normalize(axes, length=None, disallowed=None, return_allowed=False):        # csbdeep.utils, restated for the test
        allowed = "STCZYX"
        axes = str(axes).upper()
        assert all(a in allowed for a in axes) and len(set(axes)) == len(axes) and (length is None or len(axes) == length)
        return (axes, allowed) if return_allowed else axes

    def axes_dict(axes):
        axes, allowed = axes_check_and_normalize(axes, return_allowed=True)
        return {a: None if axes.find(a) == -1 else axes.find(a) for a in allowed}
    mods = {"skimage": types.ModuleType("skimage"), "skimage.measure": types.ModuleType("skimage.measure"), "skimage.draw": types.ModuleType("skimage.draw"),
            "csbdeep": types.ModuleType("csbdeep"), "csbdeep.utils": types.ModuleType("csbdeep.utils"),
            "_ref_sd": types.ModuleType("_ref_sd"), "_ref_sd.geometry": types.ModuleType("_ref_sd.geometry")}
    mods["skimage.measure"].regionprops = None; mods["skimage.draw"].polygon = None
    mods["csbdeep.utils"]._raise, mods["csbdeep.utils"].axes_check_and_normalize, mods["csbdeep.utils"].axes_dict = _raise, axes_check_and_normalize, axes_dict
    mods["_ref_sd"].__path__ = []
    mods["_ref_sd.geometry"].polygons_to_label_coord = mods["_ref_sd.geometry"].polyhedron_to_label = None
    sys.modules.update(mods)
    try:
        m = types.ModuleType("_ref_sd.big")
        m.__package__ = "_ref_sd"
        path = os.path.join(REF, "big.py")
        exec(compile(open(path).read(), path, "exec"), m.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m


def _block_row(t):
    return (t.start, t.end, t.slice_read.start, t.slice_read.stop, t.slice_write.start, t.slice_write.stop, t.slice_crop_context.start,
            t.slice_crop_context.stop, t.context_start, t.context_end, t.at_begin, t.at_end)


def test_block_covers_equal_the_reference_classes(ref_big):
    from stardist_amd import big as B
    rng = np.random.RandomState(0)
    n_ok = n_err = 0
    for it in range(600):
        grid = int(rng.choice([1, 1, 2, 4, 8]))
        size = int(rng.randint(8, 400))
        bs = int(rng.randint(4, 200)) // grid * grid
        mo = int(rng.randint(0, 40)) // grid * grid
        ctx = int(rng.randint(0, 24)) // grid * grid
        try:
            want = ref_big.Block.cover(size, bs, mo, ctx, grid, verbose=False)
        except Exception as e:                                                        # noqa: BLE001 -- the same refusal is expected
            with pytest.raises(type(e)):
                B.Block.cover(size, bs, mo, ctx, grid, verbose=False)
            n_err += 1
            continue
        got = B.Block.cover(size, bs, mo, ctx, grid, verbose=False)
        assert [_block_row(t) for t in got] == [_block_row(t) for t in want], (size, bs, mo, ctx, grid)
        # the responsibility rule on random intervals, exceptions included (big.py:89-122)
        for _ in range(12):
            a = int(rng.randint(0, size)); b = int(rng.randint(a + 1, min(size, a + mo + 6) + 1))
            for tg, tw in zip(got, want):
                try:
                    r = tw.is_responsible((a, b))
                except Exception as e:                                                # noqa: BLE001
                    with pytest.raises(Exception) as e2:
                        tg.is_responsible((a, b))
                    assert type(e2.value).__name__ == type(e).__name__, (size, bs, mo, ctx, grid, a, b)
                else:
                    assert tg.is_responsible((a, b)) == r, (size, bs, mo, ctx, grid, a, b)
        n_ok += 1
    assert n_ok > 150 and n_err > 20, (n_ok, n_err)
    for g, v in ((1, 7), (4, 8), (4, 10), (8, 3), (2, 0)):
        assert B._grid_divisible(g, v, verbose=False) == ref_big._grid_divisible(g, v, verbose=False)


def test_blocknd_covers_equal_the_reference_classes(ref_big):
    from stardist_amd import big as B
    rng = np.random.RandomState(1)
    for it in range(60):
        nd = int(rng.choice([2, 3]))
        axes = "YX" if nd == 2 else "ZYX"
        with_c = bool(rng.randint(0, 2))
        shape = tuple(int(v) for v in rng.randint(40, 160, nd))
        grid = tuple(int(v) for v in rng.choice([1, 2, 4], nd))
        bs = tuple(int(rng.randint(24, 80)) // g * g for g in grid)
        mo = tuple(int(rng.randint(0, 12)) // g * g for g in grid)
        ctx = tuple(int(rng.randint(0, 8)) // g * g for g in grid)
        if with_c:
            axes, shape, grid, bs, mo, ctx = axes + "C", shape + (3,), grid + (1,), bs + (3,), mo + (0,), ctx + (0,)
        try:
            want = ref_big.BlockND.cover(shape, axes, bs, mo, ctx, grid)
        except Exception as e:                                                        # noqa: BLE001
            with pytest.raises(type(e)):
                B.BlockND.cover(shape, axes, bs, mo, ctx, grid)
            continue
        got = B.BlockND.cover(shape, axes, bs, mo, ctx, grid)
        assert len(got) == len(want)
        x = rng.randint(0, 9, shape)
        for bg, bw in zip(got, want):
            assert bg.id == bw.id and bg.slice_read(axes) == bw.slice_read(axes) and bg.slice_write(axes) == bw.slice_write(axes)
            assert bg.slice_crop_context(axes) == bw.slice_crop_context(axes)
            assert np.array_equal(bg.read(x, axes=axes), bw.read(x, axes=axes))
            sub = bw.read(x, axes=axes)
            assert np.array_equal(bg.crop_context(sub, axes=axes), bw.crop_context(sub, axes=axes))
            pts = rng.randint(0, 30, (5, nd)).astype(float)
            ax_sp = axes.replace("C", "")
            assert np.array_equal(bg.translate_coordinates(pts.copy(), axes=ax_sp), bw.translate_coordinates(pts.copy(), axes=ax_sp))


def test_relabel_sequential_equals_the_reference_function():
    from stardist_amd.matching import relabel_sequential
    ns = ref_functions("matching.py", {"relabel_sequential"}, {"np": np})
    rng = np.random.RandomState(2)
    for it in range(200):
        dt = rng.choice([np.uint8, np.uint16, np.int32, np.int64, np.float32])
        shape = tuple(rng.randint(1, 12, rng.randint(1, 4)))
        a = (rng.randint(0, rng.choice([3, 40, 250]), shape) * rng.choice([1, 1, 7])).astype(dt)
        off = int(rng.choice([1, 1, 5, 300, 70000]))
        want = ns["relabel_sequential"](a.copy(), off)
        got = relabel_sequential(a.copy(), off)
        for g, w in zip(got, want):
            assert g.dtype == w.dtype and np.array_equal(g, w), (dt, shape, off)
    for bad, off in ((np.array([-1, 2]), 1), (np.array([1, 2]), 0)):
        with pytest.raises(ValueError):
            ns["relabel_sequential"](bad, off)
        with pytest.raises(ValueError):
            relabel_sequential(bad, off)


def test_ind_prob_thresh_and_grid_helpers_equal_the_reference_functions():
    from stardist_amd import nms as NM, utils as U
    ns = ref_functions("utils.py", {"_normalize_grid", "_is_power_of_2"}, {"np": np, "_raise": _raise})
    ref_functions("nms.py", {"_ind_prob_thresh"}, ns)
    rng = np.random.RandomState(3)
    for it in range(100):
        nd = int(rng.choice([2, 3]))
        prob = rng.uniform(0, 1, tuple(rng.randint(1, 14, nd))).astype(np.float32)
        thr = float(rng.uniform(0, 1))
        for b in (None, 0, 1, 2, 3, tuple((int(rng.randint(0, 3)), int(rng.randint(0, 3))) for _ in range(nd))):
            assert np.array_equal(NM._ind_prob_thresh(prob, thr, b=b), ns["_ind_prob_thresh"](prob, thr, b=b)), (prob.shape, b)
    for grid, n in (((1, 1), 2), ((2, 4), 2), ([1, 2, 8], 3), ((1, 3), 2), ((1, 1), 3), (2, 2), ((0, 1), 2), ((1.0, 2.0), 2), ((-2, 1), 2)):
        try:
            want = ns["_normalize_grid"](grid, n)
        except ValueError:
            with pytest.raises(ValueError):
                U._normalize_grid(grid, n)
        else:
            assert U._normalize_grid(grid, n) == want
    for i in (1, 2, 3, 4, 6, 8, 1024, 1000):
        assert U._is_power_of_2(i) == ns["_is_power_of_2"](i)


def test_geometry_host_functions_equal_the_reference_functions(tmp_path):
    from stardist_amd.geometry import geom2d, geom3d
    from stardist_amd.rays3d import Rays_GoldenSpiral
    from stardist_amd.utils import _normalize_grid
    ns = ref_functions("geometry/geom2d.py", {"ray_angles", "dist_to_coord", "_dist_to_coord_old"}, {"np": np, "_normalize_grid": _normalize_grid})
    n3 = ref_functions("geometry/geom3d.py", {"dist_to_coord3D", "export_to_obj_file3D"}, {"np": np, "_raise": _raise, "tqdm": lambda x, **k: x})
    rng = np.random.RandomState(4)
    for R in (3, 8, 32, 33, 64):
        assert np.array_equal(geom2d.ray_angles(R), ns["ray_angles"](R))
        d = rng.uniform(0.5, 30, (50, R)).astype(np.float32)
        for pts in (rng.randint(0, 500, (50, 2)), rng.uniform(0, 500, (50, 2)).astype(np.float32), rng.uniform(0, 500, (50, 2))):
            for sc in ((1, 1), (0.5, 2.0), (1 / 1.47, 1 / 0.34)):
                g, w = geom2d.dist_to_coord(d, pts, scale_dist=sc), ns["dist_to_coord"](d, pts, scale_dist=sc)
                assert g.dtype == w.dtype and np.array_equal(g, w), (R, pts.dtype, sc)
        for rhos, grid in ((rng.uniform(0, 9, (5, 7, R)).astype(np.float32), (1, 1)), (rng.uniform(0, 9, (2, 4, 6, R)), (2, 4))):
            g, w = geom2d._dist_to_coord_old(rhos, grid), ns["_dist_to_coord_old"](rhos, grid)
            assert g.dtype == w.dtype and g.shape == w.shape and np.array_equal(g, w)
    rays = Rays_GoldenSpiral(24, anisotropy=(2, 1, 1))
    d3 = rng.uniform(2, 7, (6, 24)).astype(np.float32); p3 = rng.uniform(0, 50, (6, 3)).astype(np.float32)
    assert np.array_equal(geom3d.dist_to_coord3D(d3, p3, rays.vertices), n3["dist_to_coord3D"](d3, p3, rays.vertices))
    polys = dict(dist=d3, points=p3, rays_vertices=rays.vertices, rays_faces=rays.faces)
    for kw in (dict(), dict(single_mesh=False, uv_map=True, name="cell"), dict(scale=(0.05, 0.2, 0.2)), dict(scale=2)):
        assert geom3d.export_to_obj_file3D(polys, fname=None, **kw) == n3["export_to_obj_file3D"](polys, fname=None, **kw), kw


def test_imagej_roi_bytes_equal_the_reference_functions(tmp_path):
    import datetime                                                                  # noqa: F401 -- names the reference functions look up
    from pathlib import Path
    from zipfile import ZIP_DEFLATED, ZipFile
    from stardist_amd import utils as U
    ns = ref_functions("utils.py", {"polyroi_bytearray", "export_imagej_rois"}, {"np": np, "Path": Path, "ZipFile": ZipFile, "ZIP_DEFLATED": ZIP_DEFLATED, "_raise": _raise})
    rng = np.random.RandomState(5)
    for it in range(60):
        n = int(rng.randint(3, 40))
        x, y = rng.uniform(0, 300, n), rng.uniform(0, 300, n)
        if it % 3 == 0:
            x, y = np.round(x), np.round(y)
        for pos in (None, 1, 17):
            for sub in (True, False):
                assert bytes(U.polyroi_bytearray(x, y, pos=pos, subpixel=sub)) == bytes(ns["polyroi_bytearray"](x, y, pos=pos, subpixel=sub)), (it, pos, sub)
    groups = [rng.uniform(0, 100, (4, 2, 16)), rng.uniform(0, 100, (2, 2, 16))]
    for arg, kw in ((groups, dict()), (groups[0], dict(set_position=False, subpixel=False))):
        a, b = str(tmp_path / "mine"), str(tmp_path / "ref.zip")
        U.export_imagej_rois(a, arg, **kw); ns["export_imagej_rois"](b, arg, **kw)
        za, zb = zipfile.ZipFile(a + ".zip"), zipfile.ZipFile(b)
        assert za.namelist() == zb.namelist()
        for name in za.namelist():
            assert za.read(name) == zb.read(name), name


def test_pad_and_crop_resizer_equals_the_reference_class():
    """StarDistPadAndCropResizer (models/base.py:1162-1211): reflect padding at the end of every axis to the network's divisor (incl. pads
    longer than the axis and length-1 axes), the crop of grid-subsampled outputs, the filter of points that fall into the padding"""
    import math
    import torch
    from stardist_amd.models.base import StarDistPadAndCropResizer as Mine

    def axes_check_and_normalize(axes, length=None, **k):
        axes = str(axes).upper()
        assert length is None or len(axes) == length
        return axes
    ns = {"np": np, "math": math, "Resizer": object, "axes_check_and_normalize": axes_check_and_normalize}
    Ref = ref_functions("models/base.py", {"StarDistPadAndCropResizer"}, ns)["StarDistPadAndCropResizer"]
    rng = np.random.RandomState(6)
    for it in range(200):
        nd = int(rng.choice([2, 3]))
        sp = "YX" if nd == 2 else "ZYX"
        with_c = bool(rng.randint(0, 2))
        axes = sp + ("C" if with_c else "")
        grid = {a: int(rng.choice([1, 2, 4])) for a in sp}
        div = tuple(int(grid.get(a, 1) * rng.choice([1, 2, 4, 8])) if a != "C" else 1 for a in axes)
        shape = tuple(int(rng.choice([1, 2, 3, 5, 9, 17, 30, 33])) if a != "C" else int(rng.randint(1, 4)) for a in axes)
        x = rng.uniform(0, 1, shape).astype(np.float32)
        a, b = Mine(grid), Ref(grid)
        xa, xb = a.before(torch.from_numpy(x), axes, div), b.before(x, axes, div)
        assert np.array_equal(xa.numpy(), xb), (shape, axes, div)
        assert a.pad == b.pad and a.padded_shape == b.padded_shape
        # network output on the grid: spatial axes subsampled, channels replaced
        out_axes = sp + "C"
        out_shape = tuple(xb.shape[axes.index(c)] // grid[c] for c in sp) + (5,)
        y = rng.uniform(0, 1, out_shape).astype(np.float32)
        assert np.array_equal(a.after(torch.from_numpy(y), out_axes).numpy(), b.after(y, out_axes)), (shape, axes, div, grid)
        pts = np.stack([rng.randint(0, max(1, xb.shape[axes.index(c)]), 40) for c in sp], 1)
        assert np.array_equal(np.asarray(a.filter_points(nd, pts, sp)), b.filter_points(nd, pts, sp)[0])
        assert np.array_equal(a.filter_points(nd, torch.from_numpy(pts), sp).numpy(), b.filter_points(nd, pts, sp)[0])


def _ref_configs(ref_rays):
    """the reference's Config2D / Config3D classes (model2d.py:123-269, model3d.py:129-311) on a stand-in for csbdeep's BaseConfig that sets
    what every config.json of the reference carries (models/examples/*/config.json): n_dim, axes (+ 'C'), channel counts, checkpoint names"""
    import warnings
    from packaging.version import Version

    class BaseConfig(object):
        def __init__(self, axes="YX", n_channel_in=1, n_channel_out=1, **kw):
            axes = str(axes).upper()
            axes = axes if "C" in axes else axes + "C"
            self.n_dim, self.axes = len(axes) - 1, axes
            self.n_channel_in, self.n_channel_out = int(max(1, n_channel_in)), int(max(1, n_channel_out))
            self.train_checkpoint, self.train_checkpoint_last, self.train_checkpoint_epoch = "weights_best.h5", "weights_last.h5", "weights_now.h5"

        def update_parameters(self, allow_new=False, **kwargs):
            if not allow_new:
                bad = [k for k in kwargs if not hasattr(self, k)]
                if bad:
                    raise AttributeError("Not allowed to add new parameters (%s)" % ", ".join(bad))
            for k, v in kwargs.items():
                setattr(self, k, v)
    u = ref_functions("utils.py", {"_normalize_grid", "_is_power_of_2"}, {"np": np, "_raise": _raise})
    ns = {"np": np, "warnings": warnings, "BaseConfig": BaseConfig, "_normalize_grid": u["_normalize_grid"], "backend_channels_last": lambda: True,
          "Version": Version, "keras": object(), "_raise": _raise, "Rays_GoldenSpiral": ref_rays.Rays_GoldenSpiral, "rays_from_json": ref_rays.rays_from_json}
    ref_functions("models/model2d.py", {"Config2D"}, ns)
    ref_functions("models/model3d.py", {"Config3D"}, ns)
    return ns["Config2D"], ns["Config3D"]


def _norm(v):
    if isinstance(v, dict):
        return {k: _norm(x) for k, x in v.items()}
    if isinstance(v, (tuple, list)):
        return [_norm(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def test_config_classes_equal_the_reference_classes(ref_rays, capsys):
    import json
    from stardist_amd.models import Config2D, Config3D
    R2, R3 = _ref_configs(ref_rays)
    cases2 = [dict(), dict(n_rays=64, grid=(2, 2), n_channel_in=3), dict(n_classes=2), dict(axes="YXC", n_rays=16, unet_n_depth=4, net_conv_after_unet=64, train_epochs=3),
              dict(grid=(1, 4), unet_batch_norm=True, train_patch_size=(128, 128), use_gpu=True, train_reduce_lr=None), dict(n_classes=3, train_class_weights=(1, 2, 3, 4))]
    for kw in cases2:
        a, b = Config2D(**kw), R2(**kw)
        assert list(vars(a)) == list(vars(b)), kw                          # same keys in the same order (the order config.json is written in)
        assert _norm(vars(a)) == _norm(vars(b)), kw
    cases3 = [dict(), dict(rays=32), dict(anisotropy=(2, 1, 1)), dict(rays=ref_rays.Rays_GoldenSpiral(48, anisotropy=(2, 1, 1)), anisotropy=(2, 1, 1), grid=(1, 2, 2)),
              dict(backbone="resnet", grid=(1, 2, 2), n_channel_in=2), dict(n_classes=2, rays=16), dict(n_rays=24), dict(unet_n_depth=3, train_batch_size=2)]
    for kw in cases3:
        kwm = dict(kw)
        if "rays" in kw and not np.isscalar(kw["rays"]):                   # each side gets its own ray class
            from stardist_amd.rays3d import Rays_GoldenSpiral
            kwm["rays"] = Rays_GoldenSpiral(48, anisotropy=(2, 1, 1))
        a, b = Config3D(**kwm), R3(**kw)
        assert list(vars(a)) == list(vars(b)), kw
        assert _norm(vars(a)) == _norm(vars(b)), kw
    # a configuration dictionary read back (what csbdeep's loader does with config.json): Config(**dict)
    for path, cls, rcls in (("models/examples/2D_demo/config.json", Config2D, R2), ("models/examples/3D_demo/config.json", Config3D, R3)):
        d = json.load(open(os.path.join(os.path.dirname(REF), path)))
        a, b = cls(**d), rcls(**d)
        assert _norm(vars(a)) == _norm(vars(b)) and list(vars(a)) == list(vars(b)), path
        assert json.loads(a.to_json()) == _norm(vars(b))
        c = cls.from_json(os.path.join(os.path.dirname(REF), path))        # the mirror's loader gives the same object
        assert _norm(vars(c)) == _norm(vars(a)), path
    # refusals
    for cls, rcls in ((Config2D, R2), (Config3D, R3)):
        for kw, exc in ((dict(no_such_key=1), AttributeError), (dict(backbone="vgg"), ValueError), (dict(grid=(3, 1) if cls is Config2D else (3, 1, 1)), ValueError),
                        (dict(train_loss_weights=(1, 2, 3)), ValueError), (dict(n_classes=2, train_class_weights=(1, 1)), ValueError)):
            with pytest.raises(exc):
                rcls(**kw)
            with pytest.raises(exc):
                cls(**kw)


@pytest.fixture()
def ref_nms():
    """the reference's nms.py as a module of a stand-in package whose `lib.stardist2d` / `lib.stardist3d` ARE the compiled reference natives
    (oracle/_ref) and whose `utils` holds the reference's _normalize_grid: the Python glue of the reference, end to end"""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    ref.set_threads(1)
    names = ("_ref_sd", "_ref_sd.lib", "_ref_sd.lib.stardist2d", "_ref_sd.lib.stardist3d", "_ref_sd.utils", "_ref_sd.nms")
    saved = {k: sys.modules.get(k) for k in names}
    pkg, lib, utils = types.ModuleType("_ref_sd"), types.ModuleType("_ref_sd.lib"), types.ModuleType("_ref_sd.utils")
    pkg.__path__, lib.__path__ = [], []
    u = ref_functions("utils.py", {"_normalize_grid", "_is_power_of_2"}, {"np": np, "_raise": _raise})
    utils._normalize_grid = u["_normalize_grid"]
    sys.modules.update({"_ref_sd": pkg, "_ref_sd.lib": lib, "_ref_sd.lib.stardist2d": ref.stardist2d(), "_ref_sd.lib.stardist3d": ref.stardist3d(), "_ref_sd.utils": utils})
    m = types.ModuleType("_ref_sd.nms")
    m.__package__ = "_ref_sd"
    path = os.path.join(REF, "nms.py")
    exec(compile(open(path).read(), path, "exec"), m.__dict__)
    sys.modules["_ref_sd.nms"] = m
    yield m
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_nms_python_glue_equals_the_reference_module(ref_nms, monkeypatch, capsys):
    """stardist/nms.py end to end (thresholding with a border, score sort, grid scaling, the native call, what is returned and in which
    order) with the SAME natives under both: the mirror's wrappers are given the compiled reference's functions for this test"""
    from oracle import ref
    from stardist_amd import nms as NM
    from stardist_amd.lib import stardist2d as sd2, stardist3d as sd3
    from stardist_amd.rays3d import Rays_GoldenSpiral
    m2, m3 = ref.stardist2d(), ref.stardist3d()
    monkeypatch.setattr(sd2, "c_non_max_suppression_inds", lambda d, p, a, b, c, t, **k: m2.c_non_max_suppression_inds(
        np.ascontiguousarray(d, np.float32), np.ascontiguousarray(p, np.float32), int(a), int(b), int(c), np.float32(t)).astype(bool))
    monkeypatch.setattr(sd3, "c_non_max_suppression_inds", lambda d, p, V, F, s, a, b, c, t, **k: m3.c_non_max_suppression_inds(
        np.ascontiguousarray(d, np.float32), np.ascontiguousarray(p, np.float32), np.ascontiguousarray(V, np.float32), np.ascontiguousarray(F, np.int32),
        np.ascontiguousarray(s, np.float32), int(a), int(b), int(c), np.float32(t)).astype(bool))
    rng = np.random.RandomState(7)

    def same(a, b, tag):
        assert len(a) == len(b), tag
        for x, y in zip(a, b):
            x, y = np.asarray(x), np.asarray(y)
            assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y), (tag, x.dtype, y.dtype, x.shape, y.shape)
    for it in range(12):
        R = int(rng.choice([8, 32]))
        grid = tuple(int(v) for v in rng.choice([1, 2, 4], 2))
        H, W = int(rng.randint(20, 60)), int(rng.randint(20, 60))
        dist = (6 * (1 + 0.3 * rng.uniform(-1, 1, (H, W, R)))).astype(np.float32)
        prob = rng.uniform(0, 1, (H, W)).astype(np.float32)
        kw = dict(grid=grid, b=int(rng.choice([0, 2, 3])), nms_thresh=float(rng.choice([0.3, 0.5])), prob_thresh=float(rng.choice([0.6, 0.8])))
        same(NM.non_maximum_suppression(dist, prob, **kw), ref_nms.non_maximum_suppression(dist, prob, **kw), ("dense2d", kw))
        mask = prob > 0.7
        pts = np.stack(np.where(mask), 1) * np.array(grid).reshape(1, 2)
        kws = dict(b=kw["b"], nms_thresh=kw["nms_thresh"], use_bbox=bool(it % 2), use_kdtree=bool(it % 3))
        same(NM.non_maximum_suppression_sparse(dist[mask], prob[mask], pts, **kws), ref_nms.non_maximum_suppression_sparse(dist[mask], prob[mask], pts, **kws), ("sparse2d", kws))
        sc = prob[mask]
        same([NM.non_maximum_suppression_inds(dist[mask], pts.astype(np.int32), sc, thresh=0.4, verbose=0)],
             [ref_nms.non_maximum_suppression_inds(dist[mask], pts.astype(np.int32), sc, thresh=0.4, verbose=0)], "inds2d")
    for it in range(6):
        rays = Rays_GoldenSpiral(int(rng.choice([16, 32])), anisotropy=(None if it % 2 else (2, 1, 1)))
        grid = tuple(int(v) for v in rng.choice([1, 2], 3))
        shape = tuple(int(v) for v in rng.randint(10, 18, 3))
        dist = (4 * (1 + 0.2 * rng.uniform(-1, 1, shape + (len(rays),)))).astype(np.float32)
        prob = rng.uniform(0, 1, shape).astype(np.float32)
        kw = dict(grid=grid, b=int(rng.choice([0, 2])), nms_thresh=0.3, prob_thresh=0.8)
        same(NM.non_maximum_suppression_3d(dist, prob, rays, **kw), ref_nms.non_maximum_suppression_3d(dist, prob, rays, **kw), ("dense3d", kw))
        mask = prob > 0.85
        pts = np.stack(np.where(mask), 1) * np.array(grid).reshape(1, 3)
        same(NM.non_maximum_suppression_3d_sparse(dist[mask], prob[mask], pts, rays, b=kw["b"], nms_thresh=0.3),
             ref_nms.non_maximum_suppression_3d_sparse(dist[mask], prob[mask], pts, rays, b=kw["b"], nms_thresh=0.3), "sparse3d")
        same([NM.non_maximum_suppression_3d_inds(dist[mask], pts.astype(np.int32), rays, prob[mask], thresh=0.3, verbose=0)],
             [ref_nms.non_maximum_suppression_3d_inds(dist[mask], pts.astype(np.int32), rays, prob[mask], thresh=0.3, verbose=0)], "inds3d")


def _ref_method(relpath, cls, name, ns):
    path = os.path.join(REF, relpath)
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == name:
                    exec(compile(ast.Module([f], []), path, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


def test_instances_from_prediction_equals_the_reference_methods(ref_nms, ref_rays, monkeypatch):
    """StarDist2D / StarDist3D._instances_from_prediction (model2d.py:512-563, model3d.py:589-674) -- dense and sparse input, multi-class
    probabilities, `scale`, the 3D overlap label -- label image and result dict against the reference's own methods.  Both sides run on the same
    natives: the compiled reference's NMS and polyhedron rasteriser, and for 2D the restatement of skimage.draw.polygon (== the real one on
    30 000 polygons, profiles/r05_raster2d_oracle_vs_skimage.txt)."""
    from collections import namedtuple
    from oracle import port, ref
    from stardist_amd.lib import stardist2d as sd2, stardist3d as sd3
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    from stardist_amd.rays3d import Rays_GoldenSpiral
    from stardist_amd.utils import _normalize_grid
    m2, m3 = ref.stardist2d(), ref.stardist3d()
    f32, i32 = (lambda a: np.ascontiguousarray(a, np.float32)), (lambda a: np.ascontiguousarray(a, np.int32))
    monkeypatch.setattr(sd2, "c_non_max_suppression_inds", lambda d, p, a, b, c, t, **k: m2.c_non_max_suppression_inds(f32(d), f32(p), int(a), int(b), int(c), np.float32(t)).astype(bool))
    monkeypatch.setattr(sd2, "c_polygons_to_label", lambda coord, labels, shape, window=None: port.polygons_to_label_coord(coord, shape, labels=labels))
    monkeypatch.setattr(sd3, "c_non_max_suppression_inds", lambda d, p, V, F, s, a, b, c, t, **k: m3.c_non_max_suppression_inds(f32(d), f32(p), f32(V), i32(F), f32(s), int(a), int(b), int(c), np.float32(t)).astype(bool))
    monkeypatch.setattr(sd3, "c_polyhedron_to_label", lambda d, p, V, F, l, mode, vb, uo, ol, shape, window=None: m3.c_polyhedron_to_label(f32(d), f32(p), f32(V), i32(F), i32(l), int(mode), int(vb), int(uo), int(ol), tuple(shape)))
    Thr = namedtuple("Thresholds", ("prob", "nms"))
    g2 = ref_functions("geometry/geom2d.py", {"ray_angles", "dist_to_coord", "polygons_to_label_coord", "polygons_to_label"}, {"np": np, "polygon": port.polygon, "_check_label_array": lambda *a, **k: True})
    r2 = _ref_method("models/model2d.py", "StarDist2D", "_instances_from_prediction",
                     {"np": np, "non_maximum_suppression": ref_nms.non_maximum_suppression, "non_maximum_suppression_sparse": ref_nms.non_maximum_suppression_sparse,
                      "polygons_to_label": g2["polygons_to_label"], "dist_to_coord": g2["dist_to_coord"]})
    g3 = ref_functions("geometry/geom3d.py", {"polyhedron_to_label"}, {"np": np, "c_polyhedron_to_label": m3.c_polyhedron_to_label})
    rs = ref_functions("matching.py", {"relabel_sequential"}, {"np": np})
    r3 = _ref_method("models/model3d.py", "StarDist3D", "_instances_from_prediction",
                     {"np": np, "rays_from_json": ref_rays.rays_from_json, "non_maximum_suppression_3d": ref_nms.non_maximum_suppression_3d,
                      "non_maximum_suppression_3d_sparse": ref_nms.non_maximum_suppression_3d_sparse, "polyhedron_to_label": g3["polyhedron_to_label"],
                      "relabel_sequential": rs["relabel_sequential"]})
    rng = np.random.RandomState(8)

    def same_dict(a, b, tag):
        assert set(a) == set(b), (tag, sorted(a), sorted(b))
        for k in a:
            if k == "rays":
                assert np.array_equal(a[k].vertices, b[k].vertices) and np.array_equal(a[k].faces, b[k].faces), tag
                continue
            x, y = np.asarray(a[k]), np.asarray(b[k])
            assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y), (tag, k, x.dtype, y.dtype, x.shape, y.shape)

    for it in range(8):
        grid = tuple(int(v) for v in rng.choice([1, 2], 2))
        n_cls = None if it % 2 else 2
        cfg = Config2D(n_rays=16, grid=grid, unet_n_depth=1, unet_n_filter_base=4, n_classes=n_cls)
        mine = StarDist2D(cfg, basedir=None, device="cpu")
        mine.thresholds = dict(prob=0.7, nms=0.4)
        fake = types.SimpleNamespace(thresholds=Thr(0.7, 0.4), config=types.SimpleNamespace(grid=grid))
        H, W = 40, 52
        dist = (5 * (1 + 0.3 * rng.uniform(-1, 1, (H, W, 16)))).astype(np.float32)
        prob = rng.uniform(0, 1, (H, W)).astype(np.float32)
        pc = None if n_cls is None else rng.dirichlet(np.ones(3), (H, W)).astype(np.float32)
        shape = (H * grid[0], W * grid[1])
        scale = None if it < 4 else dict(Y=0.5, X=2.0)
        la, da = mine._instances_from_prediction(shape, prob, dist, prob_class=pc, scale=scale)
        lb, db = r2(fake, shape, prob, dist, prob_class=pc, scale=scale)
        assert la.dtype == lb.dtype and np.array_equal(la, lb), ("2d dense", it)
        same_dict(da, db, ("2d dense", it))
        mask = prob > 0.75
        pts = np.stack(np.where(mask), 1) * np.array(grid).reshape(1, 2)
        pcs = None if pc is None else pc[mask]
        la, da = mine._instances_from_prediction(shape, prob[mask], dist[mask], points=pts, prob_class=pcs, scale=scale, return_labels=bool(it % 3))
        lb, db = r2(fake, shape, prob[mask], dist[mask], points=pts, prob_class=pcs, scale=scale, return_labels=bool(it % 3))
        assert (la is None and lb is None) or np.array_equal(la, lb), ("2d sparse", it)
        same_dict(da, db, ("2d sparse", it))
    with pytest.raises(NotImplementedError):
        mine._instances_from_prediction(shape, prob, dist, overlap_label=-1)

    for it in range(6):
        grid = tuple(int(v) for v in rng.choice([1, 2], 3))
        n_cls = None if it % 2 else 2
        an = None if it % 3 else (2, 1, 1)
        cfg = Config3D(rays=Rays_GoldenSpiral(16, anisotropy=an), grid=grid, anisotropy=an, unet_n_depth=1, unet_n_filter_base=4, n_classes=n_cls)
        mine = StarDist3D(cfg, basedir=None, device="cpu")
        mine.thresholds = dict(prob=0.8, nms=0.3)
        fake = types.SimpleNamespace(thresholds=Thr(0.8, 0.3), config=types.SimpleNamespace(grid=grid, rays_json=cfg.rays_json))
        S = (10, 12, 14)
        dist = (3.5 * (1 + 0.2 * rng.uniform(-1, 1, S + (16,)))).astype(np.float32)
        prob = rng.uniform(0, 1, S).astype(np.float32)
        pc = None if n_cls is None else rng.dirichlet(np.ones(3), S).astype(np.float32)
        shape = tuple(s * g for s, g in zip(S, grid))
        scale = None if it < 3 else dict(Z=0.5, Y=2.0, X=1.0)
        ol = None if it % 2 else -1
        la, da = mine._instances_from_prediction(shape, prob, dist, prob_class=pc, scale=scale, overlap_label=ol)
        lb, db = r3(fake, shape, prob, dist, prob_class=pc, scale=scale, overlap_label=ol)
        assert la.dtype == lb.dtype and np.array_equal(la, lb), ("3d dense", it, la.dtype, lb.dtype)
        same_dict(da, db, ("3d dense", it))
        mask = prob > 0.85
        pts = np.stack(np.where(mask), 1) * np.array(grid).reshape(1, 3)
        pcs = None if pc is None else pc[mask]
        la, da = mine._instances_from_prediction(shape, prob[mask], dist[mask], points=pts, prob_class=pcs, scale=scale, return_labels=bool(it % 3))
        lb, db = r3(fake, shape, prob[mask], dist[mask], points=pts, prob_class=pcs, scale=scale, return_labels=bool(it % 3))
        assert (la is None and lb is None) or (la.dtype == lb.dtype and np.array_equal(la, lb)), ("3d sparse", it)
        same_dict(da, db, ("3d sparse", it))


def test_predict_instances_big_loop_equals_the_reference_method(ref_big, monkeypatch):
    """StarDistBase.predict_instances_big (models/base.py:838-983) end to end -- blocks, context crop, responsibility filter, label offsets,
    the written label image and the merged object dict -- against the reference's own method, both driven by the same stand-in model whose
    predict_instances returns the ground-truth objects of a block.  (regionprops' label / bbox / image for the reference's filter come from
    scipy.ndimage.find_objects here; the real scikit-image pins them in tests/golden/make_big_filter_golden.py.)"""
    from scipy import ndimage as ndi
    from stardist_amd import big as B
    from stardist_amd.models.config import Config2D, Config3D

    class Reg(object):
        def __init__(self, lab, sl, labels):
            self.label, self.bbox, self.image = lab, tuple(s.start for s in sl) + tuple(s.stop for s in sl), labels[sl] == lab
    monkeypatch.setattr(ref_big, "regionprops", lambda labels: [Reg(i + 1, sl, labels) for i, sl in enumerate(ndi.find_objects(labels)) if sl is not None])
    rs = ref_functions("matching.py", {"relabel_sequential"}, {"np": np})
    mm = types.ModuleType("_ref_sd.matching"); mm.relabel_sequential = rs["relabel_sequential"]
    pk, pm = types.ModuleType("_ref_sd"), types.ModuleType("_ref_sd.models")
    pk.__path__, pm.__path__ = [], []
    for k, v in {"_ref_sd": pk, "_ref_sd.models": pm, "_ref_sd.big": ref_big, "_ref_sd.matching": mm}.items():
        monkeypatch.setitem(sys.modules, k, v)

    def axes_check_and_normalize(axes, length=None, **k):
        axes = str(axes).upper()
        assert length is None or len(axes) == length
        return axes

    def axes_dict(axes):
        return {a: (axes.find(a) if a in axes else None) for a in "STCZYX"}
    method = _ref_method("models/base.py", "StarDistBase", "predict_instances_big",
                         {"np": np, "tqdm": lambda it, **k: it, "_raise": _raise, "axes_check_and_normalize": axes_check_and_normalize, "axes_dict": axes_dict,
                          "__package__": "_ref_sd.models", "__name__": "_ref_sd.models.base"})

    class Model(object):
        def __init__(self, nd, grid):
            self.config = Config2D() if nd == 2 else Config3D()
            self._grid, self._axes_out = grid, self.config.axes
            self.calls = []

        def _axes_div_by(self, axes): return tuple(self._grid if a != "C" else 1 for a in axes)

        def _axes_tile_overlap(self, axes): return tuple(0 for a in axes)

        def predict_instances(self, x, **kwargs):
            self.calls.append(sorted(kwargs))
            x = np.asarray(x)
            if x.ndim > len(self.config.axes) - 1:
                x = x[..., 0]
            ids = np.unique(x); ids = ids[ids > 0]
            lab = np.zeros(x.shape, np.int32)
            for j, v in enumerate(ids, 1):
                lab[x == v] = j
            objs = ndi.find_objects(lab)
            pts = np.array([[0.5 * (s.start + s.stop) for s in o] for o in objs]).reshape(len(objs), x.ndim)
            polys = dict(points=pts, prob=np.linspace(1, 0.5, len(objs)), coord=np.zeros((len(objs), x.ndim, 4)) + pts[:, :, None])
            if x.ndim == 3:
                polys.update(dist=np.ones((len(objs), 5)), rays_vertices=np.eye(3))
            return lab, polys
    rng = np.random.RandomState(9)
    done = 0
    for it in range(90):
        nd = 2 if it % 2 == 0 else 3
        shape = tuple(int(v) for v in rng.randint(60, 140, nd)) if nd == 2 else tuple(int(v) for v in rng.randint(36, 60, nd))
        gt = np.zeros(shape, np.int32)
        grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
        k = 0
        for _ in range(40 if nd == 2 else 25):
            c = [rng.uniform(4, s - 4) for s in shape]; r = rng.uniform(2, 4)
            m = sum((g - ci) ** 2 for g, ci in zip(grids, c)) <= r * r
            if (gt[m] == 0).all():
                k += 1; gt[m] = k
        grid = int(rng.choice([1, 2]))
        with_c = it % 5 == 0
        axes = ("YX" if nd == 2 else "ZYX") + ("C" if with_c else "")
        img = gt[..., None].repeat(2, -1) if with_c else gt
        bs = int(rng.choice([32, 40, 48])) if nd == 2 else int(rng.choice([24, 32]))
        mo, ctx = int(rng.choice([12, 16])), int(rng.choice([0, 2, 4]))
        kw = dict(block_size=bs, min_overlap=mo, context=ctx, show_progress=False)
        extra = dict(labels_out_dtype=np.int64) if it % 7 == 0 else (dict(labels_out=False) if it % 11 == 0 else {})
        a_model, b_model = Model(nd, grid), Model(nd, grid)
        try:
            lb, pb = method(b_model, img, axes, **kw, **extra)
        except Exception as e:                                                  # noqa: BLE001 -- the reference's cover refuses many size / block combinations (big.py:186, :270)
            with pytest.raises(type(e)):
                B.predict_instances_big(a_model, img, axes, distributed=False, **kw, **extra)
            continue
        la, pa = B.predict_instances_big(a_model, img, axes, distributed=False, **kw, **extra)
        if lb is None:
            assert la is False or la is None
        else:
            assert la.dtype == lb.dtype and np.array_equal(la, lb), (it, shape, axes, kw)
        assert set(pa) == set(pb)
        for key in pa:
            assert np.asarray(pa[key]).shape == np.asarray(pb[key]).shape and np.array_equal(pa[key], pb[key]), (it, key)
        assert a_model.calls == b_model.calls                                   # the same keyword arguments reach predict_instances
        done += 1
    assert done >= 30, done


def test_model_registry_equals_the_reference_registrations():
    """models/__init__.py:19-27: the same keys, URLs, md5 sums and aliases, per model class"""
    from stardist_amd.models import pretrained as P
    want_models, want_aliases = {}, {}
    for node in ast.walk(ast.parse(open(os.path.join(REF, "models", "__init__.py")).read())):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) in ("register_model", "register_aliases"):
            args = [a.id if isinstance(a, ast.Name) else ast.literal_eval(a) for a in node.args]
            if node.func.id == "register_model":
                want_models.setdefault(args[0], {})[args[1]] = dict(url=args[2], md5=args[3])
            else:
                for alias in args[2:]:
                    want_aliases.setdefault(args[0], {})[alias] = args[1]
    assert want_models and want_aliases
    assert {c: dict(m) for c, m in P._MODELS.items()} == want_models
    assert {c: list(m) for c, m in P._MODELS.items()} == {c: list(m) for c, m in want_models.items()}         # registration order (what the listing prints)
    assert P._ALIASES == want_aliases


def test_legacy_nms_and_label_glue_equal_the_reference_functions(ref_nms, ref_rays, monkeypatch):
    """the remaining Python glue around natives, natives shared: `_non_maximum_suppression_old` (nms.py:20-74: mapping image, score order),
    `polygons_to_label` / `polygons_to_label_coord` / `_polygons_to_label_old` (geom2d.py:112-197: prob filter, paint order, label ids) and
    `polyhedron_to_label` (geom3d.py:100-198: filter, order, labels, modes, overlap label, empty inputs and its error messages)"""
    from oracle import port, ref
    from stardist_amd import nms as NM
    from stardist_amd.geometry import geom2d, geom3d
    from stardist_amd.lib import stardist2d as sd2, stardist3d as sd3
    from stardist_amd.rays3d import Rays_GoldenSpiral
    from stardist_amd.utils import _normalize_grid
    m2, m3 = ref.stardist2d(), ref.stardist3d()
    f32, i32 = (lambda a: np.ascontiguousarray(a, np.float32)), (lambda a: np.ascontiguousarray(a, np.int32))
    monkeypatch.setattr(sd2, "c_non_max_suppression_inds_old", lambda polys, mapping, t, mb, gy, gx, v: m2.c_non_max_suppression_inds_old(
        i32(polys), i32(mapping), np.float32(t), np.int32(mb), np.int32(gy), np.int32(gx), np.int32(v)).astype(bool))
    monkeypatch.setattr(sd2, "c_polygons_to_label", lambda coord, labels, shape, window=None: port.polygons_to_label_coord(coord, shape, labels=labels))
    monkeypatch.setattr(sd3, "c_polyhedron_to_label", lambda d, p, V, F, l, mode, vb, uo, ol, shape, window=None: m3.c_polyhedron_to_label(f32(d), f32(p), f32(V), i32(F), i32(l), int(mode), int(vb), int(uo), int(ol), tuple(shape)))
    g2 = ref_functions("geometry/geom2d.py", {"ray_angles", "dist_to_coord", "polygons_to_label_coord", "polygons_to_label", "_polygons_to_label_old", "_dist_to_coord_old"},
                       {"np": np, "polygon": port.polygon, "_check_label_array": lambda *a, **k: True, "_normalize_grid": _normalize_grid})
    g3 = ref_functions("geometry/geom3d.py", {"polyhedron_to_label"}, {"np": np, "c_polyhedron_to_label": m3.c_polyhedron_to_label})
    rng = np.random.RandomState(10)
    for it in range(8):
        R = int(rng.choice([8, 16, 32]))
        grid = tuple(int(v) for v in rng.choice([1, 2], 2))
        H, W = int(rng.randint(24, 48)), int(rng.randint(24, 48))
        rhos = (5 * (1 + 0.3 * rng.uniform(-1, 1, (H, W, R)))).astype(np.float32)
        prob = rng.uniform(0, 1, (H, W)).astype(np.float32)
        coord = g2["_dist_to_coord_old"](rhos, grid=grid)
        for mb in (True, False):
            kw = dict(grid=grid, b=int(rng.choice([0, 2])), nms_thresh=0.4, prob_thresh=0.7, max_bbox_search=mb)
            a, b = NM._non_maximum_suppression_old(coord, prob, **kw), ref_nms._non_maximum_suppression_old(coord, prob, **kw)
            assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b), ("old nms", it, kw)
        pts = a * np.array(grid).reshape(1, 2)
        shape = (H * grid[0], W * grid[1])
        d = rhos[tuple(a.T)]
        pr = prob[tuple(a.T)]
        for kw in (dict(), dict(prob=pr), dict(prob=pr, thr=0.8), dict(scale_dist=(0.5, 2.0))):
            x, y = geom2d.polygons_to_label(d, pts, shape, **kw), g2["polygons_to_label"](d, pts, shape, **kw)
            assert x.dtype == y.dtype and np.array_equal(x, y), ("polygons_to_label", it, sorted(kw))
        cc = g2["dist_to_coord"](d, pts)
        labs = rng.permutation(len(cc))
        for lab in (None, labs):
            x, y = geom2d.polygons_to_label_coord(cc, shape, labels=lab), g2["polygons_to_label_coord"](cc, shape, labels=lab)
            assert x.dtype == y.dtype and np.array_equal(x, y), ("polygons_to_label_coord", it)
        x, y = geom2d._polygons_to_label_old(coord, prob, a, shape=shape, thr=0.75), g2["_polygons_to_label_old"](coord, prob, a, shape=shape, thr=0.75)
        assert x.dtype == y.dtype and np.array_equal(x, y), ("_polygons_to_label_old", it)
    for it in range(6):
        an = None if it % 2 else (2, 1, 1)
        rays, rrays = Rays_GoldenSpiral(16, anisotropy=an), ref_rays.Rays_GoldenSpiral(16, anisotropy=an)
        n = 12
        pts = rng.randint(4, 26, (n, 3)); d = rng.uniform(2, 5, (n, 16)).astype(np.float32); pr = rng.uniform(0, 1, n)
        shape = (30, 30, 30)
        for kw in (dict(verbose=False), dict(prob=pr, verbose=False), dict(prob=pr, thr=0.5, verbose=False), dict(labels=np.arange(7, 7 + n), verbose=False),
                   dict(mode="kernel", verbose=False), dict(mode="hull", verbose=False), dict(mode="bbox", verbose=False), dict(overlap_label=-1, verbose=False),
                   dict(prob=pr, thr=2.0, verbose=False)):
            x, y = geom3d.polyhedron_to_label(d, pts, rays, shape, **kw), g3["polyhedron_to_label"](d, pts, rrays, shape, **kw)
            assert np.asarray(x).shape == y.shape and np.array_equal(np.asarray(x), y), ("polyhedron_to_label", it, sorted(kw))
        x, y = geom3d.polyhedron_to_label(d[:0], pts[:0], rays, shape, verbose=False), g3["polyhedron_to_label"](d[:0], pts[:0], rrays, shape, verbose=False)
        assert x.dtype == y.dtype and np.array_equal(x, y)
        for bad in (dict(d=-d), dict(d=d[:, :5]), dict(prob=pr[:3]), dict(labels=np.arange(3)), dict(mode="nope")):
            kw = dict(verbose=False); dd = bad.get("d", d)
            kw.update({k: v for k, v in bad.items() if k != "d"})
            with pytest.raises(Exception) as e1:
                g3["polyhedron_to_label"](dd, pts, rrays, shape, **kw)
            with pytest.raises(type(e1.value)) as e2:
                geom3d.polyhedron_to_label(dd, pts, rays, shape, **kw)
            assert str(e1.value) == str(e2.value), bad


def test_public_signatures_follow_the_reference():
    """every function / method the mirror shares by name with the reference's modules takes the reference's parameters, in its order, with its
    defaults -- except where this file says otherwise (the deliberate differences)"""
    import importlib
    import inspect
    allowed = {
        ("geometry/geom2d.py", "star_dist", "mode"): "'hip' replaces 'cpp' / 'opencl' (both still accepted)",
        ("geometry/geom3d.py", "star_dist3D", "mode"): "same",
        ("big.py", "Block.__init__", None): "value object instead of a linked chain (same cover, tests above)",
        ("rays3d.py", "Rays_SubDivide.split", None): "classmethod in both; the reference names its first parameter 'self'",
        ("models/base.py", "StarDistBase._predict_setup", None): "internal; progress / predict_kwargs handled by the callers here",
    }

    def ref_sigs(rel):
        out = {}

        def sig(f):
            a = f.args
            names = [x.arg for x in a.posonlyargs + a.args]
            defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
            return list(zip(names, defaults)), a.kwarg is not None
        for n in ast.parse(open(os.path.join(REF, rel)).read()).body:
            if isinstance(n, ast.FunctionDef):
                out[n.name] = sig(n)
            if isinstance(n, ast.ClassDef):
                for f in n.body:
                    if isinstance(f, ast.FunctionDef):
                        out[n.name + "." + f.name] = sig(f)
        return out

    def value(expr):
        try:
            return eval(expr, {"np": np, "ZIP_DEFLATED": zipfile.ZIP_DEFLATED})                      # noqa: S307 -- literals of the reference's signatures
        except Exception:                                                                            # noqa: BLE001 -- e.g. Config2D()
            return expr
    checked = 0
    for rel, modname in (("nms.py", "stardist_amd.nms"), ("geometry/geom2d.py", "stardist_amd.geometry.geom2d"), ("geometry/geom3d.py", "stardist_amd.geometry.geom3d"),
                         ("utils.py", "stardist_amd.utils"), ("rays3d.py", "stardist_amd.rays3d"), ("big.py", "stardist_amd.big"), ("matching.py", "stardist_amd.matching"),
                         ("models/base.py", "stardist_amd.models.base"), ("models/model2d.py", "stardist_amd.models.model2d"), ("models/model3d.py", "stardist_amd.models.model3d")):
        m = importlib.import_module(modname)
        for name, (params, has_kw) in ref_sigs(rel).items():
            leaf = name.split(".")[-1]
            if (leaf.startswith("__") and leaf != "__init__") or (rel, name, None) in allowed:
                continue
            obj = m
            try:
                for part in name.split("."):
                    obj = getattr(obj, part)
            except AttributeError:
                continue                                                    # not part of the prediction path (out of scope)
            if not callable(obj):
                continue
            mine = inspect.signature(obj).parameters
            mine_pos = [p for p in mine.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
            mine_kw = any(p.kind == p.VAR_KEYWORD for p in mine.values())
            if inspect.ismethod(obj) or (leaf != "__init__" and "." in name and params and params[0][0] in ("self", "cls") and (not mine_pos or mine_pos[0].name not in ("self", "cls"))):
                params = params[1:]                                         # bound classmethod / staticmethod view
            names = [p.name for p in mine_pos]
            ref_names = [n for n, _ in params]
            assert [n for n in ref_names if n in names] == ref_names or mine_kw, (rel, name, "missing", [n for n in ref_names if n not in names])
            assert [n for n in names if n in ref_names] == [n for n in ref_names if n in names], (rel, name, "order")
            for n, d in params:
                if n not in mine or d is None or (rel, name, n) in allowed:
                    continue
                got, want = mine[n].default, value(d)
                if isinstance(want, str) and want.endswith("()"):           # a default instance (Config2D()): same class name
                    assert type(got).__name__ == want[:-2], (rel, name, n)
                    continue
                same = (got == want) if not isinstance(want, float) else (got == want or (got != got and want != want))
                assert same or (isinstance(want, tuple) and tuple(got) == want), (rel, name, n, got, want)
            assert not has_kw or mine_kw or name.endswith("Resizer.__init__"), (rel, name, "**kwargs")
            checked += 1
    assert checked > 80, checked


def test_guess_n_tiles_equals_the_reference_method():
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    from stardist_amd.models.base import axes_check_and_normalize, axes_dict
    norm = _ref_method("models/base.py", "StarDistBase", "_normalize_axes", {"np": np, "axes_check_and_normalize": axes_check_and_normalize})
    ref = _ref_method("models/base.py", "StarDistBase", "_guess_n_tiles", {"np": np, "axes_dict": axes_dict})
    for cls, cfg, shapes in ((StarDist2D, Config2D(n_rays=8, unet_n_depth=1, unet_n_filter_base=4), [(300, 700), (2048, 2048), (100, 90)]),
                             (StarDist2D, Config2D(n_rays=8, unet_n_depth=1, unet_n_filter_base=4, n_channel_in=3, train_patch_size=(128, 64), train_batch_size=2), [(500, 400, 3)]),
                             (StarDist3D, Config3D(rays=8, unet_n_depth=1, unet_n_filter_base=4, train_batch_size=2), [(64, 300, 300), (20, 20, 20)])):
        m = cls(cfg, basedir=None, device="cpu")
        fake = types.SimpleNamespace(config=cfg)
        fake._normalize_axes = lambda img, axes, _f=fake: norm(_f, img, axes)
        for sh in shapes:
            x = np.zeros(sh, np.float32)
            assert m._guess_n_tiles(x) == ref(fake, x), (cls.__name__, sh)


def test_label_image_helpers_equal_the_reference_functions():
    """fill_label_holes, calculate_extents, sample_points (stardist/utils.py:128-193; exported by the reference package's __init__)"""
    from collections.abc import Iterable
    from scipy import ndimage as ndi
    from stardist_amd import utils as U

    class Reg(object):
        def __init__(self, sl): self.bbox = tuple(s.start for s in sl) + tuple(s.stop for s in sl)
    ns = ref_functions("utils.py", {"fill_label_holes", "_fill_label_holes", "sample_points", "calculate_extents"},
                       {"np": np, "find_objects": ndi.find_objects, "binary_fill_holes": ndi.binary_fill_holes, "Iterable": Iterable, "_raise": _raise,
                        "regionprops": lambda lbl: [Reg(sl) for sl in ndi.find_objects(lbl) if sl is not None]})
    rng = np.random.RandomState(11)
    imgs = []
    for it in range(40):
        nd = 2 if it % 3 else 3
        shape = tuple(int(v) for v in (rng.randint(20, 60, nd) if nd == 2 else rng.randint(10, 24, nd)))
        grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
        lbl = np.zeros(shape, np.uint16 if it % 2 else np.int32)
        for lab in rng.permutation(np.arange(1, 9))[:rng.randint(1, 7)]:
            c = [rng.uniform(-2, s + 2) for s in shape]; r = rng.uniform(3, 9)
            d2 = sum((g - ci) ** 2 for g, ci in zip(grids, c))
            lbl[(d2 <= r * r) & (d2 >= (0.45 * r) ** 2) & (lbl == 0)] = lab            # rings / shells: holes, some cut by the border
        imgs.append(lbl)
        a, b = U.fill_label_holes(lbl), ns["fill_label_holes"](lbl)
        assert a.dtype == b.dtype and np.array_equal(a, b), it
        for func in (np.median, np.mean, np.max):
            x, y = U.calculate_extents(lbl, func), ns["calculate_extents"](lbl, func)
            assert np.array_equal(x, y) and x.dtype == y.dtype, (it, func.__name__)
    two_d = [im for im in imgs if im.ndim == 2]
    assert np.array_equal(U.calculate_extents(two_d), ns["calculate_extents"](two_d))
    stack = np.stack([im[:10, :10, :10] for im in imgs if im.ndim == 3])
    assert np.array_equal(U.calculate_extents(stack), ns["calculate_extents"](stack))
    assert np.array_equal(U.calculate_extents(np.zeros((5, 5), np.int32)), ns["calculate_extents"](np.zeros((5, 5), np.int32)))
    with pytest.raises(ValueError):
        U.calculate_extents(np.zeros(5, np.int32))
    hole = np.zeros((12, 12), bool); hole[2:10, 2:10] = True; hole[5:7, 5:7] = False
    kw = dict(structure=np.ones((3, 3), bool))
    assert np.array_equal(U.fill_label_holes(hole.astype(np.int32), **kw), ns["fill_label_holes"](hole.astype(np.int32), **kw))
    mask = rng.uniform(0, 1, (40, 50)) > 0.6
    prob = rng.uniform(0, 1, (40, 50))
    for kw in (dict(), dict(prob=prob), dict(b=0), dict(b=None, prob=prob), dict(b=5)):
        np.random.seed(3); a = U.sample_points(25, mask, **kw)
        np.random.seed(3); b = ns["sample_points"](25, mask, **kw)
        assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b), sorted(kw)


def test_matching_metrics_equal_the_reference_module(monkeypatch):
    """stardist/matching.py (metrics: out of the hot path, kept for code written against the module) loaded as a module -- numba's jit
    stood in for by the identity, skimage / csbdeep by stubs -- against stardist_amd.matching: label_overlap, the three criteria,
    matching (scalar / several thresholds, report_matches, non-sequential ids, empty images), matching_dataset (summed and by_image)"""
    import stardist_amd.matching as mine
    stubs = {"numba": types.ModuleType("numba"), "skimage": types.ModuleType("skimage"), "skimage.measure": types.ModuleType("skimage.measure"),
             "csbdeep": types.ModuleType("csbdeep"), "csbdeep.utils": types.ModuleType("csbdeep.utils")}
    stubs["numba"].jit = lambda *a, **k: (lambda f: f)
    def regionprops(y):                                                # the two attributes the reference reads: .label, .slice (ascending labels)
        from scipy.ndimage import find_objects
        return [types.SimpleNamespace(label=i, slice=sl) for i, sl in enumerate(find_objects(y), 1) if sl is not None]
    stubs["skimage.measure"].regionprops = regionprops
    stubs["csbdeep.utils"]._raise = _raise
    for k, v in stubs.items():
        monkeypatch.setitem(sys.modules, k, v)
    spec = importlib.util.spec_from_file_location("_ref_matching", os.path.join(REF, "matching.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.RandomState(17)

    def blobs(shape, n, ids):
        y = np.zeros(shape, np.int32)
        for k in range(n):
            c = [rng.randint(0, s) for s in shape]
            r = rng.randint(2, 7)
            sl = tuple(slice(max(0, ci - r), ci + r) for ci in c)
            y[sl] = ids[k]
        return y

    def same(a, b, tag):
        assert type(a).__name__ == type(b).__name__ and a._fields == b._fields, (tag, a._fields, b._fields)
        for f, x, y in zip(a._fields, a, b):
            if isinstance(y, tuple):
                assert len(x) == len(y) and all(np.allclose(u, v) for u, v in zip(x, y)), (tag, f)
            elif isinstance(y, str) or isinstance(y, bool):
                assert x == y, (tag, f)
            else:
                assert np.isclose(x, y, rtol=1e-6, atol=0) and type(x) == type(y), (tag, f, x, y, type(x), type(y))
    pairs = []
    for it in range(12):
        shape = (40, 48) if it % 3 else (12, 20, 16)
        nt, npred = rng.randint(0, 9), rng.randint(0, 9)
        yt = blobs(shape, nt, rng.permutation(np.arange(1, 40))[:nt] if it % 2 else np.arange(1, nt + 1))
        yp = blobs(shape, npred, rng.permutation(np.arange(1, 40))[:npred])
        if it % 4 == 1 and nt:
            yp = np.where(rng.uniform(size=shape) < 0.7, yt, yp).astype(np.int32)      # a prediction that mostly agrees
        pairs.append((yt, yp))
        if yt.max() > 0 and yp.max() > 0:
            a, b = mine.relabel_sequential(yt)[0], mine.relabel_sequential(yp)[0]
            ov = ref.label_overlap(a, b)
            assert np.array_equal(mine.label_overlap(a, b), ov) and mine.label_overlap(a, b).dtype == ov.dtype
            for name in ("intersection_over_union", "intersection_over_true", "intersection_over_pred"):
                x, y = getattr(mine, name)(ov), getattr(ref, name)(ov)
                assert x.dtype == y.dtype and np.array_equal(x, y), name
        for crit in ("iou", "iot", "iop"):
            for rm in (False, True):
                same(mine.matching(yt, yp, thresh=0.5, criterion=crit, report_matches=rm), ref.matching(yt, yp, thresh=0.5, criterion=crit, report_matches=rm), (it, crit, rm))
        for x, y in zip(mine.matching(yt, yp, thresh=(0.1, 0.5, 0.9)), ref.matching(yt, yp, thresh=(0.1, 0.5, 0.9))):
            same(x, y, (it, "thresholds"))
    two = [p for p in pairs if p[0].ndim == 2]
    for by_image in (False, True):
        for thr in (0.5, (0.3, 0.7)):
            a = mine.matching_dataset([p[0] for p in two], [p[1] for p in two], thresh=thr, by_image=by_image, show_progress=False)
            b = ref.matching_dataset([p[0] for p in two], [p[1] for p in two], thresh=thr, by_image=by_image, show_progress=False)
            for x, y in zip(a if isinstance(thr, tuple) else (a,), b if isinstance(thr, tuple) else (b,)):
                same(x, y, ("dataset", by_image, thr))
    for bad in (dict(criterion="dice"),):
        with pytest.raises(ValueError):
            mine.matching(pairs[0][0], pairs[0][1], **bad)
        with pytest.raises(ValueError):
            ref.matching(pairs[0][0], pairs[0][1], **bad)
    for fn in (mine.matching, ref.matching):
        with pytest.raises(ValueError):
            fn(pairs[0][0].astype(np.float32), pairs[0][1])
        with pytest.raises(ValueError):
            fn(pairs[0][0], pairs[0][1][:-1])
    assert all(mine.label_are_sequential(y) == ref.label_are_sequential(y) for p in pairs for y in p)
    # group_matching_labels (a moving scene: frame k + 1 = frame k rolled, one object dropped, ids shuffled), _shuffle_labels (same random state)
    base = blobs((48, 56), 7, np.arange(1, 8))
    frames = [base]
    for k in range(3):
        nxt = np.roll(frames[-1], 2, axis=1).copy()
        nxt[nxt == (k + 2)] = 0
        np.random.seed(k); a = mine._shuffle_labels(nxt)
        np.random.seed(k); b = ref._shuffle_labels(nxt)
        assert np.array_equal(a, b) and a.dtype == b.dtype
        frames.append(a)
    for ys in (frames, np.stack(frames)):
        ga, gb = mine.group_matching_labels(ys), ref.group_matching_labels(ys)
        assert ga.dtype == gb.dtype == np.int32 and np.array_equal(ga, gb)
    with pytest.raises(ValueError):
        mine.group_matching_labels(frames[:1])
    for t in ((5, 1, 2), (0, 3, 4), (7, 0, 0)):
        assert all(getattr(mine, f)(*t) == getattr(ref, f)(*t) for f in ("precision", "recall", "accuracy", "f1"))


def test_optimize_thresholds_equal_the_reference_functions(monkeypatch, tmp_path, capsys):
    """stardist.utils.optimize_threshold (utils.py:271-307) and StarDistBase.optimize_thresholds (base.py:986-1044) -- the producer of the
    thresholds.json the prediction path reads -- against the reference's own function / method, both driving the same stand-in model
    (instances = connected components of prob > prob_thresh, thinned by an nms-dependent size rule): the same evaluations, the same optimum,
    the same file"""
    import datetime
    import json
    import pathlib
    from collections import namedtuple
    from scipy import ndimage as ndi
    from scipy.optimize import minimize_scalar
    from tqdm import tqdm
    import stardist_amd.matching as mm
    from stardist_amd.models.base import StarDistBase
    from stardist_amd.utils import optimize_threshold
    ns = ref_functions("utils.py", {"optimize_threshold"}, {"np": np, "_raise": _raise, "tqdm": tqdm, "minimize_scalar": minimize_scalar, "datetime": datetime,
                                                            "matching_dataset": mm.matching_dataset})
    ref_opt = ns["optimize_threshold"]
    rng = np.random.RandomState(5)
    Y, Yhat = [], []
    for k in range(3):
        y = np.zeros((64, 72), np.int32)
        for i in range(1, 9):
            c = rng.randint(8, 56, 2)
            y[c[0] - 4:c[0] + 4, c[1] - 4:c[1] + 4] = i
        prob = ndi.gaussian_filter((y > 0).astype(np.float32), 2.0) * rng.uniform(0.8, 1.0)
        Y.append(y); Yhat.append((prob, np.zeros(y.shape + (4,), np.float32)))
    calls = []

    class Toy(object):
        basedir = "x"

        def __init__(self, logdir):
            self.logdir = logdir
            self._thr = None

        def _instances_from_prediction(self, shape, prob, dist, prob_thresh=None, nms_thresh=None):
            calls.append((float(prob_thresh), float(nms_thresh)))
            lab, n = ndi.label(prob > prob_thresh)
            sizes = ndi.sum(np.ones_like(lab), lab, index=np.arange(1, n + 1)) if n else np.zeros(0)
            for i, s in enumerate(sizes, 1):
                if s < 40 * nms_thresh:
                    lab[lab == i] = 0
            return lab.astype(np.int32), {}

        def predict(self, x, **kw):
            assert kw.get("n_tiles") == (1, 1) and kw.get("show_tile_progress") is False
            return Yhat[int(x[0, 0])] + ("prob_class",)

        def _guess_n_tiles(self, x):
            return (1, 1)
        thresholds = property(lambda self: self._thr, lambda self, d: setattr(self, "_thr", namedtuple("Thresholds", d.keys())(*d.values())))
    for nms in (0.3, 0.5):
        for kw in (dict(), dict(measure="f1", iou_threshs=0.5, tol=1e-3), dict(bracket=(0.2, 0.7), maxiter=5)):
            calls.clear(); a = optimize_threshold(Y, Yhat, Toy(None), nms, verbose=0, **kw); ca = list(calls)
            calls.clear(); b = ref_opt(Y, Yhat, Toy(None), nms, verbose=0, **kw); cb = list(calls)
            assert ca == cb and len(ca) > 3 and a == b and type(a[0]) == type(b[0]), (nms, kw, a, b)
    with pytest.raises(ValueError):
        optimize_threshold(Y, Yhat, Toy(None), (0.3, 0.4))
    # the method: same optimum, same messages, same thresholds.json
    X = [np.full((4, 4), k, np.float32) for k in range(3)]
    da, db = tmp_path / "a", tmp_path / "b"
    da.mkdir(); db.mkdir()
    import sys as _sys

    def save_json(data, fpath, **kw):
        with open(fpath, "w") as f:
            f.write(json.dumps(data, default=float, **kw))            # (csbdeep's save_json; under numpy 2 the optimum stays float32)
    rm = _ref_method("models/base.py", "StarDistBase", "optimize_thresholds", {"np": np, "sys": _sys, "optimize_threshold": ref_opt, "save_json": save_json})
    ta, tb = Toy(str(da)), Toy(pathlib.Path(db))
    capsys.readouterr()
    ra = StarDistBase.optimize_thresholds(ta, X, Y, optimize_kwargs=dict(verbose=0))
    oa = capsys.readouterr().out
    rb = rm(tb, X, Y, optimize_kwargs=dict(verbose=0))
    ob = capsys.readouterr().out
    assert ra == rb and oa == ob and "Using optimized values" in oa and "Saving to 'thresholds.json'." in oa
    assert ta.thresholds == tb.thresholds and json.load(open(da / "thresholds.json")) == json.load(open(db / "thresholds.json")) == ra
