"""CPU: predict_instances() END TO END against the reference's own code.  Reference side: StarDistBase._predict_instances_generator /
predict_instances (stardist/models/base.py:645-790) -> its own _predict_sparse_generator / _predict_generator on the graph its own
_build makes (tests/_mini_keras.py) -> StarDist2D / StarDist3D._instances_from_prediction (model2d.py:512-563, model3d.py:589-674) ->
the reference's nms.py over the COMPILED reference natives (oracle/_ref) -> geom2d.polygons_to_label (over the restatement of
skimage.draw.polygon that is pinned to the real one, profiles/r05_raster2d_oracle_vs_skimage.txt) / geom3d.polyhedron_to_label over
the compiled native.  All of it taken from the reference files at run time.

Mirror side: StarDist2D / StarDist3D.predict_instances with the same variables loaded by its own loader; there is no GPU here and the
product has no CPU path, so its natives are stood in for by the same compiled reference natives (NMS, 3D raster), the 2D raster
restatement and a numpy statement of the selection kernel's contract -- what is pinned is every line of HOST logic between them, in one
piece: argument handling, axes, pad / crop, candidate order, thresholds, grid, result dict, label ids, progress tokens.  The natives
themselves are pinned on the GPU (tests/test_gpu_*).  Build container only."""
import numbers
import os
import types

import numpy as np
import pytest
import scipy.ndimage as ndi

from test_cpu_reference_build import ref_methods
from test_cpu_reference_predict import _select_standin, mirror_model, reference_model
from test_cpu_vs_reference_source import REF, _raise, _ref_configs, ref_functions, ref_nms, ref_rays  # noqa: F401  (fixtures)

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference sources (build container only)")
f32, i32 = (lambda a: np.ascontiguousarray(a, np.float32)), (lambda a: np.ascontiguousarray(a, np.int32))


def oracle_natives(monkeypatch):
    """the mirror's native entry points -> the compiled reference natives / the pinned raster restatement (see module docstring)"""
    from oracle import port, ref
    from stardist_amd.lib import stardist2d as sd2, stardist3d as sd3
    m2, m3 = ref.stardist2d(), ref.stardist3d()

    def like(x, out):                                               # the mirror hands tensors through where it holds tensors
        import torch
        return torch.from_numpy(np.ascontiguousarray(out)) if torch.is_tensor(x) else out
    monkeypatch.setattr(sd2, "c_non_max_suppression_inds", lambda d, p, a, b, c, t, **k: m2.c_non_max_suppression_inds(f32(d), f32(p), int(a), int(b), int(c), np.float32(t)).astype(bool))
    monkeypatch.setattr(sd2, "c_polygons_to_label", lambda coord, labels, shape, window=None: port.polygons_to_label_coord(coord, shape, labels=labels))
    monkeypatch.setattr(sd3, "c_non_max_suppression_inds", lambda d, p, V, F, s, a, b, c, t, **k: like(d, m3.c_non_max_suppression_inds(f32(d), f32(p), f32(V), i32(F), f32(s), int(a), int(b), int(c), np.float32(t)).astype(bool)))
    monkeypatch.setattr(sd3, "c_polyhedron_to_label", lambda d, p, V, F, l, mode, vb, uo, ol, shape, window=None: like(d, m3.c_polyhedron_to_label(f32(d), f32(p), f32(V), i32(F), i32(l), int(mode), int(vb), int(uo), int(ol), tuple(shape))))
    return m2, m3


def complete_reference_model(nd, rcfg, ref_nms, ref_rays, m3):
    """reference_model + the reference's predict_instances and _instances_from_prediction"""
    from oracle import port
    ref, graph = reference_model(nd, rcfg)
    cls = type(ref)
    ns = dict(np=np, numbers=numbers, ndi=ndi, warnings=__import__("warnings"), functools=__import__("functools"), _raise=_raise)
    for name, fn in ref_methods("models/base.py", "StarDistBase", {"_predict_instances_generator", "predict_instances"}, ns).items():
        setattr(cls, name, fn)
    if nd == 2:
        g2 = ref_functions("geometry/geom2d.py", {"ray_angles", "dist_to_coord", "polygons_to_label_coord", "polygons_to_label"},
                           {"np": np, "polygon": port.polygon, "_check_label_array": lambda *a, **k: True})
        ns2 = {"np": np, "non_maximum_suppression": ref_nms.non_maximum_suppression, "non_maximum_suppression_sparse": ref_nms.non_maximum_suppression_sparse,
               "polygons_to_label": g2["polygons_to_label"], "dist_to_coord": g2["dist_to_coord"]}
        meth = ref_methods("models/model2d.py", "StarDist2D", {"_instances_from_prediction"}, ns2)
    else:
        g3 = ref_functions("geometry/geom3d.py", {"polyhedron_to_label"}, {"np": np, "c_polyhedron_to_label": m3.c_polyhedron_to_label})
        rs = ref_functions("matching.py", {"relabel_sequential"}, {"np": np})
        ns3 = {"np": np, "rays_from_json": ref_rays.rays_from_json, "non_maximum_suppression_3d": ref_nms.non_maximum_suppression_3d,
               "non_maximum_suppression_3d_sparse": ref_nms.non_maximum_suppression_3d_sparse, "polyhedron_to_label": g3["polyhedron_to_label"],
               "relabel_sequential": rs["relabel_sequential"]}
        meth = ref_methods("models/model3d.py", "StarDist3D", {"_instances_from_prediction"}, ns3)
    for name, fn in meth.items():
        setattr(cls, name, fn)
    return ref, graph


def shape_the_heads(graph, rs, radius, spread):
    """seeded random weights give distances around zero: re-scale the two 1x1 heads (in the graph's own variables, before they are
    written for the mirror) so that polygons of a few pixels and a usable share of candidates come out -- as bench.py does on the GPU"""
    for lay in graph.layers:
        if lay.name == "dist":
            lay.variables["dist/kernel:0"] *= np.float32(spread)
            lay.variables["dist/bias:0"] = (radius + 0.3 * rs.randn(*lay.variables["dist/bias:0"].shape)).astype(np.float32)


def tokens_and_result(model, img, **kw):
    out = list(model._predict_instances_generator(img, **kw))
    return [t for t in out[:-1]], out[-1]


def same_dict(a, b, tag, tol=2e-5):
    assert set(a) == set(b), (tag, sorted(a), sorted(b))
    for k in a:
        if k == "rays":
            assert np.array_equal(a[k].vertices, b[k].vertices) and np.array_equal(a[k].faces, b[k].faces), tag
            continue
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape, (tag, k, x.shape, y.shape)
        if np.issubdtype(y.dtype, np.floating):
            assert x.dtype == y.dtype, (tag, k, x.dtype, y.dtype)
            assert x.size == 0 or np.abs(x - y).max() <= tol * max(1.0, float(np.abs(y).max())), (tag, k, float(np.abs(x - y).max()))
        else:
            assert np.array_equal(x, y), (tag, k)


S2 = dict(n_rays=16, unet_n_filter_base=4, unet_n_depth=1, net_conv_after_unet=8)
S3 = dict(rays=16, unet_n_filter_base=4, unet_n_depth=1, net_conv_after_unet=8)


@pytest.mark.parametrize("kw,axes,shape", [(dict(S2), "YX", (70, 90)), (dict(S2, grid=(2, 2), n_channel_in=2), "YXC", (101, 83, 2)),
                                           (dict(S2, n_classes=2), "XY", (64, 75))])
def test_predict_instances_2d_end_to_end_equals_the_reference(kw, axes, shape, ref_nms, ref_rays, monkeypatch):
    m2, m3 = oracle_natives(monkeypatch)
    R2, _ = _ref_configs(ref_rays)
    rcfg = R2(**kw)
    ref, graph = complete_reference_model(2, rcfg, ref_nms, ref_rays, m3)
    shape_the_heads(graph, np.random.RandomState(3), radius=5.0, spread=1.5)
    m = mirror_model(2, kw, graph)
    m._select = _select_standin
    img = np.random.RandomState(12).uniform(-1, 1, shape).astype(np.float32)
    prob = ref.predict(img, axes=axes)[0]
    thr = float(np.sort(prob.ravel())[-max(40, prob.size // 12)])            # ~8 % of the pixels are candidates
    thr = float(np.nextafter(np.float32(thr), np.float32(0)))
    ref.thresholds = types.SimpleNamespace(prob=thr, nms=0.4)
    m.thresholds = dict(prob=thr, nms=0.4)

    variants = [dict(), dict(sparse=False), dict(return_labels=False), dict(nms_thresh=0.2, prob_thresh=min(0.999, thr + 0.02)),
                dict(return_predict=True), dict(nms_kwargs=dict(use_kdtree=False)), dict(nms_kwargs=dict(use_bbox=False)), dict(verbose=True), dict(sparse=False, nms_kwargs=dict(b=4)), dict(prob_thresh=0.99999), dict(prob_thresh=0.99999, sparse=False), dict(scale=2), dict(scale=tuple(1.5 if a == "Y" else (0.8 if a == "X" else 1) for a in axes))]
    for v in variants:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tw, want = tokens_and_result(ref, img, axes=axes, **v)
            tg, got = tokens_and_result(m, img, axes=axes, **v)
        assert tw == tg == ["predict", "nms"], (v, tw, tg)
        if v.get("return_predict"):
            (lw, dw), pw = want
            (lg, dg), pg = got
            assert len(pw) == len(pg)
            for a, b in zip(pg, pw):
                assert a.shape == b.shape and np.abs(a - b).max() <= 2e-5 * max(1.0, float(np.abs(b).max()))
        else:
            (lw, dw), (lg, dg) = want, got
        if "scale" in v:
            # the zoomed (interpolated) input makes near-ties among the scores; the mirror's float32 network and the float64 graph order a few
            # of them differently and the greedy NMS then keeps another polygon of a cluster: compare as sets
            pw_, pg_ = set(map(tuple, dw["points"].tolist())), set(map(tuple, dg["points"].tolist()))
            assert len(pw_ & pg_) >= 0.97 * len(pw_ | pg_), (v, len(pw_), len(pg_), len(pw_ & pg_))
            iw = {tuple(p): i for i, p in enumerate(dw["points"].tolist())}
            common = [(i, iw[tuple(p)]) for i, p in enumerate(dg["points"].tolist()) if tuple(p) in iw]
            gi, wi = np.array(common).T
            assert np.abs(dg["coord"][gi] - dw["coord"][wi]).max() <= 2e-5 * float(np.abs(dw["coord"]).max()), v
            assert dg["coord"].dtype == dw["coord"].dtype and dg["points"].dtype == dw["points"].dtype and dg["prob"].dtype == dw["prob"].dtype
            assert lg.shape == lw.shape == tuple(img.shape[axes.index(a)] for a in "YX") and (lg > 0).sum() >= 0.97 * ((lg > 0) | (lw > 0)).sum()
            continue
        same_dict(dg, dw, v)
        assert len(dw["prob"]) >= 10 or v.get("prob_thresh", 0) > 0.9999, (v, len(dw["prob"]))
        if v.get("return_labels", True):
            assert lg.dtype == lw.dtype and lg.shape == lw.shape, (v, lg.dtype, lw.dtype)
            # the float32 network of the mirror and the float64 graph differ in the 6th digit of a vertex: a pixel centre within that of an edge may flip
            assert (lg != lw).mean() <= 2e-4, (v, float((lg != lw).mean()))
            assert lg.max() == lw.max() == len(dw["prob"]) or len(dw["prob"]) == 0
        else:
            assert lg is None and lw is None


@pytest.mark.parametrize("kw,axes,shape,radius", [(dict(S3), "ZYX", (20, 34, 40), 3.5),
                                                  (dict(S3, grid=(1, 2, 2), n_classes=2, anisotropy=(2, 1, 1)), "ZYX", (12, 50, 44), 1.8)])
def test_predict_instances_3d_end_to_end_equals_the_reference(kw, axes, shape, radius, ref_nms, ref_rays, monkeypatch):
    m2, m3 = oracle_natives(monkeypatch)
    _, R3 = _ref_configs(ref_rays)
    rcfg = R3(**kw)
    ref, graph = complete_reference_model(3, rcfg, ref_nms, ref_rays, m3)
    shape_the_heads(graph, np.random.RandomState(4), radius=radius, spread=1.0)
    m = mirror_model(3, kw, graph)
    m._select = _select_standin
    img = np.random.RandomState(13).uniform(-1, 1, shape).astype(np.float32)
    prob = ref.predict(img, axes=axes)[0]
    thr = float(np.nextafter(np.float32(np.sort(prob.ravel())[-max(40, prob.size // 25)]), np.float32(0)))
    ref.thresholds = types.SimpleNamespace(prob=thr, nms=0.3)
    m.thresholds = dict(prob=thr, nms=0.3)
    for v in (dict(), dict(sparse=False), dict(return_labels=False), dict(overlap_label=-1), dict(nms_thresh=0.1), dict(prob_thresh=0.99999),
              dict(prob_thresh=0.99999, sparse=False)):
        tw, (lw, dw) = tokens_and_result(ref, img, axes=axes, **v)
        tg, (lg, dg) = tokens_and_result(m, img, axes=axes, **v)
        assert tw == tg == ["predict", "nms"], (v, tw, tg)
        same_dict(dg, dw, v)
        assert len(dw["prob"]) >= 2 or v.get("prob_thresh", 0) > 0.9999, (v, len(dw["prob"]))
        if v.get("return_labels", True):
            assert lg.dtype == lw.dtype and lg.shape == lw.shape, (v, lg.dtype, lw.dtype)
            assert (lg != lw).mean() <= 5e-4, (v, float((lg != lw).mean()))
        else:
            assert lg is None and lw is None


def test_cli_predict2d_on_a_model_folder_equals_the_api(tmp_path, monkeypatch, capsys):
    """CPU twin of tests/test_gpu_cli_multiclass.py::test_cli_predict2d_equals_api (natives stood in for as above): the command line script on a
    csbdeep-style model folder (config.json, thresholds.json, weights_best.npz) -- tiff in, label tiff out -- equals the API call"""
    import json
    from stardist_amd.models import Config2D, StarDist2D
    from stardist_amd.models.base import StarDistBase
    from stardist_amd.scripts import predict2d
    from stardist_amd.scripts._io import imread, imwrite
    from stardist_amd.utils import normalize
    oracle_natives(monkeypatch)
    monkeypatch.setattr(StarDistBase, "_select", staticmethod(_select_standin))
    cfg = Config2D(n_rays=16, unet_n_depth=1, unet_n_filter_base=4, net_conv_after_unet=8)
    model = StarDist2D(cfg, name="my_model", basedir=str(tmp_path), device="cpu", seed=3)           # writes my_model/config.json
    import torch
    with torch.no_grad():
        model.net.dist.bias.fill_(5.0); model.net.dist.weight.mul_(0.3)
    model.save_weights_npz(str(tmp_path / "my_model" / "weights_best.npz"))
    img = np.random.RandomState(2).uniform(0, 255, (72, 88)).astype(np.float32)
    x = normalize(img, 1, 99.8)
    thr = float(np.sort(model.predict(x)[0].ravel())[-400])
    (tmp_path / "my_model" / "thresholds.json").write_text(json.dumps(dict(prob=thr, nms=0.4)))
    imwrite(str(tmp_path / "in.tif"), img)
    rc = predict2d.main(["-i", str(tmp_path / "in.tif"), "-m", str(tmp_path / "my_model"), "-o", str(tmp_path / "out"), "--n_tiles", "2", "1", "--device", "cpu"])
    out = capsys.readouterr().out
    assert rc == 0 and "Loading network weights from 'weights_best.npz'." in out and "Loading thresholds from 'thresholds.json'." in out
    got = imread(str(tmp_path / "out" / "in.stardist.tif"))
    model.thresholds = dict(prob=thr, nms=0.4)
    want, res = model.predict_instances(normalize(imread(str(tmp_path / "in.tif")), 1, 99.8), n_tiles=(2, 1))
    assert got.shape == want.shape and np.array_equal(got, want) and want.max() > 5 and len(res["prob"]) == want.max()


@pytest.mark.parametrize("use_channel", [False, True])
def test_big_equals_whole_by_the_reference_s_own_acceptance_test(use_channel, monkeypatch):
    """tests/test_big.py:86-117 (`test_predict2D`: predict_instances_big == predict_instances by matching(thresh=.99) accuracy 1.0 and
    lexsorted polygons allclose(atol=1e-2)) on the CPU with a seeded network (weights are absent) and the natives stood in for; the same
    criteria for the block-sharded design A (predict_instances_sharded, one process), whose instances must be the very same"""
    import torch
    from stardist_amd.matching import matching
    from stardist_amd.models import Config2D, StarDist2D
    from stardist_amd.models.base import StarDistBase
    from stardist_amd.utils import normalize
    oracle_natives(monkeypatch)
    monkeypatch.setattr(StarDistBase, "_select", staticmethod(_select_standin))
    model = StarDist2D(Config2D(n_rays=16, unet_n_depth=1, unet_n_filter_base=4, net_conv_after_unet=8, n_channel_in=1), basedir=None, device="cpu", seed=5)
    with torch.no_grad():
        model.net.dist.bias.fill_(5.0); model.net.dist.weight.mul_(0.3)
    rs = np.random.RandomState(7)
    from scipy.ndimage import gaussian_filter
    img = normalize(gaussian_filter(rs.uniform(0, 1, (208, 256)), 2.0), 1, 99.8)          # smooth: blobs of candidates, as a real probability map has
    axes = "YX"
    if use_channel:
        img, axes = img[..., np.newaxis], "YXC"
    thr = float(np.sort(model.predict(img, axes=axes)[0].ravel())[-2500])
    model.thresholds = dict(prob=thr, nms=0.3)
    ref_labels, ref_polys = model.predict_instances(img, axes=axes)
    assert len(ref_polys["prob"]) > 30
    res_labels, res_polys = model.predict_instances_big(img, axes=axes, block_size=128, min_overlap=32, context=16, show_progress=False)
    m = matching(ref_labels, res_labels, thresh=0.99)
    assert (1.0, 1.0) == (m.accuracy, m.mean_true_score), m
    ri, si = np.lexsort(ref_polys["points"].T), np.lexsort(res_polys["points"].T)
    for k in ("coord", "points", "prob"):
        assert np.allclose(ref_polys[k][ri], res_polys[k][si], atol=1e-2), k
    sh_labels, sh_polys = model.predict_instances_sharded(img, axes, block_size=128, min_overlap=32, context=16)
    m = matching(ref_labels, np.asarray(sh_labels), thresh=0.99)
    assert (1.0, 1.0) == (m.accuracy, m.mean_true_score), m
    hi = np.lexsort(np.asarray(sh_polys["points"]).T)
    for k in ("coord", "points", "prob"):
        assert np.allclose(ref_polys[k][ri], np.asarray(sh_polys[k])[hi], atol=1e-2), k
    assert np.array_equal(np.asarray(sh_labels), ref_labels)                             # design A numbers its instances like predict_instances


def test_big_equals_whole_3d_by_the_reference_s_own_acceptance_test(monkeypatch):
    """tests/test_big.py:123-147 (`test_predict3D`: matching(thresh=.99) accuracy 1.0, mean_true_score > 0.999, lexsorted dist / points / prob
    allclose(atol=1e-2)) on the CPU, seeded network, natives stood in for; the same for design A"""
    import torch
    from scipy.ndimage import gaussian_filter
    from stardist_amd.matching import matching
    from stardist_amd.models import Config3D, StarDist3D
    from stardist_amd.models.base import StarDistBase
    from stardist_amd.utils import normalize
    oracle_natives(monkeypatch)
    monkeypatch.setattr(StarDistBase, "_select", staticmethod(_select_standin))
    model = StarDist3D(Config3D(rays=16, unet_n_depth=1, unet_n_filter_base=4, net_conv_after_unet=8), basedir=None, device="cpu", seed=6)
    with torch.no_grad():
        model.net.dist.bias.fill_(3.0); model.net.dist.weight.mul_(0.2)
    img = normalize(gaussian_filter(np.random.RandomState(8).uniform(0, 1, (40, 72, 80)), 1.5), 1, 99.8)
    v = np.sort(model.predict(img)[0].ravel())[-3100:-2900]                   # a threshold inside the widest gap near the 3000th largest value:
    k = int(np.argmax(np.diff(v)))                                             # the blocks' forward passes differ from the whole volume's in the last
    thr = float(0.5 * (v[k] + v[k + 1]))                                       # bits (other tile shapes), which must not move a voxel across it
    model.thresholds = dict(prob=thr, nms=0.3)
    ref_labels, ref_polys = model.predict_instances(img)
    assert len(ref_polys["prob"]) > 20
    # context >= the network's receptive field (10 here), as the default context (_axes_tile_overlap) is: design A takes a band candidate
    # from the first block that reports it, so both blocks must see it with full context (design B's responsibility rule is more forgiving)
    assert max(model._axes_tile_overlap("YX")) <= 12
    kw = dict(block_size=(40, 56, 56), min_overlap=(16, 16, 16), context=(8, 12, 12))        # (Z is one block)
    res_labels, res_polys = model.predict_instances_big(img, axes="ZYX", show_progress=False, **kw)
    m = matching(ref_labels, res_labels, thresh=0.99)
    assert m.accuracy == 1.0 and m.mean_true_score > 0.999, m
    ri, si = np.lexsort(ref_polys["points"].T), np.lexsort(res_polys["points"].T)
    for k in ("dist", "points", "prob"):
        assert np.allclose(ref_polys[k][ri], res_polys[k][si], atol=1e-2), k
    sh_labels, sh_polys = model.predict_instances_sharded(img, "ZYX", **kw)
    assert np.array_equal(np.asarray(sh_labels), ref_labels)
    hi = np.lexsort(np.asarray(sh_polys["points"]).T)
    for k in ("dist", "points", "prob"):
        assert np.allclose(ref_polys[k][ri], np.asarray(sh_polys[k])[hi], atol=1e-2), k
