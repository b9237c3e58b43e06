"""CPU: the reference's OWN prediction plumbing -- StarDistBase._predict_setup / _predict_generator / _predict_sparse_generator,
_normalize_axes, _compute_receptive_field / _axes_tile_overlap (stardist/models/base.py:371-633, 1058-1110), StarDist2D/3D._axes_div_by,
StarDistPadAndCropResizer, nms._ind_prob_thresh -- taken from the reference files at run time (nothing copied) and run on a model
object whose `keras_model` is the graph the reference's own _build constructs over the minimal Keras stand-in (tests/_mini_keras.py,
tests/test_cpu_reference_build.py).  The mirror, with the same variables loaded through its own weight loader, must return what the
reference's methods return: `predict` (prob, dist[, prob_class]) and `predict_sparse` (prob, dist[, prob_class], points) for inputs
in several axis layouts, channel counts, grids, extents that need the reflect-pad / crop, multi-class heads; its tile overlap must
cover the receptive field the reference measures.

csbdeep is absent (third party, reference setup.py:140), so four small names of it are restated here: axes_check_and_normalize,
axes_dict, BaseModel._make_permute_axes (forward direction) and the no-op normaliser of _check_normalizer_resizer; csbdeep's
tile_iterator is NOT restated -- the tiled branch of the reference is therefore not run (the mirror's tiling is tested against its own
untiled result, tests/test_cpu_host_logic.py).  Build container only."""
import functools
import io
import math
import os
import types
import warnings

import numpy as np
import pytest

import _mini_keras as K
from test_cpu_reference_build import ref_methods, reference_graph
from test_cpu_vs_reference_source import REF, _raise, _ref_configs, ref_functions, ref_rays  # noqa: F401  (ref_rays: fixture)

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference sources (build container only)")


# ---- csbdeep names, restated (see module docstring)
def axes_check_and_normalize(axes, length=None, disallowed=None, return_allowed=False):
    allowed = "STCZYX"
    axes = str(axes).upper()
    assert all(a in allowed for a in axes) and all(axes.count(a) == 1 for a in axes), axes
    assert length is None or len(axes) == length, (axes, length)
    return axes


def axes_dict(axes):
    axes = axes_check_and_normalize(axes)
    return {a: (None if axes.find(a) == -1 else axes.find(a)) for a in "STCZYX"}


def _make_permute_axes(self, img_axes_in, net_axes_in, net_axes_out=None, img_axes_out=None):
    def _permute_axes(data, undo=False):
        assert not undo
        src = img_axes_in
        if "C" not in src:                                          # move_image_axes(adjust_singletons=True): a missing axis is added
            data, src = data[..., np.newaxis], src + "C"
        return np.transpose(data, [src.index(a) for a in net_axes_in])
    return _permute_axes


class _NoNormalizer(object):
    def before(self, x, axes):
        return x


class _KerasModel(object):
    """keras.Model.predict on a batch of one: list of outputs with the batch axis"""

    def __init__(self, model):
        self.model = model

    def predict(self, x, **kwargs):
        assert x.shape[0] == 1 and set(kwargs) <= {"verbose"}
        return [y[np.newaxis].astype(np.float32) for y in self.model.predict(x[0])]      # (Keras hands float32 arrays back)


def reference_model(nd, rcfg, prob_thresh=0.5):
    """an object that carries the reference's own prediction methods, its Config object and the graph its own _build made"""
    graph = reference_graph(nd, rcfg)
    u = ref_functions("utils.py", {"_is_floatarray", "_is_power_of_2"}, {"np": np})
    n = ref_functions("nms.py", {"_ind_prob_thresh"}, {"np": np})
    ns = dict(np=np, warnings=warnings, math=math, functools=functools, _raise=_raise, axes_check_and_normalize=axes_check_and_normalize,
              axes_dict=axes_dict, _is_floatarray=u["_is_floatarray"], _is_power_of_2=u["_is_power_of_2"], _ind_prob_thresh=n["_ind_prob_thresh"],
              Resizer=object, tqdm=None, tile_iterator=None, total_n_tiles=None)
    ref_functions("models/base.py", {"StarDistPadAndCropResizer"}, ns)
    meth = ref_methods("models/base.py", "StarDistBase", {"_predict_setup", "_predict_generator", "predict", "_predict_sparse_generator", "predict_sparse",
                                                           "_is_multiclass", "_normalize_axes", "_compute_receptive_field", "_axes_tile_overlap"}, ns)
    meth.update(ref_methods("models/model%dd.py" % nd, "StarDist%dD" % nd, {"_axes_div_by"}, dict(np=np, _raise=_raise, axes_check_and_normalize=axes_check_and_normalize)))
    meth["_make_permute_axes"] = _make_permute_axes
    meth["_check_normalizer_resizer"] = lambda self, normalizer, resizer: (_NoNormalizer() if normalizer is None else normalizer, resizer)
    obj = type("RefModel", (), meth)()
    obj.config = rcfg
    obj.keras_model = _KerasModel(graph)
    obj.thresholds = types.SimpleNamespace(prob=prob_thresh, nms=0.4)
    return obj, graph


def mirror_model(nd, kw, graph):
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    m = (StarDist2D if nd == 2 else StarDist3D)((Config2D if nd == 2 else Config3D)(**kw), basedir=None, device="cpu")
    buf = io.BytesIO()
    np.savez(buf, **graph.weights_in_file_order())
    buf.seek(0)
    m.load_weights_npz(buf)
    m.net.eval()
    return m


def _select_standin(prob, dist, prob_thresh, bs):
    """the contract of the selection native (sd_select_candidates_device, csrc/select.hip: strict threshold, border of (lo, hi) grid steps per
    axis, np.where order, max(1e-3, dist)) as numpy -- there is no GPU here and the product has no CPU path; what this test pins is the
    HOST logic of predict_sparse around it (axes, pad / crop, grid scaling, filter_points, class rows); the native itself is pinned on
    the GPU (tests/test_gpu_glue.py)"""
    import torch
    p = prob.numpy()
    mask = p > np.float32(prob_thresh)
    inner = np.zeros_like(mask)
    inner[tuple(slice(lo if lo > 0 else None, -hi if hi > 0 else None) for lo, hi in bs)] = True
    mask &= inner
    pts = np.stack(np.nonzero(mask), 1).astype(np.int64)
    return torch.from_numpy(p[mask]), torch.from_numpy(np.maximum(np.float32(1e-3), dist.numpy()[mask])), torch.from_numpy(pts)


def _gap_threshold(prob, lo=0.35, hi=0.65):
    """a threshold in the middle of the widest gap between neighbouring probabilities in [lo, hi]: float32 vs float64 noise cannot move a pixel across"""
    v = np.sort(prob[(prob > lo) & (prob < hi)].ravel())
    if len(v) < 2:
        return 0.5
    k = int(np.argmax(np.diff(v)))
    return float(0.5 * (v[k] + v[k + 1]))


S2 = dict(n_rays=8, unet_n_filter_base=4, net_conv_after_unet=8)
S3 = dict(rays=8, unet_n_filter_base=4, net_conv_after_unet=8)
CASES = [
    # nd, config, image axes, image shape
    (2, dict(S2), "YX", (40, 56)),
    (2, dict(S2), "XY", (37, 50)),                                            # transposed input, extents that need the reflect-pad
    (2, dict(S2, n_channel_in=3), "CYX", (3, 33, 47)),
    (2, dict(S2, n_channel_in=3, grid=(2, 2)), "YXC", (45, 62, 3)),
    (2, dict(S2, grid=(4, 2), unet_n_depth=1), "YX", (50, 30)),               # pad < grid on one axis, >= grid on the other
    (2, dict(S2, n_classes=2, grid=(2, 2)), "YX", (36, 44)),
    (3, dict(S3, unet_n_depth=1), "ZYX", (9, 14, 19)),
    (3, dict(S3, unet_n_depth=1, grid=(1, 2, 2), n_channel_in=2, n_classes=2), "ZCYX", (6, 2, 13, 18)),
]


@pytest.mark.parametrize("nd,kw,axes,shape", CASES)
def test_predict_and_predict_sparse_equal_the_reference_methods(nd, kw, axes, shape, ref_rays):
    R2, R3 = _ref_configs(ref_rays)
    rcfg = (R2 if nd == 2 else R3)(**kw)
    ref, graph = reference_model(nd, rcfg)
    m = mirror_model(nd, kw, graph)
    img = np.random.RandomState(11).uniform(-1, 1, shape).astype(np.float32)

    # ---- predict
    want = ref.predict(img, axes=axes)
    got = m.predict(img, axes=axes)
    assert len(got) == len(want) == (2 if rcfg.n_classes is None else 3)
    for name, a, b in zip(("prob", "dist", "prob_class"), got, want):
        assert tuple(a.shape) == tuple(b.shape), (name, a.shape, b.shape)
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, float(np.abs(b).max())), (name, float(np.abs(a - b).max()))
    assert (want[1] >= 1e-3).all() and (np.asarray(got[1]) >= 1e-3).all()                                  # base.py:516
    # the default axes (config.axes without 'C' for a one-channel model) -- _normalize_axes
    if axes in ("YX", "ZYX"):
        assert np.array_equal(m.predict(img)[0], got[0])

    # ---- a normaliser object (csbdeep's protocol: .before(x, axes) on the array laid out as the network wants it), and integer input
    class Affine(object):
        def before(self, x, axes):
            assert axes == rcfg.axes
            return (x.astype(np.float32) - np.float32(0.25)) * np.float32(1.5)
    wn, gn = ref.predict(img, axes=axes, normalizer=Affine()), m.predict(img, axes=axes, normalizer=Affine())
    assert all(np.abs(a - b).max() <= 2e-5 * max(1.0, float(np.abs(b).max())) for a, b in zip(gn, wn))
    assert np.abs(wn[0] - want[0]).max() > 1e-4                                                          # (the normaliser did act)
    img8 = (127.5 * (img + 1)).astype(np.uint8)
    with pytest.warns(UserWarning, match="non-float input"):
        w8 = ref.predict(img8, axes=axes)
    with pytest.warns(UserWarning, match="non-float input"):
        g8 = m.predict(img8, axes=axes)
    assert all(np.abs(a - b).max() <= 1e-4 * max(1.0, float(np.abs(b).max())) for a, b in zip(g8, w8))

    # ---- predict_sparse: same candidates, row for row (np.where order), on a threshold no rounding can move a pixel across
    thr = _gap_threshold(want[0])
    m._select = _select_standin
    for b in (2, 0, ((1, 3),) * nd):
        ws = ref.predict_sparse(img, prob_thresh=thr, axes=axes, b=b)
        gs = m.predict_sparse(img, prob_thresh=thr, axes=axes, b=b)
        assert len(ws) == len(gs) == (3 if rcfg.n_classes is None else 4)
        assert np.array_equal(np.asarray(gs[-1]), ws[-1]), (b, len(gs[-1]), len(ws[-1]))               # points: identical integers, same order
        assert len(ws[0]) > 0 or b != 0, (thr, float(want[0].min()), float(want[0].max()))
        for a, w in zip(gs[:-1], ws[:-1]):
            assert a.shape == w.shape and (len(w) == 0 or np.abs(a - w).max() <= 2e-5 * max(1.0, float(np.abs(w).max())))
        # every point lies on the grid, inside the un-padded image (resizer.filter_points), off the border by b grid steps
        assert all((ws[-1][:, d] % rcfg.grid[d] == 0).all() for d in range(nd))

    # ---- the mirror's TILED sparse prediction (its own tile iterator; csbdeep's is absent, the reference's tiled branch cannot run here)
    # against the reference's UNTILED candidates: same points, same values -- tile offsets, the per-tile border rule (base.py:583-584: the
    # border of b pixels only where a tile touches the image border), the grid and the crop of the padded part
    nt = tuple(1 if a == "C" else (2 if img.shape[i] >= 24 else 1) for i, a in enumerate(axes))
    if int(np.prod(nt)) > 1:
        ws = ref.predict_sparse(img, prob_thresh=thr, axes=axes, b=2)
        gs = m.predict_sparse(img, prob_thresh=thr, axes=axes, b=2, n_tiles=nt)
        order_w, order_g = np.lexsort(ws[-1].T[::-1]), np.lexsort(np.asarray(gs[-1]).T[::-1])
        assert np.array_equal(np.asarray(gs[-1])[order_g], ws[-1][order_w]), (nt, len(gs[-1]), len(ws[-1]))
        for a, w in zip(gs[:-1], ws[:-1]):
            assert np.abs(np.asarray(a)[order_g] - w[order_w]).max() <= 2e-5 * max(1.0, float(np.abs(w).max()))

    # ---- refusals of _predict_setup (base.py:373-391)
    for bad in (dict(n_tiles=(1,) * (img.ndim + 1)), dict(n_tiles=(0,) * img.ndim), dict(n_tiles=(1.5,) + (1,) * (img.ndim - 1))):
        with pytest.raises(ValueError):
            ref.predict(img, axes=axes, **bad)
        with pytest.raises(ValueError):
            m.predict(img, axes=axes, **bad)


@pytest.mark.parametrize("nd,kw", [(2, dict(S2)), (2, dict(S2, grid=(2, 2), unet_n_depth=2)), (2, dict(S2, unet_n_depth=1, unet_kernel_size=(5, 5))),
                                   (2, dict(S2, grid=(4, 2), unet_n_depth=1)), (2, dict(S2, grid=(1, 4), unet_n_depth=1)),
                                   (3, dict(S3, unet_n_depth=1)), (3, dict(S3, unet_n_depth=1, grid=(1, 2, 2))),
                                   (3, dict(rays=8, backbone="resnet", resnet_n_filter_base=4, net_conv_after_resnet=8, resnet_n_blocks=2)),
                                   (3, dict(rays=8, backbone="resnet", resnet_n_filter_base=2, net_conv_after_resnet=4, resnet_n_blocks=2, grid=(1, 2, 2)))])
def test_tile_overlap_covers_the_receptive_field_the_reference_measures(nd, kw, ref_rays):
    """the reference measures the receptive field of the built network by an impulse response (base.py:1068-1098) and tiles with that
    overlap; the mirror uses the analytic radius of the layer stack (an upper bound: DESIGN.md section 4) -- it must never be smaller, and
    for the default kernel it should not be wasteful either (within one grid step per pooling level of the measured one)"""
    R2, R3 = _ref_configs(ref_rays)
    rcfg = (R2 if nd == 2 else R3)(**kw)
    ref, graph = reference_model(nd, rcfg)
    m = mirror_model(nd, kw, graph)
    axes = "YX" if nd == 2 else "ZYX"
    measured = ref._axes_tile_overlap(axes)
    mine = m._axes_tile_overlap(axes)
    assert ref._axes_div_by(axes + "C") == m._axes_div_by(axes + "C")
    assert all(a >= b for a, b in zip(mine, measured)), (mine, measured)
    slack = 2 ** (getattr(rcfg, "unet_n_depth", 2) + 1) * max(rcfg.grid)
    assert all(a - b <= slack for a, b in zip(mine, measured)), (mine, measured)
    assert m._axes_tile_overlap(axes + "C")[-1] == 0 == ref._axes_tile_overlap(axes + "C")[-1]
