"""GPU: the fused network heads (stardist_amd/csrc/unet_ops.hip) through the C ABI against float64 torch arithmetic:
features epilogue + probability head (sd_bias_act_dot_device), distance head as fp32-MFMA GEMM on selected rows
(sd_head_rows_device) -- dense and sparse evaluation of the same row agree bit for bit -- and the two-source epilogue of
Concatenate+Conv (sd_add_bias_act_device).  Reference semantics: Conv 1x1 heads of model2d.py:338-343 / model3d.py:436-441."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _vp(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else None)


@pytest.mark.parametrize("C", [32, 64, 128, 256])
@pytest.mark.parametrize("act", [0, 1])
def test_bias_act_dot(C, act):
    import torch
    from stardist_amd.lib import _native as N
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + act)
    n = 10007
    x = torch.randn(n, C, generator=g).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    w = (torch.randn(C, generator=g) * 0.2).to(dev)
    wb = torch.randn(1, generator=g).to(dev)
    ref = x + bias
    if act:
        ref = torch.relu(ref)
    for inplace in (False, True):
        for sigm in (0, 1):
            src = x.clone()
            out = src if inplace else torch.full_like(x, float("nan"))
            dot = torch.full((n,), float("nan"), device=dev)
            N.dcall(src, "sd_bias_act_dot_device", _vp(src), _vp(out), _vp(bias), n, C, act, _vp(w), _vp(wb), sigm, _vp(dot))
            assert torch.equal(out, ref)
            d64 = ref.double() @ w.double() + wb.double()
            if sigm:
                d64 = torch.sigmoid(d64)
            err = float((dot.double() - d64).abs().max())
            assert err <= (2e-6 if sigm else 2e-5), err
    # without a head: plain epilogue
    src = x.clone()
    N.dcall(src, "sd_bias_act_dot_device", _vp(src), _vp(src), _vp(bias), n, C, act, None, None, 0, None)
    assert torch.equal(src, ref)
    with pytest.raises(RuntimeError):
        N.dcall(src, "sd_bias_act_dot_device", _vp(src), _vp(src), _vp(bias), n, 48, act, None, None, 0, None)


@pytest.mark.parametrize("C,R", [(128, 32), (128, 96), (32, 32), (64, 7), (64, 100), (256, 32)])
def test_head_rows_matches_float64_and_dense_equals_sparse(C, R):
    import torch
    from stardist_amd.lib import _native as N
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C * 1000 + R)
    n_pix = 5003
    feat = torch.relu(torch.randn(n_pix, C, generator=g)).to(dev)
    W = (torch.randn(R, C, generator=g) * 0.3).to(dev)
    b = torch.randn(R, generator=g).to(dev)
    ref = feat.double() @ W.double().t() + b.double()
    scale = (feat.double().abs() @ W.double().abs().t() + b.double().abs())          # sum |a b|: the float32 error scale

    def run(rows, clamp):
        n = n_pix if rows is None else rows.numel()
        out = torch.full((n, R), float("nan"), device=dev)
        N.dcall(feat, "sd_head_rows_device", _vp(feat), C, _vp(rows), n, _vp(W), _vp(b), R, float(clamp), _vp(out))
        return out
    dense = run(None, float("-inf"))
    assert float(((dense.double() - ref).abs() / scale).max()) <= 2e-6
    rows = torch.randperm(n_pix, generator=g)[:777].to(dev)
    rows[5] = rows[6]                                                               # duplicates are fine
    sparse = run(rows, float("-inf"))
    assert torch.equal(sparse, dense[rows])                                          # same fma chain whatever the batch
    for n in (1, 31, 32, 33, 64):                                                    # ragged tiles
        assert torch.equal(run(rows[:n].contiguous(), float("-inf")), dense[rows[:n]])
    clamped = run(rows, 1e-3)
    assert torch.equal(clamped, torch.clamp_min(dense[rows], 1e-3))
    # no bias
    out = torch.empty((n_pix, R), device=dev)
    N.dcall(feat, "sd_head_rows_device", _vp(feat), C, None, n_pix, _vp(W), None, R, float("-inf"), _vp(out))
    assert float(((out.double() - (ref - b.double())).abs() / scale).max()) <= 2e-6
    with pytest.raises(RuntimeError):
        N.dcall(feat, "sd_head_rows_device", _vp(feat), 130, None, 1, _vp(W), None, R, 0.0, _vp(out))


def test_add_bias_act():
    import torch
    from stardist_amd.lib import _native as N
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for shape, C, inner in (((4099, 32), 32, 1), ((3, 6, 1001), 6, 1001)):
        x = torch.randn(*shape, generator=g).to(dev); y = torch.randn(*shape, generator=g).to(dev)
        bias = torch.randn(C, generator=g).to(dev)
        n_outer = x.numel() // (C * inner)
        for act in (0, 1):
            t = x.clone()
            N.dcall(t, "sd_add_bias_act_device", _vp(t), _vp(y), _vp(bias), n_outer, C, inner, act)
            ref = (x + y) + (bias if inner == 1 else bias.view(1, C, 1))
            assert torch.equal(t, torch.relu(ref) if act else ref)


@pytest.mark.parametrize("dim", [2, 3])
def test_model_sparse_head_equals_dense_prediction(dim):
    """predict_sparse (distance head on the candidate rows only, also through the tiled path) returns exactly the masked dense
    prediction: the reference's predict_sparse contract, base.py:553-610"""
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    dev = torch.device("cuda:0")
    if dim == 2:
        img = synth.s2d_nuclei_image(384, 320, seed=2)
        m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
        bench.calibrate_heads(m, torch.from_numpy(img).to(dev), frac=0.05)
        tiles = (2, 2)
    else:
        img = synth.s3d_nuclei_image(64, seed=2)
        m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        bench.calibrate_heads(m, torch.from_numpy(img).to(dev), frac=0.02, radius=8.5, noise=0.03)
        tiles = (1, 2, 2)
    thr = float(m.thresholds.prob)
    # with the dense feature tensor (small inputs) and without it (round 6: the features layer keeps only the probability head's partial sums, the
    # features of the candidate pixels are evaluated afterwards by sd_conv3_f16x3_rows_device; the default from 4 GiB of features on)
    for lazy_min, n_tiles in ((4 << 30, None), (4 << 30, tiles), (0, None), (0, tiles)):
        m.net.lazy_features_min_bytes = lazy_min
        m.__dict__.pop("_graphs", None)
        prob, dist = m.predict(img, n_tiles=n_tiles)[:2]
        ps, ds, pts = m.predict_sparse(img, prob_thresh=thr, n_tiles=n_tiles)
        assert m._head_mode == ("sparse_lazy" if lazy_min == 0 else "sparse")
        mask = prob > thr
        b = 2
        inner = np.zeros_like(mask); inner[(slice(b, -b),) * dim] = True
        mask &= inner
        want_pts = np.stack(np.nonzero(mask), 1)
        if n_tiles is not None:                      # candidates come tile by tile (as in the reference): compare in C order
            order = np.lexsort(pts.T[::-1])
            ps, ds, pts = ps[order], ds[order], pts[order]
        assert np.array_equal(pts, want_pts)
        idx = tuple(want_pts.T)
        assert np.array_equal(ps, prob[idx])
        assert np.array_equal(ds, np.maximum(dist[idx], 1e-3))


@pytest.mark.parametrize("nd,shape,co", [(2, (37, 70), 128), (2, (64, 64), 32), (3, (5, 19, 33), 128), (3, (8, 16, 40), 64), (2, (40, 33), 256)])
def test_fused_probability_head_in_the_features_epilogue(nd, shape, co):
    """sd_conv3_f16x3_dot_ndhwc_device + sd_dot_combine_device (the probability head's first stage taken from the features layer's tile while
    it is in registers): the features are bit-identical to the plain layer's, the per-lane terms are the 4-channel dot products of the
    stored features (float64, <= 1e-5 of the scale), and the combined head equals sd_bias_act_dot_device on those features BIT FOR BIT
    (and sigmoid(features . w + b) to 2e-6) -- ragged tiles included"""
    import torch
    import torch.nn as nn
    from stardist_amd.lib import _native as N
    from stardist_amd.models import unet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(nd * 100 + co)
    Conv = nn.Conv2d if nd == 2 else nn.Conv3d
    conv = Conv(32, co, (3,) * nd, padding=1).to(dev)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) * 0.08).to(dev))
        conv.bias.copy_((torch.randn(co, generator=g) * 0.1).to(dev))
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    x = torch.randn((1, 32) + shape, generator=g).to(dev).contiguous(memory_format=cl)
    w = (torch.randn(co, generator=g) * 0.2).to(dev)
    wb = torch.randn(1, generator=g).to(dev)
    with torch.no_grad(), unet.force_conv_mode("f16x3"):
        plain = unet._hand_conv(conv, [(x, 0)], 1)
        holder = []
        feat = unet._hand_conv(conv, [(x, 0)], 1, dot=(w, holder))
    assert plain is not None and feat is not None and len(holder) == 1
    assert torch.equal(plain, feat)
    n_pix = int(np.prod(shape))
    part = holder[0]
    assert tuple(part.shape) == (n_pix, co // 4)
    f32 = feat.permute(*([0] + list(range(2, nd + 2)) + [1])).reshape(n_pix, co).contiguous()
    f = f32.double()
    ref_terms = (f * w.double()).reshape(n_pix, co // 4, 4).sum(-1)
    assert float((part.double() - ref_terms).abs().max()) <= 1e-5 * max(1.0, float(ref_terms.abs().max()))
    for sigm in (1, 0):
        prob = torch.full((n_pix,), float("nan"), device=dev)
        N.dcall(part, "sd_dot_combine_device", _vp(part), co // 32, n_pix, _vp(wb), sigm, _vp(prob))
        two_pass = torch.full((n_pix,), float("nan"), device=dev)
        N.dcall(f32, "sd_bias_act_dot_device", _vp(f32), None, None, n_pix, co, 0, _vp(w), _vp(wb), sigm, _vp(two_pass))
        assert torch.equal(prob, two_pass)
        ref = f @ w.double() + wb.double()
        if sigm:
            ref = torch.sigmoid(ref)
        assert float((prob.double() - ref).abs().max()) <= (2e-6 if sigm else 2e-5)


@pytest.mark.parametrize("nd,cin,cout,S", [(2, 32, 128, (70, 90)), (2, 64, 64, (33, 40)), (3, 32, 128, (9, 20, 37)), (3, 64, 32, (6, 10, 12)), (2, 32, 256, (20, 30))])
@pytest.mark.parametrize("split", [False, True])
def test_layer_on_selected_pixels_equals_the_dense_layer_bit_for_bit(nd, cin, cout, S, split):
    """sd_conv3_f16x3_rows_device (the sparse path's features: the layer evaluated on the candidate pixels only, round 6) == the rows of the
    dense split-fp16 layer, bit for bit -- border pixels (zero padding), a partial last wave, f32 and split16 sources, 1 / 2 / 4 / 8 groups of
    output channels; and the dense pass that only keeps the fused head's partial sums (d_out == NULL) gives the same probabilities"""
    import torch
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    Conv = torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d
    torch.manual_seed(nd * 100 + cin + cout)
    conv = Conv(cin, cout, 3, padding=1).to(dev)
    x = torch.randn((1, cin) + S, device=dev).contiguous(memory_format=cl)
    n_pix = int(np.prod(S))
    g = torch.Generator().manual_seed(5)
    rows = torch.cat([torch.tensor([0, 1, S[-1] - 1, S[-1], n_pix - 1, n_pix - S[-1]]), torch.randint(0, n_pix, (1000 + 7,), generator=g)]).to(dev)
    w = torch.randn(cout, device=dev)
    with torch.no_grad(), U.force_conv_mode("f16x3"):
        holder = []
        dense = U._hand_conv(conv, [(x, 0)], 1, dot=(w, holder))
        src = U.split16_pack(x) if split else x
        got = U.conv_rows(conv, src, 1, rows)
        h2 = []
        r = U._hand_conv(conv, [(src, 0)], 1, dot=(w, h2), no_store=True)
    want = dense[0].permute(*(list(range(1, nd + 1)) + [0])).reshape(n_pix, cout)[rows]
    assert torch.equal(got, want), float((got - want).abs().max())
    assert r is U.NO_STORE and len(h2) == 1 and torch.equal(h2[0], holder[0])
