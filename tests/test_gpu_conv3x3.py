"""GPU: the hand-written 3x3 / 3x3x3 convolution (stardist_amd/csrc/conv3x3.hip, f32 matrix cores) through the C ABI against the
same layer in float64 (torch on the CPU): Conv2D / Conv3D(kernel 3, 'same') + bias + activation of the reference's U-Net (csbdeep
unet_block as built by stardist/models/model2d.py:310-349, model3d.py:360-399), with UpSampling + Concatenate folded in for the
first layer of an up level.
Tolerance 1e-5 relative to the output scale (BASELINE north_star: 1e-5 on probabilities / distances)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(srcs, conv, kind):
    import torch
    import torch.nn.functional as F
    nd = conv.weight.dim() - 2
    xs = []
    for t, up in srcs:
        t = t.detach().cpu().double()
        up = up if isinstance(up, tuple) else (up,) * nd
        if any(up):
            t = F.interpolate(t, scale_factor=tuple(2.0 if u else 1.0 for u in up), mode="nearest")
        xs.append(t)
    f = F.conv2d if nd == 2 else F.conv3d
    y = f(torch.cat(xs, 1), conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double(), padding=1)
    return torch.relu(y) if kind == 1 else y


CASES = [  # (spatial shape, [(channels, up)], c_out)
    ((64, 96), [(32, 0)], 32), ((50, 70), [(32, 0)], 64), ((8, 32), [(32, 0)], 128), ((40, 33), [(64, 0)], 32), ((72, 64), [(64, 0)], 64),
    ((48, 80), [(32, 1), (32, 0)], 32), ((30, 46), [(1, 0)], 32), ((512, 512), [(32, 0)], 32), ((512, 384), [(32, 1), (32, 0)], 32),
    ((256, 512), [(32, 0)], 128), ((32, 64), [(128, 1), (128, 0)], 128), ((24, 40), [(128, 0)], 256), ((16, 32), [(256, 0)], 128),
    ((12, 20, 36), [(32, 0)], 32), ((6, 16, 40), [(1, 0)], 32), ((8, 24, 64), [(32, 1), (32, 0)], 32), ((10, 16, 32), [(64, 0)], 64),
    ((4, 8, 32), [(64, 1), (64, 0)], 64), ((8, 16, 32), [(32, (0, 1, 1)), (32, 0)], 32), ((32, 64, 64), [(32, 0)], 128),
    ((6, 10, 34), [(64, 0)], 128), ((16, 32), [(256, 1), (256, 0)], 256), ((4, 8, 32), [(256, 1), (256, 0)], 256),
]


@pytest.mark.parametrize("shape,chans,c_out", CASES)
@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("mode", ["hand", "bf16x6", "f16x3", "f16x3-1wg"])
def test_conv3_matches_float64(shape, chans, c_out, kind, mode, monkeypatch):
    """mode 'hand': the exact f32-MFMA kernel; 'bf16x6': six bf16 MFMAs per product; 'f16x3': the default split-fp16 kernel (three fp16
    MFMAs per product, two workgroups per CU); 'f16x3-1wg': the same kernel's one-workgroup-per-CU instance"""
    import torch
    if mode != "hand" and chans[0][0] == 1:
        pytest.skip("the one-channel first layer has no split form")
    if mode == "f16x3-1wg":
        from stardist_amd.lib import _native as N
        N.check(N.lib().sd_set_option(b"conv_f16_workgroups_per_cu", 1))
        mode = "f16x3"
        request_restore = True
    else:
        request_restore = False
    monkeypatch.setenv("STARDIST_AMD_CONV", mode)
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    nd = len(shape)
    g = torch.Generator().manual_seed(int(np.prod(shape)) * 131 + c_out + kind)
    cin = sum(c for c, _ in chans)
    conv = (torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d)(cin, c_out, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * float(np.sqrt(2.0 / (3 ** nd * cin))))
        conv.bias.copy_(torch.randn(c_out, generator=g) * 0.5)
    conv = conv.to(dev)
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    srcs = []
    for c, up in chans:
        upt = up if isinstance(up, tuple) else (up,) * nd
        t = torch.randn((1, c) + tuple(s >> u for s, u in zip(shape, upt)), generator=g).to(dev).contiguous(memory_format=cl)
        srcs.append((t, upt))
    try:
        with torch.no_grad():
            if mode == "f16x3":
                U.range_flag(dev).zero_()
            y = U._hand_conv(conv, srcs, kind)
            assert y is not None, "layer not taken by the hand-written kernel"
            y2 = U._hand_conv(conv, srcs, kind)
        torch.cuda.synchronize()
    finally:
        if request_restore:
            from stardist_amd.lib import _native as N
            N.check(N.lib().sd_set_option(b"conv_f16_workgroups_per_cu", 2))
    if mode == "f16x3":
        assert int(U.range_flag(dev).item()) == 0
    assert tuple(y.shape) == (1, c_out) + tuple(shape) and y.is_contiguous(memory_format=cl)
    assert torch.equal(y, y2), "not repeatable"
    ref = _ref(srcs, conv, kind)
    err = float((y.cpu().double() - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    print("conv3 %s %s %s c_out %d kind %d: max err / scale = %.3g" % (mode, shape, chans, c_out, kind, err / scale))
    assert err <= 1e-5 * scale, (err, scale)       # the same bar for the exact-f32 and the split kernels


def test_layout_conversion_and_refusals():
    import torch
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    with torch.no_grad():
        x32 = torch.randn(1, 32, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
        assert U._hand_conv(torch.nn.Conv2d(32, 32, 3, padding=1, dilation=2).to(dev), [(x32, 0)], 1) is None   # dilated: not covered
        assert U._hand_conv(torch.nn.Conv2d(32, 32, 3, padding=1).to(dev), [(x32, 0)], 2) is None               # activation it does not fuse
        # a source in the default (NCHW) layout is converted, not rejected
        conv = torch.nn.Conv2d(32, 32, 3, padding=1).to(dev)
        xc = x32.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)                                     # same values, NCHW-contiguous storage
        xc = xc.contiguous()
        assert torch.equal(U._hand_conv(conv, [(xc, 0)], 1), U._hand_conv(conv, [(x32, 0)], 1))


# ---- the general kernel (csrc/conv_general.hip) and the residual epilogue ------------------------------------------------------
GENERAL = [  # nd, c_in, c_out, kernel, stride, spatial, tf_same, residual
    (3, 1, 32, 7, (1, 1, 1), (20, 40, 70), False, False),        # ResNet stem (model3d.py:414)
    (3, 32, 64, 3, (1, 2, 2), (12, 33, 70), True, False),        # strided first convolution of a resnet_block, odd extent
    (3, 32, 64, 1, (1, 2, 2), (12, 33, 70), True, False),        # its 1x1x1 shortcut projection
    (3, 64, 64, 3, (2, 2, 2), (9, 20, 34), True, True),          # strided in z as well, with residual
    (3, 64, 64, 3, (1, 1, 1), (6, 18, 40), False, True),         # conv3x3.hip with the Add + Activation epilogue
    (2, 64, 64, 3, (1, 1), (40, 72), False, True),
    (2, 3, 32, 3, (1, 1), (90, 130), False, False),              # H&E first layer (model2d.py:310-316, n_channel_in = 3)
    (2, 128, 5, 1, (1, 1), (64, 96), False, False),              # prob_class head (n_classes + 1 channels)
    (2, 32, 32, 5, (1, 1), (48, 80), False, False),              # unet_kernel_size = (5, 5)
    (2, 128, 1, 1, (1, 1), (33, 47), False, False),
]


@pytest.mark.parametrize("nd,ci,co,k,stride,S,tf_same,with_res", GENERAL)
@pytest.mark.parametrize("kind", [0, 1])
def test_general_and_residual_layers_match_float64(nd, ci, co, k, stride, S, tf_same, with_res, kind):
    import torch
    import torch.nn.functional as F
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(ci * 7919 + co * 31 + k + kind)
    Conv = torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d
    conv = Conv(ci, co, k, stride=stride, padding=0 if tf_same else k // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * float(np.sqrt(2.0 / (k ** nd * ci))))
        conv.bias.copy_(torch.randn(co, generator=g) * 0.5)
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    x = torch.randn((1, ci) + tuple(S), generator=g)
    # float64 reference
    xd = x.double()
    if tf_same:
        pads = []
        for d in reversed(range(nd)):
            n, st = S[d], stride[d]
            tot = max(k - st, 0) if n % st == 0 else max(k - n % st, 0)
            pads += [tot // 2, tot - tot // 2]
        xd = F.pad(xd, pads)
    with torch.no_grad():
        ref = (F.conv2d if nd == 2 else F.conv3d)(xd, conv.weight.double(), conv.bias.double(), stride=stride, padding=0 if tf_same else k // 2)
    res = torch.randn(ref.shape, generator=g) if with_res else None
    if res is not None:
        ref = ref + res.double()
    if kind == 1:
        ref = torch.relu(ref)
    conv = conv.to(dev)
    xg = x.to(dev).contiguous(memory_format=cl)
    rg = res.to(dev).contiguous(memory_format=cl) if res is not None else None
    with torch.no_grad():
        y = U._hand_conv(conv, [(xg, 0)], kind, res=rg, tf_same=tf_same)
        assert y is not None, "layer not taken by a hand-written kernel"
        y2 = U._hand_conv(conv, [(xg, 0)], kind, res=rg, tf_same=tf_same)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape) and y.is_contiguous(memory_format=cl)
    assert torch.equal(y, y2), "not repeatable"
    err = float((y.cpu().double() - ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.abs().max())), err


def test_split_fp16_range_flag_and_fallback():
    """an activation beyond the fp16 range sets the device flag (the layer's output is then not valid); StarDistBase._net_forward
    re-evaluates with the bf16x6 form and pins it: the prediction equals the bf16x6 prediction"""
    import torch
    import warnings
    from stardist_amd.models import unet as U
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda:0")
    conv = torch.nn.Conv2d(32, 32, 3, padding=1).to(dev)
    x = torch.randn(1, 32, 40, 70, device=dev).contiguous(memory_format=torch.channels_last)
    flag = U.range_flag(dev)
    with torch.no_grad(), U.force_conv_mode("f16x3"):
        flag.zero_()
        U._hand_conv(conv, [(x, 0)], 1)
        assert int(flag.item()) == 0
        x[0, 7, 33, 69] = 7e4
        U._hand_conv(conv, [(x, 0)], 1)
        assert int(flag.item()) == 1
        flag.zero_()
        x[0, 7, 33, 69] = float("inf")
        U._hand_conv(conv, [(x, 0)], 1)
        assert int(flag.item()) == 1
        flag.zero_()
    img = np.random.RandomState(0).rand(96, 128).astype(np.float32)
    img[40, 50] = 3e7                     # a hot pixel the normaliser is told to keep
    m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = m.predict(img)
    forced = m.__dict__.get("_fp16_range_layers") or []
    if forced:                            # the flag tripped (depends on the seeded weights' gain): the layers are named, the result is finite
        assert any("fp16 range" in str(x.message) for x in w)
        assert all(m.net.get_submodule(n).__dict__.get("_sd_force_form") == "bf16x6" for n in forced)
    assert all(np.isfinite(a).all() for a in got)
    again = m.predict(img)                # the pinned layers stay pinned: no second warning, same result
    assert all(np.array_equal(a, b) for a, b in zip(got, again))


def test_fp16_range_fallback_pins_only_the_offending_layer():
    """an activation of ~1e5 produced by ONE mid-level layer: exactly the layer that reads it moves to the bf16x6 form (one warning that
    names it), every other layer stays on the split-fp16 kernel, and the prediction is within 1e-5 of a float64 evaluation"""
    import torch
    import warnings
    from stardist_amd.models import unet as U
    from stardist_amd.models import Config2D, StarDist2D
    from oracle import synth
    dev = torch.device("cuda:0")
    m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    blk = m.net.backbone.down[1]                                   # second level: [ConvAct 32 -> 64, ConvAct 64 -> 64]
    prod, cons = blk[0][0], blk[1][0]
    gain = 3.0e4
    with torch.no_grad():
        prod.weight.mul_(gain); prod.bias.mul_(gain)               # its (ReLU) outputs reach ~1e5 ...
        cons.weight.mul_(1.0 / gain)                               # ... and the layer that reads them brings the scale back
    img = synth.s2d_nuclei_image(192, 256, seed=3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        prob, dist = m.predict(img)
    forced = m.__dict__.get("_fp16_range_layers")
    assert forced == ["backbone.down.1.1.0"], forced
    assert sum("fp16 range" in str(x.message) for x in w) == 1
    convs = [mod for mod in m.net.modules() if isinstance(mod, torch.nn.Conv2d)]
    assert [c for c in convs if c.__dict__.get("_sd_force_form") == "bf16x6"] == [cons]
    mc = StarDist2D(Config2D(n_rays=32), basedir=None, device="cpu", seed=0)
    mc.net.load_state_dict({k: v.cpu() for k, v in m.net.state_dict().items()})
    with torch.no_grad():
        p64, d64 = mc.net.double()(torch.from_numpy(img)[None, None].double())
    p64 = p64[0, 0].numpy(); d64 = np.moveaxis(d64[0].numpy(), 0, -1)
    assert float(np.abs(prob - p64).max()) <= 1e-5
    assert float((np.abs(dist - np.maximum(d64, 1e-3)) / np.maximum(1.0, np.abs(d64))).max()) <= 1e-5
    # two models on one device keep separate flag words: a clean model next to the pinned one is not disturbed
    m2 = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    m2.predict(img)
    assert not m2.__dict__.get("_fp16_range_layers")


def test_batch_norm_layer_folded_matches_float64():
    """csbdeep conv_block with batch_norm=True: Conv -> BatchNormalization (moving statistics) -> Activation, folded into the
    hand-written layer's kernel and bias"""
    import torch
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    blk = U._conv(2, 32, 64, 3, "relu", batch_norm=True)
    with torch.no_grad():
        bn = blk[1]
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.uniform_(-0.5, 0.5); bn.running_var.uniform_(0.5, 2.0)
    blk.eval()
    x = torch.randn(1, 32, 70, 90)
    with torch.no_grad():
        ref = blk.double()(x.double())
        blk = blk.float().to(dev)
        y = blk(x.to(dev).contiguous(memory_format=torch.channels_last))
    assert float((y.cpu().double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
