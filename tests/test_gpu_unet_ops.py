"""GPU: the fused bias+activation epilogue equals the framework's two element-wise passes (the epilogue itself is exact: one
add, one max -- checked bit for bit on a plain tensor below)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nd,act,cl", [(2, "relu", True), (2, None, True), (2, "relu", False), (3, "relu", True), (3, None, False)])
def test_conv_bias_act_fused_equals_plain(nd, act, cl):
    import torch
    import torch.nn as nn
    from stardist_amd.models.unet import _conv
    dev = torch.device("cuda:0")
    torch.manual_seed(nd * 7 + (1 if cl else 0))
    m = _conv(nd, 5, 12, 3, act).to(dev)
    with torch.no_grad():
        m[0].bias.normal_()
        x = torch.randn((1, 5, 33, 47) if nd == 2 else (1, 5, 9, 17, 21), device=dev)
        if cl:
            fmt = torch.channels_last if nd == 2 else torch.channels_last_3d
            x = x.contiguous(memory_format=fmt); m = m.to(memory_format=fmt)
        y_fused = m(x)
        y_plain = nn.Sequential.forward(m, x)
    assert y_fused.shape == y_plain.shape
    # hand-written layer (general kernel: 5 -> 12 channels) against the framework's convolution + bias + activation (test-side comparator)
    assert torch.allclose(y_fused, y_plain, rtol=1e-5, atol=1e-5)
    if act == "relu":
        assert float(y_fused.min()) >= 0.0


def test_bias_act_odd_channels_and_error():
    import ctypes
    import torch
    from stardist_amd.lib import _native as N
    dev = torch.device("cuda:0")
    x = torch.randn(1000, 7, device=dev); b = torch.randn(7, device=dev)
    ref = torch.relu(x + b)
    N.check(N.lib().sd_bias_act_device(N.tptr(x), N.tptr(b), 1000, 7, 1, 1, N.current_stream()))
    assert torch.equal(x, ref)
    assert N.lib().sd_bias_act_device(N.tptr(x), N.tptr(b), 1000, 7, 1, 5, N.current_stream()) != 0


@pytest.mark.parametrize("shape,pool", [((1, 32, 70, 96), (2, 2)), ((1, 64, 33, 47), (2, 2)), ((1, 32, 12, 40, 50), (2, 2, 2)),
                                        ((1, 32, 9, 40, 51), (1, 2, 2)), ((1, 128, 8, 6, 10), (2, 2, 2)), ((1, 4, 16, 16), (4, 2))])
def test_native_max_pool_equals_framework(shape, pool):
    """Keras MaxPooling 'valid' (floor of the extent): the native channels-last kernel against torch's max_pool on the same tensor"""
    import torch
    import torch.nn.functional as F
    from stardist_amd.models import unet as U
    nd = len(shape) - 2
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1)).cuda().contiguous(memory_format=cl)
    with torch.no_grad():
        y = U.max_pool(x, pool)
    want = (F.max_pool2d if nd == 2 else F.max_pool3d)(x.contiguous(), pool)
    assert tuple(y.shape) == tuple(want.shape) and y.is_contiguous(memory_format=cl)
    assert torch.equal(y, want)
