"""CPU: the Python side of the hand-written convolutions (stardist_amd/models/unet.py::_hand_conv) without a GPU: which layers it
takes, and that every call it would make matches the C ABI's signature (argument count and ctypes conversions) with the right channel
counts, strides, up-sampling masks and kernel depth.  The device call itself is replaced by a recorder."""
import ctypes
import inspect

import numpy as np
import pytest


@pytest.fixture()
def recorder(monkeypatch):
    import torch  # noqa: F401
    from stardist_amd.lib import _native as N
    from stardist_amd.models import unet as U
    calls = []

    def fake_dcall(t, name, *args):
        fn = getattr(N.lib(), name)
        assert len(args) + 1 == len(fn.argtypes), (name, len(args) + 1, len(fn.argtypes))
        for a, at in zip(args, fn.argtypes):
            at.from_param(a)                      # raises if the Python value does not convert to the declared C type
        calls.append((name, [None if a is None else (a.value if isinstance(a, ctypes.c_void_p) else a) for a in args]))
    monkeypatch.setattr(N, "dcall", fake_dcall)
    # the wrapper insists on CUDA tensors; run its logic on CPU tensors by dropping exactly that check
    src = inspect.getsource(U._hand_conv)
    gsrc = inspect.getsource(U._general_conv)
    assert "t.is_cuda and " in src and "x.is_cuda and " in gsrc
    ns = dict(U.__dict__)

    class _Flag(object):
        def data_ptr(self):
            return 12345
    ns["range_flag"] = lambda device: _Flag()
    ns["_flag_ptr"] = lambda conv, device: 12345          # (the layer's word of the model's range-flag tensor)
    exec(gsrc.replace("x.is_cuda and ", ""), ns)
    exec(src.replace("t.is_cuda and ", ""), ns)
    return ns["_hand_conv"], calls


def _t(shape, cl):
    import torch
    return torch.randn(shape).contiguous(memory_format=cl)


def test_dispatch_2d_and_3d(recorder, monkeypatch):
    import torch
    hand, calls = recorder
    monkeypatch.setenv("STARDIST_AMD_CONV", "hand")                       # the exact-f32 kernel's entry point (same arguments as the split one)
    cl2, cl3 = torch.channels_last, torch.channels_last_3d
    with torch.no_grad():
        y = hand(torch.nn.Conv2d(64, 32, 3, padding=1), [(_t((1, 32, 8, 12), cl2), (1, 1)), (_t((1, 32, 16, 24), cl2), 0)], 1)
        assert tuple(y.shape) == (1, 32, 16, 24) and y.is_contiguous(memory_format=cl2)
        name, a = calls[-1]
        assert name == "sd_conv3_res_ndhwc_device"
        #      c0  stride0 up0        c1  stride1 up1   D  H   W   kz            res, res_stride     c_out act
        assert a[1:4] == [32, 32, 3] and a[5:8] == [32, 32, 0] and a[8:12] == [1, 16, 24, 1] and a[14:16] == [None, 0] and a[16:18] == [32, 1]
        y = hand(torch.nn.Conv3d(64, 64, 3, padding=1), [(_t((1, 32, 4, 8, 8), cl3), (1, 1, 1)), (_t((1, 32, 8, 16, 16), cl3), 0)], 0)
        assert tuple(y.shape) == (1, 64, 8, 16, 16) and y.is_contiguous(memory_format=cl3)
        name, a = calls[-1]
        assert a[1:4] == [32, 32, 7] and a[8:12] == [8, 16, 16, 3] and a[16:18] == [64, 0]
        hand(torch.nn.Conv3d(64, 32, 3, padding=1), [(_t((1, 32, 8, 8, 8), cl3), (0, 1, 1)), (_t((1, 32, 8, 16, 16), cl3), 0)], 1)
        assert calls[-1][1][3] == 3                                   # z not up-sampled: mask = x | y
        y = hand(torch.nn.Conv3d(1, 32, 3, padding=1), [(_t((1, 1, 6, 8, 10), cl3), 0)], 1)
        assert calls[-1][1][1:4] == [1, 1, 0] and calls[-1][1][4] is None and tuple(y.shape) == (1, 32, 6, 8, 10)
        # a source in the default layout is converted, not rejected
        y = hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(torch.randn(1, 32, 8, 8), 0)], 1)
        assert y is not None
        # residual (resnet_block's Add + Activation in the epilogue)
        r = _t((1, 64, 8, 16, 16), cl3)
        y = hand(torch.nn.Conv3d(64, 64, 3, padding=1), [(_t((1, 64, 8, 16, 16), cl3), 0)], 1, res=r)
        assert calls[-1][0] == "sd_conv3_res_ndhwc_device" and calls[-1][1][14:18] == [r.data_ptr(), 64, 64, 1]


def test_dispatch_general_kernel(recorder, monkeypatch):
    """layers that are not stride-1 3x3 over 32-channel chunks go to sd_convg_ndhwc_device: the ResNet stem / strided convolutions /
    1x1 projection (model3d.py:400-447, TensorFlow 'same' padding for strided layers), 3-channel first layers, narrow heads"""
    import torch
    from stardist_amd.models.unet import tf_same_pad_before
    hand, calls = recorder
    monkeypatch.delenv("STARDIST_AMD_CONV", raising=False)
    cl2, cl3 = torch.channels_last, torch.channels_last_3d
    assert [tf_same_pad_before(n, 3, 2) for n in (8, 9)] == [0, 1] and tf_same_pad_before(8, 7, 1) == 3 and tf_same_pad_before(9, 1, 2) == 0
    with torch.no_grad():
        y = hand(torch.nn.Conv3d(1, 32, 7, padding=3), [(_t((1, 1, 6, 9, 10), cl3), 0)], 0)
        name, a = calls[-1]
        #                                       c_in xs  D  H  W   k        s        p        Do Ho Wo
        assert name == "sd_convg_ndhwc_device" and a[1:18] == [1, 1, 6, 9, 10, 7, 7, 7, 1, 1, 1, 3, 3, 3, 6, 9, 10] and tuple(y.shape) == (1, 32, 6, 9, 10)
        assert a[20:26] == [None, 32, 32, 0, y.data_ptr(), 32]
        y = hand(torch.nn.Conv3d(32, 64, 3, stride=(1, 2, 2), padding=0), [(_t((1, 32, 6, 9, 10), cl3), 0)], 1, tf_same=True)
        a = calls[-1][1]
        assert a[6:18] == [3, 3, 3, 1, 2, 2, 1, 1, 0, 6, 5, 5] and tuple(y.shape) == (1, 64, 6, 5, 5) and a[23] == 1
        y = hand(torch.nn.Conv3d(32, 64, 1, stride=(1, 2, 2)), [(_t((1, 32, 6, 9, 10), cl3), 0)], 0, tf_same=True)
        assert calls[-1][1][6:18] == [1, 1, 1, 1, 2, 2, 0, 0, 0, 6, 5, 5]
        y = hand(torch.nn.Conv2d(3, 32, 3, padding=1), [(_t((1, 3, 8, 12), cl2), 0)], 1)
        assert calls[-1][1][1:18] == [3, 3, 1, 8, 12, 1, 3, 3, 1, 1, 1, 0, 1, 1, 1, 8, 12]
        y = hand(torch.nn.Conv2d(128, 5, 1), [(_t((1, 128, 8, 12), cl2), 0)], 0)
        assert calls[-1][0] == "sd_convg_ndhwc_device" and tuple(y.shape) == (1, 5, 8, 12)
        # batch-norm between convolution and activation: folded into kernel and bias (same entry point)
        bn = torch.nn.BatchNorm2d(32).eval()
        n0 = len(calls)
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(_t((1, 32, 8, 8), cl2), 0)], 1, bn=bn) is not None and len(calls) == n0 + 1
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(_t((1, 32, 8, 8), cl2), 0)], 1, bn=torch.nn.BatchNorm2d(32).train()) is None


def test_bn_fold_matches_module():
    import torch
    from stardist_amd.models.unet import _bn_fold
    torch.manual_seed(0)
    conv, bn = torch.nn.Conv2d(8, 16, 3, padding=1).double(), torch.nn.BatchNorm2d(16, eps=1e-3).double().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-1, 1); bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
        w, b = _bn_fold(conv, bn)
        x = torch.randn(1, 8, 10, 10, dtype=torch.float64)
        want = bn(conv(x))
        got = torch.nn.functional.conv2d(x, torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)
    assert float((want - got).abs().max()) < 1e-5


def test_dispatch_rejections_and_modes(recorder, monkeypatch):
    import torch
    hand, calls = recorder
    cl2 = torch.channels_last
    x32, x48 = _t((1, 32, 8, 8), cl2), _t((1, 48, 8, 8), cl2)
    with torch.no_grad():
        monkeypatch.delenv("STARDIST_AMD_CONV", raising=False)
        assert hand(torch.nn.Conv2d(48, 32, 3, padding=1), [(x48, 0)], 1) is not None and calls[-1][0] == "sd_convg_ndhwc_device"   # small-channel form
        assert hand(torch.nn.Conv2d(1000, 32, 3, padding=1), [(_t((1, 1000, 8, 8), cl2), 0)], 1) is None   # neither 32-chunks nor small
        for conv in (torch.nn.Conv2d(32, 32, 5, padding=2), torch.nn.Conv2d(32, 32, 3, padding=1, stride=2), torch.nn.Conv2d(32, 40, 3, padding=1)):
            assert hand(conv, [(x32, 0)], 1) is not None and calls[-1][0] == "sd_convg_ndhwc_device"   # 5x5 / strided / c_out 40: general kernel
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1, dilation=2), [(x32, 0)], 1) is None     # dilated
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(x32, 0)], -1) is None                # activation it does not fuse
        assert hand(torch.nn.Conv2d(64, 32, 3, padding=1), [(x32, 1), (_t((1, 32, 8, 8), cl2), 0)], 1) is None   # shapes do not match after x2
        for mode, entry in (("bf16x6", "sd_conv3_bf16x6_res_ndhwc_device"), (None, "sd_conv3_f16x3_res_ndhwc_device"), ("hand", "sd_conv3_res_ndhwc_device"),
                            ("f32", "sd_conv3_res_ndhwc_device"), ("f16x3", "sd_conv3_f16x3_res_ndhwc_device"),
                            ("miopen", "sd_conv3_f16x3_res_ndhwc_device")):      # default = the split-fp16 kernel; there is no library mode
            if mode is None:
                monkeypatch.delenv("STARDIST_AMD_CONV", raising=False)
            else:
                monkeypatch.setenv("STARDIST_AMD_CONV", mode)
            assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(x32, 0)], 1) is not None and calls[-1][0] == entry, mode
        monkeypatch.setenv("STARDIST_AMD_CONV", "bf16x6")
        assert hand(torch.nn.Conv2d(1, 32, 3, padding=1), [(_t((1, 1, 8, 8), cl2), 0)], 1) is not None and calls[-1][0] == "sd_conv3_res_ndhwc_device"
        # the split-fp16 form: the range-flag pointer rides in front of the stream; weights beyond the fp16 range -> the bf16x6 form
        monkeypatch.delenv("STARDIST_AMD_CONV", raising=False)
        conv = torch.nn.Conv2d(32, 32, 3, padding=1)
        assert hand(conv, [(x32, 0)], 1) is not None and calls[-1][0] == "sd_conv3_f16x3_res_ndhwc_device" and calls[-1][1][-1] == 12345
        conv.weight[0, 0, 0, 0] = 1e5
        assert hand(conv, [(x32, 0)], 1) is not None and calls[-1][0] == "sd_conv3_bf16x6_res_ndhwc_device"
        from stardist_amd.models import unet as U
        with U.force_conv_mode("hand"):
            assert U.conv_mode() == "hand"
        assert U.conv_mode() == "f16x3"
    with torch.enable_grad():
        monkeypatch.delenv("STARDIST_AMD_CONV", raising=False)
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(x32, 0)], 1) is None                 # training: plain modules


def test_split_weight_packing_roundtrip():
    """the split-bf16 packer of the C ABI: hi + mid + lo of every packed weight reproduces the f32 weight to <= 2^-24 relative, at the
    position the kernel reads it from (conv3x3_layout.h: [group][unit][dy][dx][block][plane][h][cout][8])"""
    from stardist_amd.lib import _native as N
    l = N.lib()
    rs = np.random.RandomState(1)
    for ci, co, kz in ((32, 32, 1), (64, 64, 3)):
        w = rs.randn(*((co, ci) + ((3, 3, 3) if kz == 3 else (3, 3)))).astype(np.float32)
        n = l.sd_conv3_bf16x6_packed_floats(ci, co, kz)
        assert n == co * ci * 9 * kz * 3 * 2 // 4 + 4
        out = np.empty(n, np.float32)
        N.check(l.sd_conv3_bf16x6_pack_weights_host(N.ptr(w), ci, co, kz, N.ptr(out)))
        assert not out[-4:].any()
        u16 = out[:-4].view(np.uint16).reshape(co // 32, (ci // 32) * kz, 3, 3, 2, 3, 2, 32, 8)   # g, unit, dy, dx, block, plane, h, cout, j
        planes = (u16.astype(np.uint32) << 16).view(np.float32)
        rec = planes.sum(axis=5, dtype=np.float64)                                               # g, unit, dy, dx, block, h, cout, j
        w6 = w.reshape(co // 32, 32, ci // 32, 2, 2, 8, kz, 3, 3)                                   # g, cout, chunk, block, h, j, z, dy, dx
        want = np.transpose(w6, (0, 2, 6, 7, 8, 3, 4, 1, 5)).reshape(rec.shape)                    # g, (chunk, z), dy, dx, block, h, cout, j
        assert np.abs(rec - want).max() <= np.abs(want).max() * 2.0 ** -23
    assert l.sd_conv3_bf16x6_packed_floats(1, 32, 1) == -1


def test_split_fp16_weight_packing_roundtrip():
    """the split-fp16 packer of the C ABI: hi + lo' * 2^-11 of every packed weight reproduces the f32 weight to <= 2^-21 relative at the
    position the kernel reads it from (conv3x3_layout.h: [group][unit][dy][dx][block][plane][h][cout][8]); hi is numpy's float16(w)
    (round to nearest even, subnormals, overflow to infinity); weights beyond the fp16 range are reported with -2"""
    from stardist_amd.lib import _native as N
    l = N.lib()
    rs = np.random.RandomState(2)
    for ci, co, kz in ((32, 32, 1), (64, 64, 3), (512, 32, 1)):
        w = (rs.randn(*((co, ci) + ((3, 3, 3) if kz == 3 else (3, 3)))) * np.exp(rs.uniform(-12, 3, 1))).astype(np.float32)
        w.reshape(-1)[:7] = [0.0, 6.1e-5, 5.9e-8, 3e-8, 65504.0, -1.0, 2.0 ** -14]
        n = l.sd_conv3_f16x3_packed_floats(ci, co, kz)
        assert n == co * ci * 9 * kz * 2 * 2 // 4 + 4
        out = np.empty(n, np.float32)
        N.check(l.sd_conv3_f16x3_pack_weights_host(N.ptr(w), ci, co, kz, N.ptr(out)))
        assert not out[-4:].any()
        f16 = out[:-4].view(np.float16).reshape(co // 32, (ci // 32) * kz, 3, 3, 2, 2, 2, 32, 8)   # g, unit, dy, dx, block, plane, h, cout, j
        w6 = w.reshape(co // 32, 32, ci // 32, 2, 2, 8, kz, 3, 3)                                   # g, cout, chunk, block, h, j, z, dy, dx
        want = np.transpose(w6, (0, 2, 6, 7, 8, 3, 4, 1, 5)).reshape(f16[:, :, :, :, :, 0].shape)   # g, (chunk, z), dy, dx, block, h, cout, j
        hi, lo = f16[:, :, :, :, :, 0], f16[:, :, :, :, :, 1]
        assert np.array_equal(hi.view(np.uint16), want.astype(np.float16).view(np.uint16))
        rem = (want - hi.astype(np.float32)) * np.float32(2048.0)
        assert np.array_equal(lo.view(np.uint16), rem.astype(np.float16).view(np.uint16))
        rec = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
        big = np.abs(want) >= 2.0 ** -14
        assert (np.abs(rec - want)[big] <= np.abs(want)[big] * 2.0 ** -21).all()
        assert np.abs(rec - want).max() <= max(np.abs(want).max() * 2.0 ** -21, 2.0 ** -35)
    w = np.zeros((32, 32, 3, 3), np.float32); w[3, 5, 1, 1] = 70000.0
    out = np.empty(l.sd_conv3_f16x3_packed_floats(32, 32, 1), np.float32)
    assert l.sd_conv3_f16x3_pack_weights_host(N.ptr(w), 32, 32, 1, N.ptr(out)) == -2 and b"fp16 range" in l.sd_last_error()
    assert l.sd_conv3_f16x3_packed_floats(1, 32, 1) == -1 and l.sd_conv3_f16x3_packed_floats(544, 32, 1) == -1
