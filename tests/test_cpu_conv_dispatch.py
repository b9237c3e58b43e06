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
    assert "t.is_cuda and " in src
    ns = dict(U.__dict__)
    exec(src.replace("t.is_cuda and ", ""), ns)
    return ns["_hand_conv"], calls


def _t(shape, cl):
    import torch
    return torch.randn(shape).contiguous(memory_format=cl)


def test_dispatch_2d_and_3d(recorder, monkeypatch):
    import torch
    hand, calls = recorder
    monkeypatch.delenv("STARDIST_AMD_CONV", raising=False)
    cl2, cl3 = torch.channels_last, torch.channels_last_3d
    with torch.no_grad():
        y = hand(torch.nn.Conv2d(64, 32, 3, padding=1), [(_t((1, 32, 8, 12), cl2), (1, 1)), (_t((1, 32, 16, 24), cl2), 0)], 1)
        assert tuple(y.shape) == (1, 32, 16, 24) and y.is_contiguous(memory_format=cl2)
        name, a = calls[-1]
        assert name == "sd_conv3_ndhwc_device"
        #      c0  stride0 up0        c1  stride1 up1   D  H   W   kz                 c_out act
        assert a[1:4] == [32, 32, 3] and a[5:8] == [32, 32, 0] and a[8:12] == [1, 16, 24, 1] and a[14:16] == [32, 1]
        y = hand(torch.nn.Conv3d(64, 64, 3, padding=1), [(_t((1, 32, 4, 8, 8), cl3), (1, 1, 1)), (_t((1, 32, 8, 16, 16), cl3), 0)], 0)
        assert tuple(y.shape) == (1, 64, 8, 16, 16) and y.is_contiguous(memory_format=cl3)
        name, a = calls[-1]
        assert a[1:4] == [32, 32, 7] and a[8:12] == [8, 16, 16, 3] and a[14:16] == [64, 0]
        hand(torch.nn.Conv3d(64, 32, 3, padding=1), [(_t((1, 32, 8, 8, 8), cl3), (0, 1, 1)), (_t((1, 32, 8, 16, 16), cl3), 0)], 1)
        assert calls[-1][1][3] == 3                                   # z not up-sampled: mask = x | y
        y = hand(torch.nn.Conv3d(1, 32, 3, padding=1), [(_t((1, 1, 6, 8, 10), cl3), 0)], 1)
        assert calls[-1][1][1:4] == [1, 1, 0] and calls[-1][1][4] is None and tuple(y.shape) == (1, 32, 6, 8, 10)
        # a source in the default layout is converted, not rejected
        y = hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(torch.randn(1, 32, 8, 8), 0)], 1)
        assert y is not None


def test_dispatch_rejections_and_modes(recorder, monkeypatch):
    import torch
    hand, calls = recorder
    cl2 = torch.channels_last
    x32, x48 = _t((1, 32, 8, 8), cl2), _t((1, 48, 8, 8), cl2)
    with torch.no_grad():
        monkeypatch.delenv("STARDIST_AMD_CONV", raising=False)
        assert hand(torch.nn.Conv2d(48, 32, 3, padding=1), [(x48, 0)], 1) is None                 # not a multiple of 32
        assert hand(torch.nn.Conv2d(32, 32, 5, padding=2), [(x32, 0)], 1) is None                 # 5x5
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1, stride=2), [(x32, 0)], 1) is None       # strided
        assert hand(torch.nn.Conv2d(32, 40, 3, padding=1), [(x32, 0)], 1) is None                 # c_out not a multiple of 32
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(x32, 0)], -1) is None                # activation it does not fuse
        assert hand(torch.nn.Conv2d(64, 32, 3, padding=1), [(x32, 1), (_t((1, 32, 8, 8), cl2), 0)], 1) is None   # shapes do not match after x2
        n0 = len(calls)
        monkeypatch.setenv("STARDIST_AMD_CONV", "miopen")
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(x32, 0)], 1) is None and len(calls) == n0
        monkeypatch.setenv("STARDIST_AMD_CONV", "bf16x6")
        assert hand(torch.nn.Conv2d(32, 32, 3, padding=1), [(x32, 0)], 1) is not None and calls[-1][0] == "sd_conv3_bf16x6_ndhwc_device"
        assert hand(torch.nn.Conv2d(1, 32, 3, padding=1), [(_t((1, 1, 8, 8), cl2), 0)], 1) is not None and calls[-1][0] == "sd_conv3_ndhwc_device"
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
