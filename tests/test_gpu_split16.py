"""GPU: split16 activation tensors (include/stardist_hip.h "split16", round 6) -- the two fp16 terms (hi, lo') the split-fp16 convolution
multiplies with, made once by the PRODUCING layer instead of once per consumer workgroup and unit.  The claim is bit-identity: a layer gives the
same result whichever form carries its operands, a split16 output is exactly numpy's split of the f32 output, max-pooling commutes with
the split -- hence a network's outputs are the same bits with the form on or off (csbdeep unet_block as built by
stardist/models/model2d.py:310-349, model3d.py:360-399)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def np_split16(x_cl):
    """numpy statement of the form for a channels-last array (..., C), C % 32 == 0: float32 array of the same shape holding, per pixel
    and 32-channel chunk, 32 float16 hi terms then 32 float16 lo' terms (hi = fp16(x), lo' = fp16((x - hi) * 2^11), round to nearest even)"""
    x = np.ascontiguousarray(x_cl, np.float32)
    C = x.shape[-1]
    assert C % 32 == 0
    v = x.reshape(-1, C // 32, 32)
    with np.errstate(over="ignore", invalid="ignore"):
        hi = v.astype(np.float16)
        lo = ((v - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    both = np.concatenate([hi, lo], axis=-1)                     # (n, chunks, 64) float16
    return np.ascontiguousarray(both).view(np.float32).reshape(x.shape)


def np_unsplit16(s_cl):
    s = np.ascontiguousarray(s_cl, np.float32)
    C = s.shape[-1]
    h = s.reshape(-1, C // 32, 32).view(np.float16).reshape(-1, C // 32, 64)
    return (h[..., :32].astype(np.float32) + h[..., 32:].astype(np.float32) * np.float32(2.0 ** -11)).reshape(s.shape)


def _cl(t):
    """(1, C, *S) channels-last torch tensor -> numpy (*S, C)"""
    nd = t.dim() - 2
    return t[0].permute(*(list(range(1, nd + 1)) + [0])).contiguous().cpu().numpy()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _rand(shape, cl, dev, seed, scale=1.0, ties=False):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * scale
    # a spread of magnitudes: tiny values (subnormal hi terms), exact fp16 values (lo' = 0), zeros
    flat = x.view(-1)
    flat[::7] *= 1e-6
    flat[::11] = flat[::11].half().float()
    flat[::13] = 0.0
    if ties:
        # near-ties between neighbours along x: values one f32 step apart (same hi, adjacent lo'), and pairs straddling the midpoint of two
        # fp16 values (different hi, the same hi + lo' 2^-11)
        n = flat[1::2].numel()
        flat[1::2] = torch.nextafter(flat[0::2][:n], torch.full((n,), 10.0))
        h = flat[0::8].half().float()
        mid = h + (torch.nextafter(h.half(), torch.full_like(h, 10.0).half()).float() - h) * 0.5
        m = mid.numel()
        flat[0::8] = torch.nextafter(mid, torch.full((m,), -10.0))
        flat[1::8][:m] = torch.nextafter(mid, torch.full((m,), 10.0))[: flat[1::8].numel()]
    # (no -0.0: the form keeps the sign of a zero but not the difference between +0.0 and 0 < x < 2^-36, so max-pooling a window that
    #  holds -0.0 AND such a value cannot be told from one that holds -0.0 and +0.0 -- the one input class, never produced behind a
    #  ReLU, on which the pooled forms differ, in the sign of a zero)
    x[x == 0] = 0.0
    return x.to(dev).contiguous(memory_format=cl)


def test_pack_unpack_match_numpy():
    import torch
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    x = _rand((1, 96, 37, 53), torch.channels_last, dev, 1, 3.0)
    s = U.split16_pack(x)
    assert U.is_split16(s)
    want = np_split16(_cl(x))
    assert np.array_equal(_bits(_cl(s)), _bits(want))
    back = U.split16_unpack(s)
    assert np.array_equal(_bits(_cl(back)), _bits(np_unsplit16(want)))
    # the 22 bits: |x - (hi + lo' 2^-11)| <= 2^-22 |x| (+ the absolute floor of subnormal terms)
    err = np.abs(_cl(back).astype(np.float64) - _cl(x).astype(np.float64))
    assert float((err / np.maximum(np.abs(_cl(x)), 1e-4)).max()) <= 2.0 ** -21
    # range flag (bit 1 = "a value of the split16 tensor is not representable")
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    U.split16_pack(x, flag.data_ptr())
    assert int(flag.item()) == 0
    y = x.clone(); y[0, 5, 3, 4] = 7.0e4
    U.split16_pack(y, flag.data_ptr())
    assert int(flag.item()) == 2


CASES = [
    # nd, sources [(channels, up)], c_out, spatial, act
    (2, [(32, 0)], 32, (45, 70), 1),
    (2, [(64, 0)], 64, (64, 96), 1),
    (2, [(32, 0)], 128, (40, 64), 0),
    (2, [(64, 1), (32, 0)], 32, (48, 80), 1),
    (2, [(128, 1), (128, 0)], 128, (24, 40), 1),
    (3, [(32, 0)], 32, (9, 20, 37), 1),
    (3, [(32, 0)], 64, (8, 16, 32), 1),
    (3, [(64, 1), (32, 0)], 32, (8, 24, 40), 1),
    (3, [(32, 0)], 128, (6, 17, 33), 0),
]


@pytest.mark.parametrize("nd,srcs,co,S,act", CASES)
def test_layer_is_bit_identical_in_every_form(nd, srcs, co, S, act):
    """f32 -> f32 (the round-5 kernel), split16 -> f32, f32 -> split16, split16 -> split16: same values, bit for bit; the split16 output
    is numpy's split of the f32 output"""
    import torch
    from stardist_amd.lib import _native as N
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    Conv = torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d
    torch.manual_seed(sum(S) + co)
    conv = Conv(sum(c for c, _ in srcs), co, 3, padding=1).to(dev)
    ts = []
    for k, (c, up) in enumerate(srcs):
        shp = tuple(s >> up for s in S)
        ts.append((_rand((1, c) + shp, cl, dev, 10 + k, 2.0), up))
    with torch.no_grad(), U.force_conv_mode("f16x3"):
        ref = U._hand_conv(conv, ts, act)                                       # f32 tensors in and out
        packed = [(U.split16_pack(t), up) for t, up in ts]
        got_in = U._hand_conv(conv, packed, act)                               # split16 in, f32 out
        assert not U.is_split16(got_in) and torch.equal(got_in, ref)
        conv.__dict__["_sd_split_out"] = True
        got_out = U._hand_conv(conv, ts, act)                                   # f32 in, split16 out
        got_both = U._hand_conv(conv, packed, act)
        conv.__dict__["_sd_split_out"] = False
    assert U.is_split16(got_out) and U.is_split16(got_both)
    want = _bits(np_split16(_cl(ref)))
    assert np.array_equal(_bits(_cl(got_out)), want)
    assert np.array_equal(_bits(_cl(got_both)), want)
    assert int(U.range_flag(dev).item()) == 0


def test_split16_output_flags_a_value_beyond_the_range():
    import torch
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    conv = torch.nn.Conv2d(32, 32, 3, padding=1).to(dev)
    x = _rand((1, 32, 40, 64), torch.channels_last, dev, 3, 1.0)
    with torch.no_grad(), U.force_conv_mode("f16x3"):
        conv.weight.mul_(1.0e4); conv.bias.fill_(7.0e4)
        conv.__dict__["_sd_split_out"] = True
        U.range_flag(dev).zero_()
        U._hand_conv(conv, [(x, 0)], 1)
        assert int(U.range_flag(dev).item()) & 2
        U.range_flag(dev).zero_()


@pytest.mark.parametrize("kz,S", [(1, (61, 83)), (3, (7, 21, 34))])
def test_first_layer_split16_is_the_split_of_its_f32_output(kz, S):
    import torch
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    nd = 2 if kz == 1 else 3
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    Conv = torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d
    torch.manual_seed(5)
    conv = Conv(1, 32, 3, padding=1).to(dev)
    x = torch.randn((1, 1) + S, device=dev).contiguous(memory_format=cl)
    with torch.no_grad(), U.force_conv_mode("f16x3"):
        ref = U._hand_conv(conv, [(x, 0)], 1)
        conv.__dict__["_sd_split_out"] = True
        got = U._hand_conv(conv, [(x, 0)], 1)
    assert U.is_split16(got) and not U.is_split16(ref)
    assert np.array_equal(_bits(_cl(got)), _bits(np_split16(_cl(ref))))


@pytest.mark.parametrize("shape,pool", [((1, 64, 38, 50), (2, 2)), ((1, 32, 8, 18, 22), (2, 2, 2)), ((1, 32, 6, 18, 22), (1, 2, 2)), ((1, 96, 33, 47), (2, 2))])
def test_maxpool_commutes_with_the_split(shape, pool):
    import torch
    from stardist_amd.models import unet as U
    dev = torch.device("cuda:0")
    cl = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
    x = _rand(shape, cl, dev, 7, 2.0, ties=True)
    with torch.no_grad():
        ref = U.max_pool(x, pool)
        got = U.max_pool(U.split16_pack(x), pool)
    assert U.is_split16(got)
    g, w = _bits(_cl(got)), _bits(np_split16(_cl(ref)))
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        msg = ["%d of %d words differ" % (len(bad), g.size)]
        xa = _cl(x)
        for ix in bad[:4]:
            *pos, c = [int(v) for v in ix]
            ch = (c // 32) * 32 + (c % 16) * 2                          # (first of the two channels packed in the word)
            win = xa[tuple(slice(p * q, p * q + q) for p, q in zip(pos, pool)) + (slice(ch, ch + 2),)].reshape(-1, 2)
            msg.append("out %s word %d (%s): got %08x want %08x, window (two channels) %s" % (pos, c, "hi" if c % 32 < 16 else "lo'", g[tuple(ix)], w[tuple(ix)],
                                                                                           [["%.9g" % v for v in r] for r in win]))
        raise AssertionError("\n".join(msg))


@pytest.mark.parametrize("which", ["2d", "3d", "2d-grid2", "2d-bn", "2d-48"])
def test_network_outputs_do_not_depend_on_the_form(which):
    """whole networks with split16 activations on (the default) and off: identical bits in every output; every planned layer really wrote the form"""
    import torch
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    from stardist_amd.models import unet as U
    from oracle import synth
    dev = torch.device("cuda:0")
    if which == "3d":
        m = StarDist3D(Config3D(n_rays=32), basedir=None, device=dev, seed=0)
        img = np.random.RandomState(0).rand(40, 64, 72).astype(np.float32)
    else:
        kw = {"2d": {}, "2d-grid2": dict(grid=(2, 2)), "2d-bn": dict(unet_batch_norm=True), "2d-48": dict(unet_n_filter_base=48)}[which]
        m = StarDist2D(Config2D(n_rays=32, **kw), basedir=None, device=dev, seed=0)
        img = synth.s2d_nuclei_image(200, 264, seed=3)
    m.net.eval()
    calls = []
    from stardist_amd.lib import _native as N
    orig = N.dcall

    def spy(t, name, *a):
        calls.append(name)
        return orig(t, name, *a)
    N.dcall = spy
    try:
        with U.force_split16(True):
            on = m.predict(img)
        n_fmt = sum(c in ("sd_conv3_f16x3_fmt_ndhwc_device", "sd_conv3_c1x32_split16_device", "sd_maxpool_split16_ndhwc_device") for c in calls)
        n_unpack = sum(c == "sd_split16_unpack_device" for c in calls)
        del calls[:]
        with U.force_split16(False):
            off = m.predict(img)
        assert not any("split16" in c or "_fmt_" in c for c in calls)
    finally:
        N.dcall = orig
    assert n_fmt >= 5 and n_unpack == 0, (n_fmt, n_unpack)
    for a, b in zip(on, off):
        assert np.array_equal(_bits(a), _bits(b))
