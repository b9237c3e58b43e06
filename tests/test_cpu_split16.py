"""CPU: the host side of the split16 activation form (models/unet.py, round 6) -- which layers of which topology write it, and the
arithmetic fact the max-pooling kernel rests on (x -> (hi, lo') is monotone; equal values with different pairs always differ in hi the way x
does).  The kernels themselves: tests/test_gpu_split16.py."""
import numpy as np
import pytest


def _marked(net):
    import torch
    return [n for n, m in net.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d)) and m.__dict__.get("_sd_split_out")]


def test_plan_marks_exactly_the_layers_whose_every_reader_is_a_split_fp16_layer():
    from stardist_amd.models import Config2D, Config3D
    from stardist_amd.models.unet import StarDistNet
    # default U-Nets: every 3x3 layer except the features layer (read by the 1x1 heads); the 1x1 heads never
    n2 = StarDistNet(Config2D(n_rays=32))
    m2 = _marked(n2)
    assert len(m2) == 14 and "features.0" not in m2 and "prob" not in m2 and "backbone.down.0.0.0" in m2 and "backbone.up.2.1.0" in m2
    assert len(_marked(StarDistNet(Config3D(rays=96)))) == 10
    assert len(_marked(StarDistNet(Config2D(n_rays=32, grid=(2, 2))))) == 16                       # the grid stem's two layers as well
    # three input channels: the first layer is read as f32 from the general kernel's output? no -- it WRITES f32 (3 -> 32 is no split-fp16 layer)
    assert "backbone.down.0.0.0" not in _marked(StarDistNet(Config2D(n_rays=32, n_channel_in=3)))
    # 48 base filters: only tensors with 96 / 192 channels whose readers are 32-chunk layers
    m48 = _marked(StarDistNet(Config2D(n_rays=32, unet_n_filter_base=48)))
    assert m48 and all(n not in m48 for n in ("backbone.down.0.0.0", "backbone.down.0.1.0"))
    # no convolution after the U-Net: its last layer feeds the heads directly -> f32
    assert "backbone.up.2.1.0" not in _marked(StarDistNet(Config2D(n_rays=32, net_conv_after_unet=0)))
    # ResNet: nothing (shortcuts, strided layers and the stem read f32)
    assert _marked(StarDistNet(Config3D(rays=96, backbone="resnet"))) == []


def test_split_is_monotone_and_ties_are_ordered_by_hi():
    """what k_maxpool_split16 relies on: for x <= y the values hi + lo' 2^-11 satisfy v(x) <= v(y), and two DIFFERENT pairs with the same value
    (x just below / just above the midpoint of two fp16 numbers) always have hi(x) < hi(y) -- so 'largest value, ties: larger hi' selects
    the pair of max(x, y)"""
    r = np.random.RandomState(0)
    x = np.concatenate([(r.randn(1500000) * 3).astype(np.float32), (r.randn(200000) * 1e-4).astype(np.float32), (r.rand(200000) * 6e4).astype(np.float32)])
    h = x.astype(np.float16)
    nx = np.nextafter(h, np.float16(np.inf))
    mid = ((h.astype(np.float32) + nx.astype(np.float32)) * np.float32(0.5))[np.isfinite(nx)]
    x = np.sort(np.concatenate([x, np.nextafter(mid, np.float32(-np.inf)), np.nextafter(mid, np.float32(np.inf)), mid]))
    x = x[np.abs(x) < 65000]
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    v = hi.astype(np.float32) + lo.astype(np.float32) * np.float32(2.0 ** -11)
    assert np.all(np.diff(v) >= 0)
    tie = (np.diff(v) == 0) & ((hi[1:] != hi[:-1]) | (lo[1:] != lo[:-1]))
    assert tie.sum() > 1000                                                       # the midpoint pairs really produce such ties
    assert np.all(hi[1:][tie].astype(np.float32) > hi[:-1][tie].astype(np.float32))
    # and the 22 bits: relative error of the pair's value (absolute floor below the fp16 normal range)
    big = np.abs(x) > 1e-3
    assert float((np.abs(v[big].astype(np.float64) - x[big]) / np.abs(x[big])).max()) <= 2.0 ** -21


def test_sharded_input_serves_prefetched_regions_and_reads_the_rest_on_demand():
    from stardist_amd.big import ShardedInput

    class Src(object):
        def __init__(self, a): self.a, self.shape, self.dtype, self.reads = a, a.shape, a.dtype, []

        def __getitem__(self, sl):
            self.reads.append(tuple((s.start, s.stop) for s in sl))
            return self.a[sl]
    a = np.arange(40 * 50, dtype=np.float32).reshape(40, 50)
    src = Src(a)
    x = ShardedInput(src)
    assert x.shape == (40, 50) and x.ndim == 2
    sl = (slice(8, 24), slice(0, 32))
    x.prefetch(sl)
    assert src.reads == [((8, 24), (0, 32))] and x.bytes_held == 16 * 32 * 4
    assert np.array_equal(x[sl], a[sl]) and len(src.reads) == 1                      # served from what is held
    assert np.array_equal(x[slice(0, 4), slice(1, 3)], a[0:4, 1:3]) and len(src.reads) == 2   # anything else: from the source
