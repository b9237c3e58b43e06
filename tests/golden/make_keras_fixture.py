"""Keras-layout weight fixtures written WITHOUT this repository's code (no stardist_amd import): for two small models of the
reference's topologies the script
  * names every layer as Keras / csbdeep name it (csbdeep.internals.blocks.unet_block: down_level_N_no_I, middle_I, up_level_N_no_I;
    unnamed layers get Keras' automatic conv2d, conv2d_1, ... / conv3d, conv3d_1, ... in creation order; heads: features, prob, dist
    as in stardist/models/model2d.py:310-349 and model3d.py:400-447) and stores the variables the way Keras' save_weights stores
    them -- "<layer>/kernel:0" in (k..., c_in, c_out) layout, "<layer>/bias:0" -- in model.layers order (by graph depth; layers of
    equal depth here in creation order: a resnet_block's last body convolution precedes its shortcut projection.  Keras itself breaks
    that tie by its traversal from the outputs, which with csbdeep's Add()([shortcut, body]) lists the projection FIRST: the loader
    therefore restores the creation order from the automatic names and reads both file orders -- tests/test_cpu_reference_build.py runs
    the reference's own _build code over a stand-in Keras with Keras' ordering rule and both operand orders);
  * evaluates the network with its own numpy forward pass written from the Keras layer semantics (Conv 'same' = TensorFlow SAME
    padding incl. strides, MaxPooling 'valid', UpSampling nearest, Concatenate([up, skip]), Add, ReLU, sigmoid);
  * writes config.json + weights_best.npz + expected.npz (input, prob, dist) into tests/golden/keras_fixture/<class>/<name>/.
tests/test_cpu_pretrained.py loads these folders through from_pretrained and must reproduce `expected` -- the loader is thereby
checked against an artefact it did not write.  TEST INFRASTRUCTURE.  usage: python tests/golden/make_keras_fixture.py"""
import json
import os
from itertools import product

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def conv(x, w, b, strides=None):
    """Keras ConvND, padding='same', channels_last: x (*S, cin), w (*k, cin, cout)"""
    nd = x.ndim - 1
    k = w.shape[:nd]
    strides = (1,) * nd if strides is None else strides
    out_shape = tuple(-(-n // s) for n, s in zip(x.shape[:nd], strides))
    pads = []
    for n, kk, s in zip(x.shape[:nd], k, strides):                     # TensorFlow SAME
        tot = max(kk - s, 0) if n % s == 0 else max(kk - n % s, 0)
        pads.append((tot // 2, tot - tot // 2))
    xp = np.pad(x, pads + [(0, 0)])
    out = np.zeros(out_shape + (w.shape[-1],))
    for tap in product(*[range(kk) for kk in k]):
        sl = tuple(slice(t, t + (o - 1) * s + 1, s) for t, o, s in zip(tap, out_shape, strides))
        out += xp[sl] @ w[tap]
    return out + b


def maxpool(x, pool):
    nd = x.ndim - 1
    S = tuple(n // p for n, p in zip(x.shape[:nd], pool))
    x = x[tuple(slice(0, s * p) for s, p in zip(S, pool))]
    shp = sum(((s, p) for s, p in zip(S, pool)), ()) + (x.shape[-1],)
    return x.reshape(shp).max(axis=tuple(range(1, 2 * nd, 2)))


def upsample(x, pool):
    for a, p in enumerate(pool):
        x = np.repeat(x, p, axis=a)
    return x


relu = lambda a: np.maximum(a, 0)
sigmoid = lambda a: 1 / (1 + np.exp(-a))


class Weights(object):
    """variables in creation = file order; Keras' automatic names for unnamed conv layers"""

    def __init__(self, nd, seed):
        self.nd, self.rs, self.auto, self.store = nd, np.random.RandomState(seed), 0, {}

    def make(self, name, cin, cout, k):
        if name is None:
            name = "conv%dd" % self.nd + ("_%d" % self.auto if self.auto else "")
            self.auto += 1
        w = (self.rs.randn(*(tuple(k) + (cin, cout))) * np.sqrt(2.0 / (np.prod(k) * cin))).astype(np.float32)
        b = (self.rs.randn(cout) * 0.1).astype(np.float32)
        self.store[name + "/kernel:0"] = w
        self.store[name + "/bias:0"] = b
        return w.astype(np.float64), b.astype(np.float64)


def unet2d(x, W, n_rays=8, base=4, depth=2, after=6, grid=(2, 2)):
    """model2d.py:310-349 with csbdeep unet_block(n_depth, n_filter_base, (3,3), n_conv_per_depth=2, pool=(2,2))"""
    c = x.shape[-1]
    h = x
    pooled = np.array([1, 1])
    while tuple(pooled) != tuple(grid):                      # model2d.py:317-325: unnamed convs + max-pool in front of the U-Net
        pool = 1 + (np.asarray(grid) > pooled)
        pooled = pooled * pool
        for _ in range(2):
            h = relu(conv(h, *W.make(None, c, base, (3, 3)))); c = base
        h = maxpool(h, tuple(pool))
    skips = []
    for n in range(depth):
        for i in range(2):
            h = relu(conv(h, *W.make("down_level_%d_no_%d" % (n, i), c, base * 2 ** n, (3, 3)))); c = base * 2 ** n
        skips.append(h)
        h = maxpool(h, (2, 2))
    h = relu(conv(h, *W.make("middle_0", c, base * 2 ** depth, (3, 3)))); c = base * 2 ** depth
    h = relu(conv(h, *W.make("middle_2", c, base * 2 ** (depth - 1), (3, 3)))); c = base * 2 ** (depth - 1)
    for n in reversed(range(depth)):
        h = np.concatenate([upsample(h, (2, 2)), skips[n]], -1); c = c + base * 2 ** n
        h = relu(conv(h, *W.make("up_level_%d_no_0" % n, c, base * 2 ** n, (3, 3)))); c = base * 2 ** n
        h = relu(conv(h, *W.make("up_level_%d_no_2" % n, c, base * 2 ** max(0, n - 1), (3, 3)))); c = base * 2 ** max(0, n - 1)
    f = relu(conv(h, *W.make("features", c, after, (3, 3))))
    prob = sigmoid(conv(f, *W.make("prob", after, 1, (1, 1))))[..., 0]
    dist = conv(f, *W.make("dist", after, n_rays, (1, 1)))
    return prob, dist


def resnet3d(x, W, n_rays=6, base=4, n_blocks=2, after=6, grid=(1, 2, 2)):
    """model3d.py:400-447 with csbdeep resnet_block(n_filter, (3,3,3), pool, n_conv_per_block=3, activation='relu')"""
    c = x.shape[-1]
    h = conv(x, *W.make(None, c, base, (7, 7, 7))); c = base                      # linear (model3d.py:414-415)
    h = conv(h, *W.make(None, c, base, (3, 3, 3)))
    pooled = np.array([1, 1, 1])
    nf = base
    for _ in range(n_blocks):
        pool = 1 + (np.asarray(grid) > pooled)
        pooled = pooled * pool
        if any(p > 1 for p in pool):
            nf *= 2
        inp = h
        y = relu(conv(inp, *W.make(None, c, nf, (3, 3, 3)), strides=tuple(pool)))
        y = relu(conv(y, *W.make(None, nf, nf, (3, 3, 3))))
        y = conv(y, *W.make(None, nf, nf, (3, 3, 3)))
        if any(p > 1 for p in pool) or nf != c:
            inp = conv(inp, *W.make(None, c, nf, (1, 1, 1)), strides=tuple(pool))  # created last in the block; same depth as the conv above
        h = relu(inp + y); c = nf
    f = relu(conv(h, *W.make("features", c, after, (3, 3, 3))))
    prob = sigmoid(conv(f, *W.make("prob", after, 1, (1, 1, 1))))[..., 0]
    dist = conv(f, *W.make("dist", after, n_rays, (1, 1, 1)))
    return prob, dist


def write(cls, name, config, W, x, prob, dist):
    d = os.path.join(HERE, "keras_fixture", cls, name)
    os.makedirs(d, exist_ok=True)
    json.dump(config, open(os.path.join(d, "config.json"), "w"))
    json.dump({"prob": 0.5, "nms": 0.4}, open(os.path.join(d, "thresholds.json"), "w"))
    np.savez(os.path.join(d, "weights_best.npz"), **W.store)
    np.savez(os.path.join(d, "expected.npz"), x=x.astype(np.float32), prob=prob.astype(np.float32), dist=dist.astype(np.float32))
    print(cls, name, list(W.store)[:6], "...", len(W.store) // 2, "layers", prob.shape, dist.shape)


rs = np.random.RandomState(7)
x2 = rs.rand(32, 40, 1)
W2 = Weights(2, 1)
p2, d2 = unet2d(x2, W2)
write("StarDist2D", "fixture2d", dict(n_dim=2, axes="YXC", n_channel_in=1, n_rays=8, grid=[2, 2], backbone="unet", unet_n_depth=2,
                                       unet_kernel_size=[3, 3], unet_n_filter_base=4, unet_n_conv_per_depth=2, unet_pool=[2, 2],
                                       unet_activation="relu", unet_last_activation="relu", unet_batch_norm=False, net_conv_after_unet=6),
      W2, x2[..., 0], p2, d2)

x3 = rs.rand(6, 14, 16, 1)
W3 = Weights(3, 2)
p3, d3 = resnet3d(x3, W3)
write("StarDist3D", "fixture3d", dict(n_dim=3, axes="ZYXC", n_channel_in=1, n_rays=6, grid=[1, 2, 2], backbone="resnet",
                                       rays_json={"name": "Rays_GoldenSpiral", "kwargs": {"n": 6, "anisotropy": None}}, resnet_n_blocks=2,
                                       resnet_kernel_size=[3, 3, 3], resnet_n_filter_base=4, resnet_n_conv_per_block=3,
                                       resnet_activation="relu", resnet_batch_norm=False, net_conv_after_resnet=6),
      W3, x3[..., 0], p3, d3)
