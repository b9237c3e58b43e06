"""TEST INFRASTRUCTURE ONLY (never imported by the product): an executable stand-in for the handful of Keras names the reference's
network-building code uses, so that the reference's OWN `_build` / `_build_unet` / `_build_resnet` (stardist/models/model2d.py:310-349,
model3d.py:360-447) can be run here -- where neither TensorFlow nor csbdeep exist -- and the graph they build can be evaluated.

What is restated, and from what:

* Keras functional API, inference semantics (tf.keras 2.x; unpinned by the reference's setup.py): `Input`, `Conv2D` / `Conv3D`
  (padding 'same' = TensorFlow SAME incl. strides, `activation=`, `use_bias=`), `MaxPooling2D/3D` ('valid', stride = pool),
  `UpSampling2D/3D` (nearest), `Concatenate`, `Add`, `Activation`, `BatchNormalization` (moving statistics, epsilon 1e-3),
  `Dropout` (identity), `Model`.  Arithmetic in float64 numpy, channels last, one image (no batch axis).
* Keras' automatic layer names (snake-cased class name + "_<k>", zero based, one counter per class and session) and the order of
  `model.layers` -- by graph depth (longest path to an output), layers of equal depth in the order of the depth-first traversal from the
  outputs that visits a layer BEFORE its inputs (keras/engine/functional.py `_map_graph_network` / `_build_map_helper`).  That order is
  the order `save_weights` writes the variables in, i.e. what a weights_*.h5 of the reference holds.
* csbdeep.internals.blocks `unet_block` / `resnet_block` / `conv_block2/3` (csbdeep >= 0.8.0, reference setup.py:140; a third-party
  dependency that is NOT under /root/reference): layer names `down_level_N_no_I`, `middle_I`, `up_level_N_no_I` (+ prefix), block
  structure, `Add()([shortcut, body])`.  Restated from the published source; it cannot be checked offline, so the tests run BOTH operand
  orders of the residual `Add` (it decides which of a block's two equal-depth convolutions comes first in `model.layers`).
"""
import re
from itertools import product

import numpy as np


# --------------------------------------------------------------------------------------------------------------- session state
class _Session(object):
    def __init__(self):
        self.reset()

    def reset(self, seed=0):
        self.counters = {}
        self.rs = np.random.RandomState(seed)
        self.created = []                       # layers with variables, in creation order


SESSION = _Session()


def _snake(name):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    s = re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()
    return s


class KTensor(object):
    """symbolic output of one layer call; shape = spatial extents (None when unknown) + (channels,)"""

    def __init__(self, layer, inputs, shape):
        self.layer, self.inputs, self.shape = layer, list(inputs), tuple(shape)


# ------------------------------------------------------------------------------------------------------------------ layers
class Layer(object):
    def __init__(self, name=None, **kwargs):
        unknown = set(kwargs) - {"kernel_initializer", "axis"}
        if unknown:
            raise TypeError("mini-Keras %s: unexpected arguments %s" % (type(self).__name__, sorted(unknown)))
        if name is None:
            base = _snake(type(self).__name__)
            k = SESSION.counters.get(base, 0)
            SESSION.counters[base] = k + 1
            name = base if k == 0 else "%s_%d" % (base, k)
        self.name = name
        self.variables = {}                     # "<layer>/<variable>:0" -> float32 array, in Keras' per-layer order

    def __call__(self, x):
        inputs = list(x) if isinstance(x, (list, tuple)) else [x]
        assert all(isinstance(t, KTensor) for t in inputs)
        self.build([t.shape for t in inputs])
        return KTensor(self, inputs, self.out_shape([t.shape for t in inputs]))

    def build(self, shapes):
        pass

    def out_shape(self, shapes):
        return shapes[0]


class InputLayer(Layer):
    pass


def Input(shape, name=None):
    lay = InputLayer(name=name)
    return KTensor(lay, [], tuple(shape))


def _tup(v, nd):
    return tuple(int(a) for a in v) if isinstance(v, (tuple, list, np.ndarray)) else (int(v),) * nd


def _activation(name):
    if name in (None, "linear"):
        return lambda a: a
    if name == "relu":
        return lambda a: np.maximum(a, 0)
    if name == "sigmoid":
        return lambda a: 1.0 / (1.0 + np.exp(-a))
    if name == "softmax":
        def f(a):
            e = np.exp(a - a.max(-1, keepdims=True))
            return e / e.sum(-1, keepdims=True)
        return f
    raise ValueError("mini-Keras: activation %r" % (name,))


class _Conv(Layer):
    nd = 0

    def __init__(self, filters, kernel_size, strides=1, padding="valid", activation=None, use_bias=True, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if padding != "same":
            raise ValueError("mini-Keras: only padding='same' is restated")
        self.filters, self.k, self.s = int(filters), _tup(kernel_size, self.nd), _tup(strides, self.nd)
        assert len(self.k) == self.nd and len(self.s) == self.nd
        self.activation, self.use_bias = activation, bool(use_bias)

    def build(self, shapes):
        cin = shapes[0][-1]
        w = (SESSION.rs.randn(*(self.k + (cin, self.filters))) * np.sqrt(2.0 / (np.prod(self.k) * cin))).astype(np.float32)
        self.variables[self.name + "/kernel:0"] = w
        if self.use_bias:
            self.variables[self.name + "/bias:0"] = (SESSION.rs.randn(self.filters) * 0.1).astype(np.float32)
        SESSION.created.append(self)

    def out_shape(self, shapes):
        return tuple(None if n is None else -(-n // s) for n, s in zip(shapes[0][:-1], self.s)) + (self.filters,)

    def compute(self, xs):
        x = xs[0]
        w = self.variables[self.name + "/kernel:0"].astype(np.float64)
        out_shape = tuple(-(-n // s) for n, s in zip(x.shape[:-1], self.s))
        pads = []
        for n, kk, s in zip(x.shape[:-1], self.k, self.s):              # TensorFlow SAME
            tot = max(kk - s, 0) if n % s == 0 else max(kk - n % s, 0)
            pads.append((tot // 2, tot - tot // 2))
        xp = np.pad(x, pads + [(0, 0)])
        out = np.zeros(out_shape + (self.filters,))
        for tap in product(*[range(kk) for kk in self.k]):
            sl = tuple(slice(t, t + (o - 1) * s + 1, s) for t, o, s in zip(tap, out_shape, self.s))
            out += xp[sl] @ w[tap]
        if self.use_bias:
            out = out + self.variables[self.name + "/bias:0"].astype(np.float64)
        return _activation(self.activation)(out)


class Conv2D(_Conv):
    nd = 2


class Conv3D(_Conv):
    nd = 3


class _Pool(Layer):
    nd = 0

    def __init__(self, pool_size=2, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.pool = _tup(pool_size, self.nd)
        assert len(self.pool) == self.nd

    def out_shape(self, shapes):
        return tuple(None if n is None else n // p for n, p in zip(shapes[0][:-1], self.pool)) + (shapes[0][-1],)

    def compute(self, xs):
        x = xs[0]
        S = tuple(n // p for n, p in zip(x.shape[:-1], self.pool))
        x = x[tuple(slice(0, s * p) for s, p in zip(S, self.pool))]
        shp = sum(((s, p) for s, p in zip(S, self.pool)), ()) + (x.shape[-1],)
        return x.reshape(shp).max(axis=tuple(range(1, 2 * self.nd, 2)))


class MaxPooling2D(_Pool):
    nd = 2


class MaxPooling3D(_Pool):
    nd = 3


class _Up(Layer):
    nd = 0

    def __init__(self, size=2, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.size = _tup(size, self.nd)

    def out_shape(self, shapes):
        return tuple(None if n is None else n * p for n, p in zip(shapes[0][:-1], self.size)) + (shapes[0][-1],)

    def compute(self, xs):
        x = xs[0]
        for a, p in enumerate(self.size):
            x = np.repeat(x, p, axis=a)
        return x


class UpSampling2D(_Up):
    nd = 2


class UpSampling3D(_Up):
    nd = 3


class Concatenate(Layer):
    def out_shape(self, shapes):
        return shapes[0][:-1] + (sum(s[-1] for s in shapes),)

    def compute(self, xs):
        return np.concatenate(xs, axis=-1)


class Add(Layer):
    def compute(self, xs):
        out = xs[0]
        for x in xs[1:]:
            out = out + x
        return out


class Activation(Layer):
    def __init__(self, activation, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.activation = activation

    def compute(self, xs):
        return _activation(self.activation)(xs[0])


class Dropout(Layer):
    def __init__(self, rate, name=None, **kwargs):
        super().__init__(name=name, **kwargs)

    def compute(self, xs):
        return xs[0]


class BatchNormalization(Layer):
    def build(self, shapes):
        c = shapes[0][-1]
        rs = SESSION.rs
        for var, val in (("gamma", rs.uniform(0.5, 1.5, c)), ("beta", rs.randn(c) * 0.1), ("moving_mean", rs.randn(c) * 0.1),
                         ("moving_variance", rs.uniform(0.5, 1.5, c))):
            self.variables["%s/%s:0" % (self.name, var)] = val.astype(np.float32)
        SESSION.created.append(self)

    def compute(self, xs):
        v = {k.split("/")[1][:-2]: a.astype(np.float64) for k, a in self.variables.items()}
        return v["gamma"] * (xs[0] - v["moving_mean"]) / np.sqrt(v["moving_variance"] + 1e-3) + v["beta"]


# ------------------------------------------------------------------------------------------------------------------- Model
class Model(object):
    def __init__(self, inputs, outputs):
        self.inputs, self.outputs = list(inputs), list(outputs)
        self.layers = self._map_graph()

    def _map_graph(self):
        """keras/engine/functional.py _map_graph_network: every layer is called once in these graphs, so a node is its tensor"""
        layer_indices, order, finished = {}, [], set()

        def build_map(t):
            if id(t) in finished:
                return
            if t.layer not in layer_indices:                           # traversal order: a layer before its inputs
                layer_indices[t.layer] = len(layer_indices)
            for p in t.inputs:
                build_map(p)
            finished.add(id(t))
            order.append(t)
        for t in self.outputs:
            build_map(t)
        depth = {}
        for t in reversed(order):                                      # consumers before producers
            d = depth.setdefault(id(t), 0)
            for p in t.inputs:
                depth[id(p)] = max(d + 1, depth.get(id(p), 0))
        by_depth = {}
        for t in order:
            d = depth[id(t)]
            if not t.inputs:                                           # input layers sit at the maximal depth
                d = max(depth.values())
            by_depth.setdefault(d, []).append(t.layer)
        layers = []
        for d in sorted(by_depth, reverse=True):
            layers += sorted(by_depth[d], key=lambda l: layer_indices[l])
        return layers

    def weights_in_file_order(self):
        """{variable name: array} in the order keras.Model.save_weights stores them (model.layers order, per layer its own order)"""
        out = {}
        for lay in self.layers:
            out.update(lay.variables)
        return out

    def predict(self, x):
        """x: one image, spatial axes + channels; returns the list of outputs (float64)"""
        memo = {}

        def ev(t):
            if id(t) not in memo:
                memo[id(t)] = np.asarray(x, np.float64) if not t.inputs else t.layer.compute([ev(p) for p in t.inputs])
            return memo[id(t)]
        return [ev(t) for t in self.outputs]


# --------------------------------------------------------------------------- csbdeep.internals.blocks, restated (see module docstring)
def _conv_block(nd):
    Conv = Conv2D if nd == 2 else Conv3D

    def conv_block(n_filter, *kernel, activation="relu", border_mode="same", dropout=0.0, batch_norm=False, init="glorot_uniform", **kwargs):
        def f(lay):
            if batch_norm:
                s = Conv(n_filter, kernel, padding=border_mode, kernel_initializer=init, **kwargs)(lay)
                s = BatchNormalization()(s)
                s = Activation(activation)(s)
            else:
                s = Conv(n_filter, kernel, padding=border_mode, kernel_initializer=init, activation=activation, **kwargs)(lay)
            if dropout is not None and dropout > 0:
                s = Dropout(dropout)(s)
            return s
        return f
    return conv_block


def unet_block(n_depth=2, n_filter_base=16, kernel_size=(3, 3), n_conv_per_depth=2, activation="relu", batch_norm=False, dropout=0.0,
               last_activation=None, pool=(2, 2), kernel_init="glorot_uniform", expansion=2, prefix=""):
    if len(pool) != len(kernel_size):
        raise ValueError("kernel and pool sizes must match.")
    nd = len(kernel_size)
    if nd not in (2, 3):
        raise ValueError("unet_block only 2d or 3d.")
    conv_block = _conv_block(nd)
    Pool = MaxPooling2D if nd == 2 else MaxPooling3D
    Up = UpSampling2D if nd == 2 else UpSampling3D
    if last_activation is None:
        last_activation = activation
    _name = lambda s: prefix + s

    def f(inp):
        skips, layer = [], inp
        for n in range(n_depth):
            for i in range(n_conv_per_depth):
                layer = conv_block(int(n_filter_base * expansion ** n), *kernel_size, dropout=dropout, activation=activation, init=kernel_init,
                                   batch_norm=batch_norm, name=_name("down_level_%s_no_%s" % (n, i)))(layer)
            skips.append(layer)
            layer = Pool(pool, name=_name("max_%s" % n))(layer)
        for i in range(n_conv_per_depth - 1):
            layer = conv_block(int(n_filter_base * expansion ** n_depth), *kernel_size, dropout=dropout, activation=activation, init=kernel_init,
                               batch_norm=batch_norm, name=_name("middle_%s" % i))(layer)
        layer = conv_block(int(n_filter_base * expansion ** max(0, n_depth - 1)), *kernel_size, dropout=dropout, activation=activation,
                           init=kernel_init, batch_norm=batch_norm, name=_name("middle_%s" % n_conv_per_depth))(layer)
        for n in reversed(range(n_depth)):
            layer = Concatenate(axis=-1)([Up(pool)(layer), skips[n]])
            for i in range(n_conv_per_depth - 1):
                layer = conv_block(int(n_filter_base * expansion ** n), *kernel_size, dropout=dropout, activation=activation, init=kernel_init,
                                   batch_norm=batch_norm, name=_name("up_level_%s_no_%s" % (n, i)))(layer)
            layer = conv_block(int(n_filter_base * expansion ** max(0, n - 1)), *kernel_size, dropout=dropout,
                               activation=activation if n > 0 else last_activation, init=kernel_init, batch_norm=batch_norm,
                               name=_name("up_level_%s_no_%s" % (n, n_conv_per_depth)))(layer)
        return layer
    return f


def make_resnet_block(shortcut_first=True):
    """csbdeep resnet_block; shortcut_first: the operand order of the residual Add -- Add()([shortcut, body]) (the published source) or
    the other way round (run as well: the order cannot be verified offline and decides the tie in model.layers)"""

    def resnet_block(n_filter, kernel_size=(3, 3), pool=(1, 1), n_conv_per_block=2, batch_norm=False, kernel_initializer="he_normal",
                     activation="relu"):
        if n_conv_per_block < 2:
            raise ValueError("required: n_conv_per_block >= 2")
        if len(pool) != len(kernel_size):
            raise ValueError("kernel and pool sizes must match.")
        nd = len(kernel_size)
        Conv = Conv2D if nd == 2 else Conv3D
        kw = dict(padding="same", use_bias=not batch_norm, kernel_initializer=kernel_initializer)

        def f(inp):
            x = Conv(n_filter, kernel_size, strides=pool, **kw)(inp)
            if batch_norm:
                x = BatchNormalization()(x)
            x = Activation(activation)(x)
            for _ in range(n_conv_per_block - 2):
                x = Conv(n_filter, kernel_size, **kw)(x)
                if batch_norm:
                    x = BatchNormalization()(x)
                x = Activation(activation)(x)
            x = Conv(n_filter, kernel_size, **kw)(x)
            if batch_norm:
                x = BatchNormalization()(x)
            if any(p != 1 for p in pool) or n_filter != inp.shape[-1]:
                inp = Conv(n_filter, (1,) * nd, strides=pool, **kw)(inp)
            x = Add()([inp, x] if shortcut_first else [x, inp])
            return Activation(activation)(x)
        return f
    return resnet_block
