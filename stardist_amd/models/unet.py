"""The StarDist network re-expressed in PyTorch-ROCm (channels_last, convolutions on MFMA).

Topology restated from the reference's Keras graph:
  StarDist2D._build          stardist/models/model2d.py:310-349
  StarDist3D._build_unet     stardist/models/model3d.py:360-399
  StarDist3D._build_resnet   stardist/models/model3d.py:402-447
  csbdeep.internals.blocks.unet_block / resnet_block (csbdeep>=0.8.0, not vendored; published
  semantics restated: 'same' zero padding, max-pool 'valid' stride=pool, nearest up-sampling,
  Concatenate([up, skip]) in that order).
Keras layer names are kept as module names so a Keras weight file maps 1:1 (kernel
(k..., cin, cout) -> torch (cout, cin, k...)).
U-Net parity against TensorFlow is unpinned in this environment (no TF, no weights).
"""
import ctypes
import threading

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _act(name):
    if name in (None, "linear"):
        return nn.Identity()
    if name == "relu":
        return nn.ReLU(inplace=True)
    if name == "elu":
        return nn.ELU(inplace=True)
    if name == "sigmoid":
        return nn.Sigmoid()
    if name == "tanh":
        return nn.Tanh()
    raise ValueError("activation %s not supported" % name)


# ---- hand-written convolutions (csrc/conv3x3*.hip, conv_general.hip) -----------------------------------------------------------
# GPU inference runs EVERY convolution, pooling and head of the network on the library's own kernels; a layer none of them covers
# raises UnsupportedLayer with the layer's shape (there is no library / framework fallback on the device).  The plain torch modules
# below remain what they are everywhere else: on the CPU (float64 references of the tests, the flagged CPU baseline of bench.py) and
# under autograd.
#
# Conv2D(3x3) / Conv3D(3x3x3) 'same' layers with 1 or a multiple of 32 (<= 512) input channels and a multiple of 32 output channels
# -- including Concatenate([UpSampling(x), skip]) in front of them -- run as implicit GEMMs on the matrix cores with up-sampling,
# concatenation, bias, batch-norm and activation folded in.
class UnsupportedLayer(NotImplementedError):
    """a network layer that no hand-written kernel covers (GPU inference has no library fallback)"""


_CONV_MODES = ("f16x3", "bf16x6", "hand")
_mode_override = []


def conv_mode():
    """Which kernel the 3x3 / 3x3x3 layers over 32-channel chunks run on (STARDIST_AMD_CONV, read per call; force_conv_mode overrides):
      'f16x3' (default)   csrc/conv3x3_f16.hip: every f32 product as three fp16 x fp16 MFMA products (two fp16 terms per operand, the
                          cross terms in their own f32 accumulator) -- f32-accurate: layers and networks within 3e-6 of a float64
                          evaluation, the same 1e-5 tests as the exact kernel.  An activation outside the fp16 range raises a device
                          flag; the model then re-evaluates with 'bf16x6' (StarDistBase._net_forward).
      'bf16x6'            csrc/conv3x3_bf16.hip: six bf16 x bf16 products per f32 product (three bf16 terms per operand); no range limit
      'hand' / 'f32'      csrc/conv3x3.hip: exact f32 MFMA kernel (one fma chain per output)
    The one-channel first layer and the general kernel (csrc/conv_general.hip) are exact f32 in every mode."""
    if _mode_override:
        return _mode_override[-1]
    import os
    m = os.environ.get("STARDIST_AMD_CONV", "f16x3")
    return "hand" if m in ("hand", "f32") else (m if m in _CONV_MODES else "f16x3")


class force_conv_mode(object):
    """context manager: `with force_conv_mode("bf16x6"): ...` (takes precedence over the environment variable)"""

    def __init__(self, mode):
        assert mode in _CONV_MODES, mode
        self.mode = mode

    def __enter__(self):
        _mode_override.append(self.mode)
        return self

    def __exit__(self, *exc):
        _mode_override.pop()
        return False


_range_flags = {}
N_FLAG_SLOTS = 256
_slot_counter = [0]


class _PerThread(threading.local):
    """state of the forward pass a thread is running: the flag tensor of ITS model (two threads predicting with two models on one
    device each report into their own words), and whether a layer asked for the pass to be repeated (split16_replan)"""

    def __init__(self):
        self.flag_stack = []
        self.replan = False


_tls = _PerThread()


def range_flag(device):
    """the device word the split-fp16 convolutions OR with 1 when an activation they READ lies outside the fp16 range (|x| > 65504 or
    infinite; a NaN simply propagates into the result as it does in any float32 evaluation) -- the default word, used by layers
    evaluated outside a model's forward pass (one per device)"""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _range_flags.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _range_flags[key] = t
    return t


class use_range_flags(object):
    """context manager: inside it every split-fp16 layer reports into ITS OWN word of `flags` (int32 tensor of N_FLAG_SLOTS words owned by
    the calling model -- two models on one device do not share state, and the captured HIP graphs keep pointing at their model's words):
    StarDistBase._net_forward reads the words after a pass and moves exactly the offending layers to the bf16x6 form"""

    def __init__(self, flags):
        assert flags.dtype == torch.int32 and flags.numel() == N_FLAG_SLOTS
        self.flags = flags

    def __enter__(self):
        _tls.flag_stack.append(self.flags)
        return self

    def __exit__(self, *exc):
        _tls.flag_stack.pop()
        return False


def flag_slot(conv):
    """the word (1 .. N_FLAG_SLOTS - 1) a convolution module reports its range flag into; assigned on first use.  A model is not
    re-entrant: one thread at a time per model (its flag words and captured graphs are per model, the stack of active flag tensors per thread)"""
    s = conv.__dict__.get("_sd_flag_slot")
    if s is None:
        _slot_counter[0] = _slot_counter[0] % (N_FLAG_SLOTS - 1) + 1
        s = conv.__dict__["_sd_flag_slot"] = _slot_counter[0]
    return s


def _flag_ptr(conv, device):
    st = _tls.flag_stack
    if st and st[-1].device == torch.device(device):
        return st[-1].data_ptr() + 4 * flag_slot(conv)
    return range_flag(device).data_ptr()


# ---- split16 activations (include/stardist_hip.h "split16"; csrc/conv3x3_layout.h) -------------------------------------------------
# Between two split-fp16 layers an activation tensor travels as the two fp16 terms (hi, lo') the consuming kernel multiplies with --
# made once in the producer's epilogue instead of once per consumer workgroup and unit.  Same shape, strides and bytes per value as
# the f32 tensor it stands for (torch dtype float32, tagged with `_sd_split16`); results are bit-identical to the f32 form.
# Which layers write it is planned from the topology (StarDistNet._plan_split16: a layer whose every consumer is a 3x3 layer over
# 32-channel chunks, directly or through a max-pooling); a consumer that cannot read the form after all (pinned to bf16x6) unpacks it,
# clears the producer's mark and asks for the pass to be repeated (`split16_replan`), so a result never depends on the form.
_split16_override = []


def split16_replan(value=None):
    """the calling thread's "repeat the pass" request (set by _unpack_for, read and cleared by StarDistBase._net_forward)"""
    if value is not None:
        _tls.replan = bool(value)
    return _tls.replan


def split16_enabled():
    """split16 activations between split-fp16 layers (STARDIST_AMD_SPLIT16=0 or force_split16(False): f32 tensors everywhere)"""
    if _split16_override:
        return _split16_override[-1]
    import os
    return os.environ.get("STARDIST_AMD_SPLIT16", "1") != "0"


class force_split16(object):
    """context manager: `with force_split16(False): ...` (takes precedence over the environment variable)"""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        _split16_override.append(self.on)
        return self

    def __exit__(self, *exc):
        _split16_override.pop()
        return False


def is_split16(t):
    return bool(getattr(t, "_sd_split16", False))


def _tag_split16(t, producer):
    t._sd_split16 = True
    t._sd_producer = producer
    return t


def split16_unpack(t):
    """the f32 tensor hi + lo' * 2^-11 of a split16 tensor (a consumer that only reads f32); a plain tensor is returned as it is"""
    if not is_split16(t):
        return t
    from ..lib import _native as N
    out = torch.empty_like(t)
    C = int(t.shape[1])
    N.dcall(t, "sd_split16_unpack_device", ctypes.c_void_p(t.data_ptr()), int(t.numel() // C), C, ctypes.c_void_p(out.data_ptr()))
    return out


def split16_pack(t, flag_ptr=None):
    """split16 form of a channels-last f32 tensor (1, C, *spatial), C a multiple of 32 (tests; sd_split16_pack_device)"""
    from ..lib import _native as N
    nd = t.dim() - 2
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    assert t.dtype == torch.float32 and t.shape[0] == 1 and t.shape[1] % 32 == 0 and t.is_contiguous(memory_format=cl)
    out = torch.empty_like(t)
    C = int(t.shape[1])
    N.dcall(t, "sd_split16_pack_device", ctypes.c_void_p(t.data_ptr()), int(t.numel() // C), C, ctypes.c_void_p(out.data_ptr()),
            ctypes.c_void_p(flag_ptr) if flag_ptr else None)
    return _tag_split16(out, None)


def _unpack_for(conv, t):
    """`conv` cannot read the split16 tensor t: f32 copy for this pass; its producer writes f32 from now on and the pass is repeated"""
    prod = getattr(t, "_sd_producer", None)
    if prod is not None and prod.__dict__.get("_sd_split_out"):
        prod.__dict__["_sd_split_out"] = False
        split16_replan(True)
    return split16_unpack(t)


def _native_inference(x):
    """the hand-written path applies: tensor on a HIP device, no autograd, no autocast"""
    return x.is_cuda and not torch.is_grad_enabled() and not torch.is_autocast_enabled()


def _layer_desc(conv, srcs=None):
    return "%s %s -> %d, kernel %s, stride %s%s" % (
        type(conv).__name__, conv.in_channels if srcs is None else " + ".join(str(int(t.shape[1])) + ("(up)" if any(np.atleast_1d(u)) else "") for t, u in srcs),
        conv.out_channels, tuple(conv.kernel_size), tuple(conv.stride),
        "" if srcs is None else ", input %s" % (tuple(srcs[0][0].shape[2:]),))


def _bn_fold(conv, bn):
    """(kernel, bias) float32 numpy of conv followed by an inference BatchNormalization (csbdeep conv_block: Conv -> BN -> Activation,
    Keras moving statistics): w' = w * s, b' = (b - mean) * s + beta with s = gamma / sqrt(var + eps), folded in float64"""
    w = conv.weight.detach().double().cpu().numpy()
    b = conv.bias.detach().double().cpu().numpy() if conv.bias is not None else np.zeros(w.shape[0])
    if bn is not None:
        g = bn.weight.detach().double().cpu().numpy() if bn.weight is not None else np.ones(w.shape[0])
        beta = bn.bias.detach().double().cpu().numpy() if bn.bias is not None else np.zeros(w.shape[0])
        sc = g / np.sqrt(bn.running_var.detach().double().cpu().numpy() + bn.eps)
        w = w * sc.reshape((-1,) + (1,) * (w.ndim - 1))
        b = (b - bn.running_mean.detach().double().cpu().numpy()) * sc + beta
    return np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)


_FORM_PREFIX = {"conv3": "sd_conv3", "bf16x6": "sd_conv3_bf16x6", "f16x3": "sd_conv3_f16x3"}


class _WeightRange(ValueError):
    """a kernel with weights outside the fp16 range: the layer takes the bf16x6 form"""


def _packed_conv_weights(conv, form="conv3", bn=None):
    """(packed kernel, bias) on the device for the native layer `form`: 'conv3' (sd_conv3_ndhwc_device), 'bf16x6'
    (sd_conv3_bf16x6_ndhwc_device), 'f16x3' (sd_conv3_f16x3_ndhwc_device) or 'general' (sd_convg_ndhwc_device); an inference batch-norm
    layer behind the convolution is folded in.  Cached per module (inference: invalidated when a parameter changes)."""
    from ..lib import _native as N
    ver = lambda t: None if t is None else (t.data_ptr(), t._version)
    key = (ver(conv.weight), ver(conv.bias), str(conv.weight.device)) + \
        (() if bn is None else (ver(bn.weight), ver(bn.bias), ver(bn.running_mean), ver(bn.running_var), bn.eps))
    slot = "_sd_packed_" + form
    cache = conv.__dict__.get(slot)
    if cache is None or cache[0] != key:
        w, b = _bn_fold(conv, bn)
        co, ci = int(w.shape[0]), int(w.shape[1])
        k = tuple(int(v) for v in w.shape[2:])
        L = N.lib()
        if form == "general":
            kz, ky, kx = ((1,) + k) if len(k) == 2 else k
            n = int(L.sd_convg_packed_floats(ci, co, kz, ky, kx))
            if n < 0:
                raise ValueError("sd_convg: unsupported layer %d -> %d, kernel %s" % (ci, co, k))
            packed = np.zeros(n, np.float32)
            N.check(L.sd_convg_pack_weights_host(N.ptr(w), ci, co, kz, ky, kx, N.ptr(packed)))
        else:
            prefix = _FORM_PREFIX[form]
            kz = 3 if w.ndim == 5 else 1
            n = int(getattr(L, prefix + "_packed_floats")(ci, co, kz))
            if n < 0:
                raise ValueError("%s: unsupported channel counts %d -> %d" % (prefix, ci, co))
            packed = np.empty(n, np.float32)
            rc = getattr(L, prefix + "_pack_weights_host")(N.ptr(w), ci, co, kz, N.ptr(packed))
            if form == "f16x3" and rc == -2:
                conv.__dict__[slot] = (key, None, None)
                raise _WeightRange(L.sd_last_error().decode(errors="replace"))
            N.check(rc)
        has_bias = conv.bias is not None or bn is not None
        cache = (key, torch.from_numpy(packed).to(conv.weight.device), torch.from_numpy(b).to(conv.weight.device) if has_bias else None)
        conv.__dict__[slot] = cache
    if cache[1] is None:
        raise _WeightRange("weights outside the fp16 range")
    return cache[1], cache[2]


def tf_same_pad_before(n, k, s):
    """TensorFlow 'SAME': total = max(k - s, 0) if n % s == 0 else max(k - n % s, 0); the smaller half goes in front"""
    total = max(k - s, 0) if n % s == 0 else max(k - n % s, 0)
    return total // 2


def _general_conv(conv, x, kind, res=None, bn=None, tf_same=False):
    """act(conv(x) + bias (+ res)) by the general hand-written kernel (any kernel size / stride / channel counts; csrc/conv_general.hip).
    tf_same: Keras padding='same' semantics for a strided layer (asymmetric, computed from the input size) instead of conv.padding.
    None when the layer is not covered."""
    nd = x.dim() - 2
    if not (nd in (2, 3) and x.shape[0] == 1 and conv.groups == 1 and all(d == 1 for d in conv.dilation) and conv.weight.dtype == torch.float32
            and x.dtype == torch.float32 and x.is_cuda and x.device == conv.weight.device and x.shape[1] == conv.in_channels):
        return None
    from ..lib import _native as N
    k3 = (1,) * (3 - nd) + tuple(int(v) for v in conv.kernel_size)
    s3 = (1,) * (3 - nd) + tuple(int(v) for v in conv.stride)
    if int(N.lib().sd_convg_packed_floats(conv.in_channels, conv.out_channels, *k3)) < 0:
        return None
    S3 = (1,) * (3 - nd) + tuple(int(v) for v in x.shape[2:])
    if tf_same:
        p3 = tuple(tf_same_pad_before(n, k, st) for n, k, st in zip(S3, k3, s3))
        O3 = tuple(-(-n // st) for n, st in zip(S3, s3))
    else:
        if not all(isinstance(v, int) for v in conv.padding):
            return None
        p3 = (0,) * (3 - nd) + tuple(int(v) for v in conv.padding)
        O3 = tuple((n + 2 * p - k) // st + 1 for n, p, k, st in zip(S3, p3, k3, s3))
    if any(o <= 0 for o in O3):
        return None
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    if not (x.is_contiguous(memory_format=cl) and x.data_ptr() % 16 == 0):
        x = x.clone(memory_format=cl)
    co = conv.out_channels
    wp, bias = _packed_conv_weights(conv, "general", bn)
    out = torch.empty((1, co) + O3[3 - nd:], dtype=torch.float32, device=x.device, memory_format=cl)
    if res is not None:
        if not (tuple(res.shape) == tuple(out.shape) and res.dtype == torch.float32 and res.is_contiguous(memory_format=cl)):
            return None
    N.dcall(x, "sd_convg_ndhwc_device", ctypes.c_void_p(x.data_ptr()), conv.in_channels, conv.in_channels, *S3, *k3, *s3, *p3, *O3,
            ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bias.data_ptr()) if bias is not None else None,
            ctypes.c_void_p(res.data_ptr()) if res is not None else None, co, co, kind, ctypes.c_void_p(out.data_ptr()), co)
    return out


_MAX_CHUNK_CHANNELS = 512          # csrc/conv3x3_layout.h MAX_CHUNKS * 32


NO_STORE = object()          # _hand_conv(..., no_store=True): the layer ran without writing its output (fused head only)


def _hand_conv(conv, srcs, kind, res=None, bn=None, tf_same=False, dot=None, no_store=False):
    """act(conv(cat(srcs, 1)) + bias (+ res)) by a hand-written kernel; srcs = [(tensor (1, C, *spatial) channels-last float32, up)] with
    up = per-axis tuple of 0/1 (or one int for all axes): 1 where the source has half the output resolution and the reference
    up-samples it (nearest, x2) first.  res: residual added before the activation (resnet_block's Add); bn: inference batch-norm layer
    between convolution and activation (folded into kernel and bias).  3x3(x3) stride-1 'same' layers over 32-channel chunks (and the
    one-channel first layer) go to csrc/conv3x3*.hip, everything else with one full-resolution source to csrc/conv_general.hip.
    dot = (weights (c_out,), holder list): a one-channel head fused into the layer's epilogue when the split-fp16 kernel takes the layer
    (sd_conv3_f16x3_dot_ndhwc_device) -- holder[0] then receives the per-lane terms (n_pix, c_out / 4); left empty otherwise.
    no_store (with dot): when the fused head is taken the layer's own output is NOT written and NO_STORE is returned (the caller evaluates
    the layer on the pixels it needs with conv_rows); otherwise ignored.
    None when the layer is not covered (the callers raise UnsupportedLayer)."""
    nd = 2 if isinstance(conv, nn.Conv2d) else (3 if isinstance(conv, nn.Conv3d) else 0)
    if not (nd and kind in (0, 1) and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and conv.groups == 1 and conv.weight.dtype == torch.float32 and 1 <= len(srcs) <= 2) or (bn is not None and bn.training):
        return None
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    cs, ups = [], []
    for t, up in srcs:
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == nd + 2 and t.shape[0] == 1 and t.device == conv.weight.device):
            return None
        cs.append(int(t.shape[1]))
        ups.append(tuple(int(bool(v)) for v in up) if isinstance(up, (tuple, list)) else (int(bool(up)),) * nd)
    co = conv.out_channels
    if sum(cs) != conv.in_channels:
        return None
    is3 = (tuple(conv.kernel_size) == (3,) * nd and tuple(conv.stride) == (1,) * nd and tuple(conv.padding) == (1,) * nd
           and tuple(conv.dilation) == (1,) * nd and not tf_same)
    if cs == [1]:
        ok = is3 and co % 4 == 0 and not any(ups[0]) and res is None
    else:
        ok = is3 and all(c % 32 == 0 and c > 0 for c in cs) and sum(cs) <= _MAX_CHUNK_CHANNELS and co % 32 == 0
    for t, _ in srcs:                                    # (who reads a layer's output: the range fallback pins the readers of a split16 tensor)
        prod = getattr(t, "_sd_producer", None)
        if prod is not None:
            prod.__dict__.setdefault("_sd_consumers", set()).add(conv)
    if not ok:
        if len(srcs) == 1 and not any(ups[0]):
            return _general_conv(conv, _unpack_for(conv, srcs[0][0]), kind, res, bn, tf_same)
        return None
    shape = tuple(int(s) << u for s, u in zip(srcs[0][0].shape[2:], ups[0]))          # output = full resolution
    for (t, _), up in zip(srcs, ups):
        if tuple(int(s) << u for s, u in zip(t.shape[2:], up)) != shape:
            return None
    from ..lib import _native as N
    # channels-last operands (a pooling layer may hand over a tensor in the default layout: one copy at its resolution)
    srcs = [(t if is_split16(t) or (t.is_contiguous(memory_format=cl) and t.data_ptr() % 16 == 0) else t.clone(memory_format=cl), up) for t, up in srcs]
    form = "conv3" if (cs == [1] or conv_mode() == "hand") else conv_mode()
    if form == "f16x3" and conv.__dict__.get("_sd_force_form") == "bf16x6":
        form = "bf16x6"                                  # this layer has seen an activation beyond the fp16 range (StarDistBase._net_forward)
    if form == "f16x3":
        try:
            wp, bias = _packed_conv_weights(conv, form, bn)
        except _WeightRange:
            form = "bf16x6"
    if form != "f16x3":
        wp, bias = _packed_conv_weights(conv, form, bn)
    # split16 operands: all sources or none (a layer on another kernel form reads f32 only)
    in_split = form == "f16x3" and res is None and all(is_split16(t) for t, _ in srcs)
    if not in_split:
        srcs = [(_unpack_for(conv, t), up) for t, up in srcs]
    out_split = bool(conv.__dict__.get("_sd_split_out")) and split16_enabled() and res is None and dot is None and conv_mode() == "f16x3" \
        and (form == "f16x3" or (cs == [1] and co == 32))
    dot_ok = dot is not None and res is None and dot[0].numel() == co and dot[0].data_ptr() % 16 == 0
    skip_out = bool(no_store) and form == "f16x3" and dot_ok and not out_split
    out = None if skip_out else torch.empty((1, co) + shape, dtype=torch.float32, device=conv.weight.device, memory_format=cl)
    if res is not None and not (tuple(res.shape) == (1, co) + shape and res.dtype == torch.float32 and res.is_contiguous(memory_format=cl)):
        return None
    D, H, W = ((1,) + shape) if nd == 2 else shape
    mask = lambda up: sum(b << k for k, b in enumerate(reversed(up)))                     # bit 0: x, 1: y, 2: z
    a, b = srcs[0][0], (srcs[1][0] if len(srcs) == 2 else None)
    if cs == [1] and out_split:
        # the one-channel first layer writing the split16 form its reader takes
        N.dcall(a, "sd_conv3_c1x32_split16_device", ctypes.c_void_p(a.data_ptr()), D, H, W, 1 if nd == 2 else 3, ctypes.c_void_p(wp.data_ptr()),
                ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, kind, ctypes.c_void_p(out.data_ptr()),
                ctypes.c_void_p(_flag_ptr(conv, a.device)))
        return _tag_split16(out, conv)
    if form == "f16x3" and (in_split or out_split or skip_out):
        part = None
        if dot_ok:
            part = torch.empty((D * H * W, co // 4), dtype=torch.float32, device=a.device)
        N.dcall(a, "sd_conv3_f16x3_fmt_ndhwc_device", ctypes.c_void_p(a.data_ptr()), cs[0], mask(ups[0]),
                ctypes.c_void_p(b.data_ptr()) if b is not None else None, cs[1] if b is not None else 0, mask(ups[1]) if b is not None else 0,
                D, H, W, 1 if nd == 2 else 3, ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bias.data_ptr()) if bias is not None else None,
                co, kind, ctypes.c_void_p(out.data_ptr()) if out is not None else None, int(in_split), int(out_split), ctypes.c_void_p(_flag_ptr(conv, a.device)),
                ctypes.c_void_p(dot[0].data_ptr()) if part is not None else None, ctypes.c_void_p(part.data_ptr()) if part is not None else None)
        if part is not None:
            dot[1].append(part)
        if skip_out:
            return NO_STORE
        return _tag_split16(out, conv) if out_split else out
    args = [ctypes.c_void_p(a.data_ptr()), cs[0], cs[0], mask(ups[0]),
            ctypes.c_void_p(b.data_ptr()) if b is not None else None, cs[1] if b is not None else 0, cs[1] if b is not None else 0,
            mask(ups[1]) if b is not None else 0, D, H, W, 1 if nd == 2 else 3, ctypes.c_void_p(wp.data_ptr()),
            ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, ctypes.c_void_p(res.data_ptr()) if res is not None else None,
            co if res is not None else 0, co, kind, ctypes.c_void_p(out.data_ptr())]
    if form == "f16x3":
        args.append(ctypes.c_void_p(_flag_ptr(conv, a.device)))
        if dot is not None and res is None and dot[0].numel() == co and dot[0].data_ptr() % 16 == 0:
            part = torch.empty((D * H * W, co // 4), dtype=torch.float32, device=a.device)
            dargs = args[:14] + args[16:] + [ctypes.c_void_p(dot[0].data_ptr()), ctypes.c_void_p(part.data_ptr())]      # (no residual arguments)
            N.dcall(a, "sd_conv3_f16x3_dot_ndhwc_device", *dargs)
            dot[1].append(part)
            return out
    N.dcall(a, _FORM_PREFIX[form] + "_res_ndhwc_device", *args)
    return out


def conv_rows(conv, x, kind, rows, bn=None):
    """act(conv(x) + bias) on the pixels `rows` (int64 linear indices into x's spatial grid) of a 3x3(x3) layer the split-fp16 kernel takes:
    (len(rows), c_out) float32, bit-identical to the rows of the dense layer output (sd_conv3_f16x3_rows_device); x (1, C, *spatial)
    channels-last, f32 or split16"""
    from ..lib import _native as N
    nd = x.dim() - 2
    wp, bias = _packed_conv_weights(conv, "f16x3", bn)
    co = conv.out_channels
    out = torch.empty((int(rows.shape[0]), co), dtype=torch.float32, device=x.device)
    if rows.shape[0]:
        S = (1,) * (3 - nd) + tuple(int(v) for v in x.shape[2:])
        N.dcall(x, "sd_conv3_f16x3_rows_device", ctypes.c_void_p(x.data_ptr()), int(x.shape[1]), int(is_split16(x)), *S, 1 if nd == 2 else 3,
                ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, co, kind,
                ctypes.c_void_p(rows.data_ptr()), int(rows.shape[0]), ctypes.c_void_p(out.data_ptr()))
    return out


def _upcat_general(conv, x, skip, pool, kind, bn=None):
    """coverage path of an up level the fused kernels do not take (e.g. n_filter_base = 48: 96 + 96 input channels): UpSampling +
    Concatenate materialised by the native one-pass kernel (sd_upcat_ndhwc_device), then the general convolution kernel.  None when
    not applicable."""
    nd = x.dim() - 2
    if not (nd in (2, 3) and all(p in (1, 2) for p in pool) and x.shape[0] == 1 and x.dtype == torch.float32 and skip.dtype == torch.float32
            and x.shape[1] % 4 == 0 and skip.shape[1] % 4 == 0 and x.shape[1] + skip.shape[1] == conv.in_channels
            and tuple(int(s) * int(p) for s, p in zip(x.shape[2:], pool)) == tuple(int(s) for s in skip.shape[2:])):
        return None
    from ..lib import _native as N
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    x, skip = _unpack_for(conv, x), _unpack_for(conv, skip)
    a = x if x.is_contiguous(memory_format=cl) and x.data_ptr() % 16 == 0 else x.clone(memory_format=cl)
    b = skip if skip.is_contiguous(memory_format=cl) and skip.data_ptr() % 16 == 0 else skip.clone(memory_format=cl)
    S = (1,) * (3 - nd) + tuple(int(v) for v in skip.shape[2:])
    up = sum((1 << k) for k, p in enumerate(reversed(pool)) if p == 2)                    # bit 0: x, 1: y, 2: z
    cat = torch.empty((1, a.shape[1] + b.shape[1]) + tuple(skip.shape[2:]), dtype=torch.float32, device=x.device, memory_format=cl)
    N.dcall(a, "sd_upcat_ndhwc_device", ctypes.c_void_p(a.data_ptr()), int(a.shape[1]), up, ctypes.c_void_p(b.data_ptr()), int(b.shape[1]), *S,
            ctypes.c_void_p(cat.data_ptr()))
    return _general_conv(conv, cat, kind, None, bn)


def _conv_bias_act(conv, x, kind):
    """conv + bias + (0 linear | 1 relu) of GPU inference by a hand-written kernel; None when the hand-written path does not apply
    (CPU, autograd, autocast: the caller runs the plain modules); raises UnsupportedLayer for a layer no kernel covers"""
    if not (_native_inference(x) and x.dtype == torch.float32):
        return None
    y = _hand_conv(conv, [(x, 0)], kind)
    if y is None:
        raise UnsupportedLayer(_layer_desc(conv, [(x, 0)]))
    return y


class ConvAct(nn.Sequential):
    """[conv, activation] (or [conv, batch-norm, activation]) with the Keras layer's parameter names.  GPU inference: one launch of a
    hand-written kernel (bias, folded batch-norm and linear / relu activation in its epilogue); everywhere else the plain Sequential."""

    def parts(self):
        """(conv, batch-norm or None, kind) with kind 0 linear / 1 relu / -1 another activation"""
        conv, bn, act = (self[0], None, self[1]) if len(self) == 2 else (self[0], self[1], self[2])
        return conv, bn, (0 if isinstance(act, nn.Identity) else (1 if isinstance(act, nn.ReLU) else -1))

    def forward(self, x):
        conv, bn, kind = self.parts()
        if not (_native_inference(x) and x.dtype == torch.float32):
            return super().forward(x)
        y = _hand_conv(conv, [(x, 0)], kind, bn=bn) if kind >= 0 else None
        if y is None:
            raise UnsupportedLayer(_layer_desc(conv, [(x, 0)]) + ("" if kind >= 0 else ", activation %s" % type(self[-1]).__name__))
        return y


def _conv(nd, cin, cout, k, act="relu", bias=True, batch_norm=False):
    k = tuple(k) if isinstance(k, (tuple, list)) else (k,) * nd
    Conv = nn.Conv2d if nd == 2 else nn.Conv3d
    assert all(kk % 2 == 1 for kk in k), "Keras 'same' padding restated for odd kernels only"
    conv = Conv(cin, cout, k, padding=tuple(kk // 2 for kk in k), bias=bias)
    if batch_norm:
        # csbdeep conv_block2/3 (csbdeep/internals/blocks.py): Conv -> BatchNormalization -> Activation; Keras defaults
        # epsilon = 1e-3, momentum 0.99 (inference uses the moving statistics)
        BN = nn.BatchNorm2d if nd == 2 else nn.BatchNorm3d
        return ConvAct(conv, BN(cout, eps=1e-3, momentum=0.01), _act(act))
    return ConvAct(conv, _act(act))


def max_pool(x, pool):
    """Keras MaxPooling ('valid', stride = pool).  GPU inference: the native one-pass channels-last kernel (sd_maxpool_ndhwc_device,
    64-bit indexing); everywhere else F.max_pool."""
    nd = x.dim() - 2
    pool = tuple(int(p) for p in pool)
    if _native_inference(x):
        cl = torch.channels_last if nd == 2 else torch.channels_last_3d
        if not (nd in (2, 3) and x.shape[0] == 1 and x.dtype == torch.float32 and x.shape[1] % 4 == 0):
            raise UnsupportedLayer("MaxPooling %s on %s %s" % (pool, x.dtype, tuple(x.shape)))
        if is_split16(x):
            # the pooled split16 tensor == split16 of the pooled f32 tensor (x -> (hi, lo') is monotone): same readers, same bits
            from ..lib import _native as N
            S = (1,) * (3 - nd) + tuple(int(v) for v in x.shape[2:])
            P = (1,) * (3 - nd) + pool
            out = torch.empty((1, x.shape[1]) + tuple(s // p for s, p in zip(x.shape[2:], pool)), dtype=torch.float32, device=x.device, memory_format=cl)
            if out.numel():
                N.dcall(x, "sd_maxpool_split16_ndhwc_device", ctypes.c_void_p(x.data_ptr()), int(x.shape[1]), *S, *P, ctypes.c_void_p(out.data_ptr()))
            return _tag_split16(out, getattr(x, "_sd_producer", None))
        if not (x.is_contiguous(memory_format=cl) and x.data_ptr() % 16 == 0):
            x = x.clone(memory_format=cl)
        from ..lib import _native as N
        S = (1,) * (3 - nd) + tuple(int(v) for v in x.shape[2:])
        P = (1,) * (3 - nd) + pool
        out = torch.empty((1, x.shape[1]) + tuple(s // p for s, p in zip(x.shape[2:], pool)), dtype=torch.float32, device=x.device, memory_format=cl)
        if out.numel():
            N.dcall(x, "sd_maxpool_ndhwc_device", ctypes.c_void_p(x.data_ptr()), int(x.shape[1]), *S, *P, ctypes.c_void_p(out.data_ptr()))
        return out
    return (F.max_pool2d if nd == 2 else F.max_pool3d)(x, pool)


class UNetBlock(nn.Module):
    """csbdeep unet_block(n_depth, n_filter_base, kernel_size, n_conv_per_depth, activation,
    last_activation, pool) without batch-norm/dropout (inference)."""

    def __init__(self, nd, cin, n_depth, n_filter_base, kernel_size, n_conv_per_depth, activation, last_activation, pool,
                 batch_norm=False):
        super().__init__()
        self.nd, self.n_depth, self.pool = nd, n_depth, tuple(pool)
        import functools
        _conv = functools.partial(globals()["_conv"], batch_norm=batch_norm)
        self.down = nn.ModuleList()
        c = cin
        for n in range(n_depth):
            convs = []
            for i in range(n_conv_per_depth):
                convs.append(_conv(nd, c, n_filter_base * 2 ** n, kernel_size, activation)); c = n_filter_base * 2 ** n
            self.down.append(nn.Sequential(*convs))
        mid = []
        for i in range(n_conv_per_depth - 1):
            mid.append(_conv(nd, c, n_filter_base * 2 ** n_depth, kernel_size, activation)); c = n_filter_base * 2 ** n_depth
        mid.append(_conv(nd, c, n_filter_base * 2 ** max(0, n_depth - 1), kernel_size, activation)); c = n_filter_base * 2 ** max(0, n_depth - 1)
        self.middle = nn.Sequential(*mid)
        self.up = nn.ModuleList()
        for n in reversed(range(n_depth)):
            c = c + n_filter_base * 2 ** n          # concat [up, skip]
            convs = []
            for i in range(n_conv_per_depth - 1):
                convs.append(_conv(nd, c, n_filter_base * 2 ** n, kernel_size, activation)); c = n_filter_base * 2 ** n
            convs.append(_conv(nd, c, n_filter_base * 2 ** max(0, n - 1), kernel_size, activation if n > 0 else last_activation))
            c = n_filter_base * 2 ** max(0, n - 1)
            self.up.append(nn.Sequential(*convs))
        self.out_channels = c

    def forward(self, x):
        skips = []
        for blk in self.down:
            x = blk(x)
            skips.append(x)
            x = max_pool(x, self.pool)
        x = self.middle(x)
        for blk, skip in zip(self.up, reversed(skips)):
            if _native_inference(x):
                # UpSampling + Concatenate + Conv (+ BN) + bias + activation as ONE launch: the up-sampled and the concatenated tensors
                # of the reference's graph are never written
                first = blk[0]
                conv0, bn, kind = first.parts()
                srcs = [(x, tuple(p == 2 for p in self.pool)), (skip, 0)]
                y = _hand_conv(conv0, srcs, kind, bn=bn) if (all(p in (1, 2) for p in self.pool) and kind >= 0) else None
                if y is None and kind >= 0:
                    y = _upcat_general(conv0, x, skip, self.pool, kind, bn)      # coverage path (channel counts not in 32-chunks)
                if y is None:
                    raise UnsupportedLayer("up-level " + _layer_desc(conv0, srcs) + ", pool %s" % (self.pool,))
                x = blk[1:](y)
                continue
            x = F.interpolate(x, scale_factor=tuple(float(p) for p in self.pool), mode="nearest")
            x = blk(torch.cat([x, skip], dim=1))
        return x


class ResNetBlock(nn.Module):
    """csbdeep resnet_block(n_filter, kernel_size, pool, n_conv_per_block, batch_norm, activation): first conv strided by
    `pool`, last conv linear, 1x1 strided projection on the shortcut when shape changes, add, activation.
    batch_norm=True (model3d.py:402-412 hands resnet_batch_norm through): every convolution of the block is bias-free
    (use_bias = not batch_norm, the shortcut projection included) and each BODY convolution -- the last one too, i.e. before the
    Add -- is followed by a BatchNormalization; the projection has none."""

    def __init__(self, nd, cin, n_filter, kernel_size, pool, n_conv_per_block, activation, batch_norm=False):
        super().__init__()
        Conv = nn.Conv2d if nd == 2 else nn.Conv3d
        BN = nn.BatchNorm2d if nd == 2 else nn.BatchNorm3d
        k = tuple(kernel_size)
        pad = tuple(kk // 2 for kk in k)
        self.pool = tuple(pool)
        self.k = k
        bias = not batch_norm
        bn = (lambda: [BN(n_filter, eps=1e-3, momentum=0.01)]) if batch_norm else (lambda: [])       # Keras defaults (see _conv)
        self.first = Conv(cin, n_filter, k, stride=self.pool, padding=0, bias=bias)   # Keras 'same' + stride pads asymmetrically
        layers = bn() + [_act(activation)]
        for _ in range(n_conv_per_block - 2):
            layers += [Conv(n_filter, n_filter, k, padding=pad, bias=bias)] + bn() + [_act(activation)]
        layers += [Conv(n_filter, n_filter, k, padding=pad, bias=bias)] + bn()
        self.body = nn.Sequential(*layers)
        self.proj = None
        if any(p != 1 for p in self.pool) or cin != n_filter:
            self.proj = Conv(cin, n_filter, (1,) * nd, stride=self.pool, bias=bias)
        self.act = _act(activation)

    def _same_pad(self, x):
        # TensorFlow 'SAME': total = max(k - s, 0) if n % s == 0 else max(k - n % s, 0); before = total // 2
        pads = []
        for d in reversed(range(len(self.k))):
            n, k, s = x.shape[2 + d], self.k[d], self.pool[d]
            total = max(k - s, 0) if n % s == 0 else max(k - n % s, 0)
            pads += [total // 2, total - total // 2]
        return F.pad(x, pads)

    def _stages(self):
        """[(conv, batch-norm or None, activation module or None)] in graph order: the strided first convolution, then the body's"""
        BNs = (nn.BatchNorm2d, nn.BatchNorm3d)
        out, cur = [], [self.first, None, None]
        for m in self.body:
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                out.append(tuple(cur))
                cur = [m, None, None]
            elif isinstance(m, BNs):
                cur[1] = m
            else:
                cur[2] = m
        out.append(tuple(cur))
        return out

    def _forward_hand(self, x):
        """the block on the hand-written kernels: strided first convolution (TensorFlow 'same' padding) with its activation, body
        convolutions, the strided 1x1 projection, and Add + Activation folded into the last convolution's epilogue; batch-norm layers
        folded into the (bias-free) kernels and a bias -- the last one before the Add, as the reference's graph has it"""
        kind = lambda a: 0 if isinstance(a, nn.Identity) else (1 if isinstance(a, nn.ReLU) else -1)
        stages = self._stages()
        if not (all(kind(a) >= 0 for _, _, a in stages[:-1]) and kind(self.act) >= 0):
            raise UnsupportedLayer("resnet_block activation %s" % type(self.act).__name__)
        if any(b is not None and b.training for _, b, _ in stages):
            raise UnsupportedLayer("resnet_block with batch-norm layers in training mode")

        def need(y, conv, src):
            if y is None:
                raise UnsupportedLayer("resnet_block " + _layer_desc(conv, [(src, 0)]))
            return y
        conv, bn, act = stages[0]
        y = need(_hand_conv(conv, [(x, 0)], kind(act), bn=bn, tf_same=True), conv, x)
        sc = x
        if self.proj is not None:
            sc = need(_hand_conv(self.proj, [(x, 0)], 0, tf_same=True), self.proj, x)
        for conv, bn, act in stages[1:]:
            last = act is None
            y = need(_hand_conv(conv, [(y, 0)], kind(self.act) if last else kind(act), res=sc if last else None, bn=bn), conv, y)
        return y

    def forward(self, x):
        if _native_inference(x):
            return self._forward_hand(x)
        y = self.body(self.first(self._same_pad(x)))
        if self.proj is not None:
            x = self.proj(x)
        return self.act(x + y)


class StarDistNet(nn.Module):
    """input (N,C,...) -> prob (N,1,...), dist (N,n_rays,...)[, prob_class (N,n_classes+1,...)]"""

    def __init__(self, config):
        super().__init__()
        cfg = config
        nd = cfg.n_dim
        self.nd = nd
        self.pre = nn.ModuleList()
        c = cfg.n_channel_in
        grid = np.asarray(cfg.grid)
        if cfg.backbone == "unet":
            pooled = np.ones(nd, int)
            while tuple(pooled) != tuple(grid):                       # model2d.py:317-325
                pool = 1 + (grid > pooled)
                pooled = pooled * pool
                convs = []
                for _ in range(cfg.unet_n_conv_per_depth):
                    convs.append(_conv(nd, c, cfg.unet_n_filter_base, cfg.unet_kernel_size, cfg.unet_activation)); c = cfg.unet_n_filter_base
                self.pre.append(nn.ModuleDict(dict(convs=nn.Sequential(*convs))))
                self.pre[-1].pool = tuple(int(p) for p in pool)
            self.backbone = UNetBlock(nd, c, cfg.unet_n_depth, cfg.unet_n_filter_base, cfg.unet_kernel_size,
                                      cfg.unet_n_conv_per_depth, cfg.unet_activation, cfg.unet_last_activation,
                                      cfg.unet_pool, cfg.unet_batch_norm)
            c = self.backbone.out_channels
            n_after, k_after, act_after = cfg.net_conv_after_unet, cfg.unet_kernel_size, cfg.unet_activation
        elif cfg.backbone == "resnet":                                 # model3d.py:402-447
            n_filter = cfg.resnet_n_filter_base
            blocks = [_conv(nd, c, n_filter, (7,) * nd, None),        # linear (no activation) model3d.py:416-417
                      _conv(nd, n_filter, n_filter, (3,) * nd, None)]
            c = n_filter
            pooled = np.ones(nd, int)
            for n in range(cfg.resnet_n_blocks):
                pool = 1 + (grid > pooled)
                pooled = pooled * pool
                if any(p > 1 for p in pool):
                    n_filter *= 2
                blocks.append(ResNetBlock(nd, c, n_filter, cfg.resnet_kernel_size, tuple(int(p) for p in pool),
                                          cfg.resnet_n_conv_per_block, cfg.resnet_activation, bool(getattr(cfg, "resnet_batch_norm", False))))
                c = n_filter
            self.backbone = nn.Sequential(*blocks)
            n_after, k_after, act_after = cfg.net_conv_after_resnet, cfg.resnet_kernel_size, cfg.resnet_activation
        else:
            raise ValueError(cfg.backbone)
        self.features = _conv(nd, c, n_after, k_after, act_after) if n_after > 0 else nn.Identity()
        cf = n_after if n_after > 0 else c
        Conv = nn.Conv2d if nd == 2 else nn.Conv3d
        self.prob = Conv(cf, 1, (1,) * nd)
        self.dist = Conv(cf, cfg.n_rays, (1,) * nd)
        self.n_classes = cfg.n_classes
        if cfg.n_classes is not None:
            self.features_class = _conv(nd, c, n_after, k_after, act_after) if n_after > 0 else nn.Identity()
            self.prob_class = Conv(cf, cfg.n_classes + 1, (1,) * nd)
        self._plan_split16()

    def _plan_split16(self):
        """mark (conv.__dict__["_sd_split_out"]) the layers whose output travels as a split16 tensor (see split16_enabled): 3x3(x3)
        stride-1 layers with relu / linear activation over 32-channel chunks (or the one-channel first layer with 32 outputs) whose EVERY
        reader is such a layer, directly or through a max-pooling.  U-Net backbones only (a ResNet block's shortcut and strided layers
        read f32)."""
        nd = self.nd

        def layer_ok(m, as_producer):
            if not isinstance(m, ConvAct):
                return False
            conv, _, kind = m.parts()
            Conv = nn.Conv2d if nd == 2 else nn.Conv3d
            if not (isinstance(conv, Conv) and kind >= 0 and tuple(conv.kernel_size) == (3,) * nd and tuple(conv.stride) == (1,) * nd
                    and tuple(conv.padding) == (1,) * nd and tuple(conv.dilation) == (1,) * nd and conv.groups == 1):
                return False
            ci, co = conv.in_channels, conv.out_channels
            if as_producer:
                return co % 32 == 0 and ((ci % 32 == 0 and ci <= _MAX_CHUNK_CHANNELS) or (ci == 1 and co == 32))
            return ci % 32 == 0 and ci <= _MAX_CHUNK_CHANNELS and co % 32 == 0

        readers = {}

        def feed(prods, reader):
            for p in prods:
                readers.setdefault(p, []).append(reader)

        def run(seq, cur):
            for m in seq:
                feed(cur, m)
                cur = [m]
            return cur
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                m.__dict__["_sd_split_out"] = False
        if not isinstance(self.backbone, UNetBlock):
            return
        cur = []
        for st in self.pre:
            cur = run(st["convs"], cur)
        bb = self.backbone
        if not all(p in (1, 2) for p in bb.pool):
            return
        skips = []
        for blk in bb.down:
            cur = run(blk, cur)
            skips.append(cur)
        cur = run(bb.middle, cur)
        for blk, skip in zip(bb.up, reversed(skips)):
            cur = run(blk, cur + skip)
        feed(cur, self.features)
        if self.n_classes is not None:
            feed(cur, self.features_class)
        for p, rs in readers.items():
            if layer_ok(p, True) and all(layer_ok(r, False) for r in rs):
                p.parts()[0].__dict__["_sd_split_out"] = True

    def _heads(self, base):
        """the plain graph: features conv, prob (Conv 1x1 + sigmoid), dist (Conv 1x1)[, class head] -- on the GPU every convolution is
        a hand-written kernel (64-bit indexing: no slabs whatever the volume), elsewhere the torch modules"""
        f = self.features(base)
        p = _conv_bias_act(self.prob, f, 0)
        prob = torch.sigmoid_(p) if p is not None else torch.sigmoid(self.prob(f))
        dist = _conv_bias_act(self.dist, f, 0)
        if dist is None:
            dist = self.dist(f)
        if self.n_classes is not None:
            return prob, dist, self._class_head(base)
        return prob, dist

    def _class_head(self, base):
        """softmax(prob_class(features_class(base))): the multi-class head (model2d.py:339-347, model3d.py:443-452)"""
        f = self.features_class(base)
        y = _conv_bias_act(self.prob_class, f, 0)
        return torch.softmax(y if y is not None else self.prob_class(f), dim=1)

    # ---- fused heads (GPU inference) ----------------------------------------------------------------------------------------
    # features conv -> ONE pass doing the probability head (sd_bias_act_dot_device) -> distance head as an fp32-MFMA
    # GEMM over the rows asked for (sd_head_rows_device): every pixel for the dense prediction, or -- sparse_head=True -- none here:
    # the caller selects the candidate pixels from the probabilities and evaluates the distance head on those rows only
    # (StarDistBase._predict_sparse_generator), so the dense n_rays-channel tensor of the reference's predict_sparse
    # (base.py:553-610: full prediction, then masking) is never written.  Both paths run the same kernels with a fixed summation
    # order per output, hence agree bit for bit.
    fused_heads = True                        # False: plain graph on the GPU as well (statistics hooks, A/B timing)

    def _fused_heads_ok(self, base):
        f = self.features
        if not (self.fused_heads and _native_inference(base) and base.shape[0] == 1 and base.dtype == torch.float32
                and isinstance(f, ConvAct) and len(f) == 2 and f[0].bias is not None):
            return False
        kind = 0 if isinstance(f[1], nn.Identity) else (1 if isinstance(f[1], nn.ReLU) else -1)
        C, R = f[0].out_channels, self.dist.out_channels
        return kind >= 0 and C in (32, 64, 128, 256) and R <= 128 and C * (((R + 31) // 32) * 32 + 1) * 4 <= 64 * 1024

    # The sparse path without the dense feature tensor (round 6): the features layer runs with the probability head fused and WITHOUT its
    # store (head_mode "sparse_lazy": forward returns (prob, backbone output)); dist_rows then evaluates the layer on the candidate rows
    # (conv_rows, bit-identical to the dense layer) and the distance head on those.  2 GiB (2048^2) / 8.6 GB (256^3) are never written.
    # Measured (round 6): the row kernel costs 0.25 ms at 4e5 (2D) / 1.7e5 (3D) candidates (half of it the weight blocks every workgroup
    # stages, 0.1 ms the gathers), the dense store it replaces 0.15 ms at 2048^2 (2.1 GB) and 0.3 ms at 256^3 (8.6 GB): the dense tensor is
    # kept while it is small, from lazy_features_min_bytes on (and for the blocks of a sharded input: 10 GB / 90 GB each) it is not written.
    lazy_features = True                      # False (or STARDIST_AMD_LAZY_FEATURES=0): the dense feature tensor always
    lazy_features_min_bytes = 4 << 30

    def _lazy_ok(self, base):
        import os
        f = self.features
        if not (self.lazy_features and os.environ.get("STARDIST_AMD_LAZY_FEATURES", "1") != "0" and conv_mode() == "f16x3" and isinstance(f, ConvAct)):
            return False
        conv, bn, kind = f.parts()
        nd = self.nd
        if int(np.prod(base.shape[2:])) * conv.out_channels * 4 < self.lazy_features_min_bytes:
            return False
        return (bn is None and kind >= 0 and tuple(conv.kernel_size) == (3,) * nd and tuple(conv.stride) == (1,) * nd and tuple(conv.padding) == (1,) * nd
                and tuple(conv.dilation) == (1,) * nd and conv.groups == 1 and conv.in_channels % 32 == 0 and 0 < conv.in_channels <= _MAX_CHUNK_CHANNELS
                and conv.out_channels % 32 == 0 and conv.__dict__.get("_sd_force_form") != "bf16x6" and base.shape[1] == conv.in_channels)

    def feature_rows(self, base_cl, rows):
        """features (after bias + activation) of the pixels `rows` from the backbone output given as its channels-last view (..., C_in)"""
        nd = base_cl.dim() - 1
        x = base_cl.permute(*([nd] + list(range(nd)))).unsqueeze(0)          # back to (1, C, *spatial): the channels-last tensor itself
        if getattr(self, "_lazy_split16", False):
            x._sd_split16 = True
        conv, _, kind = self.features.parts()
        return conv_rows(conv, x, kind, rows)

    def dist_rows(self, feat, rows, clamp_min, lazy=False, order=None):
        """distance head on rows of the channels-last feature matrix feat (n_pix, C): (len(rows), n_rays); rows None = all.
        lazy: `feat` is the BACKBONE output (head_mode "sparse_lazy") -- the features of the pixels `rows` are evaluated first (give the
        rows in SPATIAL order: the gathered 3x3 neighbourhoods then share cache lines) and the head runs on them, in the order `order`
        (indices into rows; None: as they are)"""
        from ..lib import _native as N
        if lazy:
            feat, rows = self.feature_rows(feat, rows), order
        C, R = feat.shape[-1], self.dist.out_channels
        feat = feat.reshape(-1, C)
        n = feat.shape[0] if rows is None else int(rows.shape[0])
        out = torch.empty((n, R), dtype=torch.float32, device=feat.device)
        if n:
            w = self.dist.weight.detach().reshape(R, C).contiguous()
            b = self.dist.bias
            N.dcall(feat, "sd_head_rows_device", ctypes.c_void_p(feat.data_ptr()), C, ctypes.c_void_p(rows.data_ptr() if rows is not None else None),
                    n, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr() if b is not None else None), R, float(clamp_min),
                    ctypes.c_void_p(out.data_ptr()))
        return out

    def _heads_fused(self, base, sparse_head):
        from ..lib import _native as N
        conv, act = self.features[0], self.features[1]
        kind = 1 if isinstance(act, nn.ReLU) else 0
        nd = base.dim() - 2
        S = tuple(base.shape[2:])
        C = conv.out_channels
        wp = self.prob.weight.detach().reshape(-1).contiguous()
        bp = self.prob.bias
        prob = torch.empty((1, 1) + S, dtype=torch.float32, device=base.device)
        holder = []
        # features conv with bias + activation fused (64-bit indexing: no slabs); the split-fp16 kernel also takes the probability head's
        # dot product over each workgroup's 32 channels while the tile is in registers
        lazy = bool(sparse_head) and self._lazy_ok(base)
        feat = _hand_conv(conv, [(base, 0)], kind, dot=(wp, holder), no_store=lazy)
        if feat is None:
            raise UnsupportedLayer("features " + _layer_desc(conv, [(base, 0)]))
        self._lazy_now = feat is NO_STORE
        if feat is NO_STORE:
            self._lazy_split16 = is_split16(base)
            feat = base                       # what the caller gets in place of the features: dist_rows(..., lazy=True) works from it
        if holder:
            # ... the per-lane terms -> probabilities, bit-identical to sd_bias_act_dot_device on the same features, which are not re-read
            N.dcall(feat, "sd_dot_combine_device", ctypes.c_void_p(holder[0].data_ptr()), C // 32, int(np.prod(S)),
                    ctypes.c_void_p(bp.data_ptr() if bp is not None else None), 1, ctypes.c_void_p(prob.data_ptr()))
        else:
            # ... then the probability head alone: one read of the features
            N.dcall(feat, "sd_bias_act_dot_device", ctypes.c_void_p(feat.data_ptr()), None, None, int(np.prod(S)), C, 0, ctypes.c_void_p(wp.data_ptr()),
                    ctypes.c_void_p(bp.data_ptr() if bp is not None else None), 1, ctypes.c_void_p(prob.data_ptr()))
        if sparse_head:
            return prob, feat
        R = self.dist.out_channels
        dist = self.dist_rows(feat.permute(*([0] + list(range(2, nd + 2)) + [1])), None, float("-inf"))
        return prob, dist.view((1,) + S + (R,)).permute(*([0, nd + 1] + list(range(1, nd + 1))))

    def forward(self, x, sparse_head=False):
        """(prob, dist[, prob_class]); with sparse_head=True and the fused heads available: (prob, features[, prob_class]) and
        self.head_mode == "sparse" -- the distance head is then evaluated by the caller on the rows it selects (dist_rows)"""
        for st in self.pre:
            x = max_pool(st["convs"](x), st.pool)
        base = self.backbone(x)
        self.head_mode = "dense"
        if self._fused_heads_ok(base):
            out = self._heads_fused(base, sparse_head)
            if sparse_head:
                self.head_mode = "sparse_lazy" if getattr(self, "_lazy_now", False) else "sparse"
            if self.n_classes is not None:
                out = tuple(out) + (self._class_head(base),)
            return tuple(out)
        return self._heads(base)


def init_he_normal_(net, seed=0):
    """Seeded He-normal kernels / small biases: real weights are not available offline
    (.MISSING_LARGE_BLOBS); throughput is weight independent."""
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            fan_in = m.in_channels * int(np.prod(m.kernel_size))
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * float(np.sqrt(2.0 / fan_in)))
                if m.bias is not None:
                    m.bias.zero_()
    return net


def conv_macs_per_input_pixel(net, cfg):
    """analytic multiply-accumulates per INPUT pixel of the conv stack (for the MFMA roofline): counted by forward hooks on a CPU copy
    of the modules (the GPU inference path does not go through the modules' forward)."""
    import copy
    nd = cfg.n_dim
    size = 64 if nd == 2 else 32
    shape = tuple(size * g for g in cfg.grid)
    macs = [0.0]
    hooks = []
    net = copy.deepcopy(net).cpu()

    def hook(m, inp, out):
        k = float(np.prod(m.kernel_size))
        macs[0] += out.numel() / out.shape[0] * (m.in_channels * k)
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        net(torch.zeros((1, cfg.n_channel_in) + shape))
    for h in hooks:
        h.remove()
    return macs[0] / float(np.prod(shape))
