"""CPU study of split-operand convolution arithmetic against a float64 evaluation of the same network (no GPU needed).

Every 3x3(x3) layer over 32-channel chunks (the layers the split kernels run) is evaluated with its f32 operands written as a sum of
low-precision terms and only the listed cross products kept; each product is exact in f32 (<= 22 significant bits), the sums are
taken in float64 here (the f32 accumulation error is the same for every kernel form and is measured on the GPU).  Activations are
rounded to f32 between layers, like the tensors in HBM.

  bf16x6   x = hi + mid + lo (bf16), six leading products          (csrc/conv3x3_bf16.hip, the round-3 default)
  bf16x3   x = hi + mid,             hi*hi + hi*mid + mid*hi
  f16x3    x = hi + lo * 2^-11 (fp16, lo scaled so it stays normal), hi*hi + (hi*lo' + lo'*hi) * 2^-11
  f16x3-ftz  the same with fp16 subnormal operands flushed to zero (worst case for a matrix pipe that flushes)
  f32      operands as they are (the float64 sum of f32 products: the floor for any f32-operand kernel)

usage: python tools/split_study.py [2d] [3d] [resnet]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import synth
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D


def bf16(x):
    return x.float().bfloat16().double()


def f16(x, ftz=False):
    h = x.float().half()
    if ftz:
        h = torch.where(h.abs() < 6.103515625e-05, torch.zeros_like(h), h)
    return h.double()


def terms(x, mode):
    """list of (term tensor float64, weight class) for operand x (float64 holding f32 values)"""
    if mode == "f32":
        return [x]
    if mode.startswith("bf16"):
        hi = bf16(x); mid = bf16(x - hi); lo = bf16(x - hi - mid)
        return [hi, mid, lo]
    ftz = mode.endswith("ftz")
    hi = f16(x, ftz)
    lo = f16((x - hi) * 2048.0, ftz) / 2048.0
    return [hi, lo]


PAIRS = {"f32": [(0, 0)], "bf16x6": [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)], "bf16x3": [(0, 0), (0, 1), (1, 0)],
         "f16x3": [(0, 0), (0, 1), (1, 0)], "f16x3-ftz": [(0, 0), (0, 1), (1, 0)], "f16x4": [(0, 0), (0, 1), (1, 0), (1, 1)]}


def patch(net, mode, stats):
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            k = tuple(m.kernel_size)
            split = all(v == 3 for v in k) and m.in_channels % 32 == 0 and m.out_channels % 32 == 0 and all(s == 1 for s in m.stride)

            def fwd(x, m=m, split=split):
                conv = F.conv2d if isinstance(m, nn.Conv2d) else F.conv3d
                x = x.float().double()                     # tensors live in HBM as f32
                w = m.weight.float().double()
                if mode == "f64" or not split:
                    y = conv(x, w, None, m.stride, m.padding)
                else:
                    stats["max_act"] = max(stats.get("max_act", 0.0), float(x.abs().max()))
                    nz = x[x != 0].abs()
                    if nz.numel():
                        stats["frac_act_sub"] = max(stats.get("frac_act_sub", 0.0), float((nz < 6.1e-5).double().mean()))
                    tx, tw = terms(x, mode), terms(w, mode)
                    y = 0
                    for a, b in reversed(PAIRS[mode]):
                        y = y + conv(tx[a], tw[b], None, m.stride, m.padding)
                if m.bias is not None:
                    y = y + m.bias.double().view((1, -1) + (1,) * (y.dim() - 2))
                return y
            m.forward = fwd


def evaluate(tag, make, img):
    ref = None
    for mode in ("f64", "f32", "bf16x6", "bf16x3", "f16x3", "f16x3-ftz", "f16x4"):
        model = make()
        net = model.net.double()
        stats = {}
        patch(net, mode, stats)
        x = torch.from_numpy(img).double()
        x = x[None, None] if x.dim() == model.config.n_dim else x.movedim(-1, 0)[None]
        with torch.no_grad():
            out = net(x)
        prob, dist = out[0].numpy(), out[1].numpy()
        if ref is None:
            ref = (prob, dist)
            print("%s: float64 reference, prob in [%.3g, %.3g], |dist| mean %.3g" % (tag, prob.min(), prob.max(), np.abs(dist).mean()), flush=True)
            continue
        dp = np.abs(prob - ref[0]).max()
        dd = (np.abs(dist - ref[1]) / np.maximum(np.abs(ref[1]), 1e-3)).max()
        dda = np.abs(dist - ref[1]).max() / np.abs(ref[1]).max()
        print("  %-10s max|dprob| %.3g   max rel|ddist| %.3g   max|ddist|/scale %.3g   %s" % (
            mode, dp, dd, dda, " ".join("%s=%.3g" % kv for kv in sorted(stats.items()))), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    which = sys.argv[1:] or ["2d", "3d"]
    import bench
    if "2d" in which:
        img = synth.s2d_nuclei_image(256, 256, seed=1)

        def make2():
            m = StarDist2D(Config2D(n_rays=32), basedir=None, device="cpu", seed=0)
            bench.calibrate_heads(m, torch.from_numpy(img))
            return m
        evaluate("U-Net 2D 256^2", make2, img)
    if "3d" in which:
        img3 = synth.s3d_nuclei_image(48, seed=1)

        def make3():
            m = StarDist3D(Config3D(rays=96), basedir=None, device="cpu", seed=0)
            bench.calibrate_heads(m, torch.from_numpy(img3), frac=0.02, radius=8.5, noise=0.03)
            return m
        evaluate("U-Net 3D 48^3", make3, img3)
    if "resnet" in which:
        img3 = synth.s3d_nuclei_image(32, seed=2)

        def make4():
            m = StarDist3D(Config3D(rays=96, backbone="resnet", grid=(1, 2, 2)), basedir=None, device="cpu", seed=0)
            bench.calibrate_heads(m, torch.from_numpy(img3), frac=0.02, radius=8.5, noise=0.03)
            return m
        evaluate("ResNet 3D 32^3", make4, img3)
