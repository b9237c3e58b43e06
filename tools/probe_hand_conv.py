"""Per-layer A/B of the hand-written convolution (csrc/conv3x3.hip) against the same layer through MIOpen (+ the separate
interpolate / cat / bias+activation passes it needs): HIP-event times and TFLOP/s for every layer shape of the two bench networks.
usage: python tools/probe_hand_conv.py [--size 2048] [--size3d 256] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stardist_amd  # noqa: E402,F401
from stardist_amd.models import unet as U  # noqa: E402


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


args_split = True
skip_lib = False


def layer(nd, shape, chans, cout, reps, dev):
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    cin = sum(c for c, _ in chans)
    conv = (torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d)(cin, cout, 3, padding=1).to(dev)
    srcs = [(torch.randn((1, c) + tuple(s >> u for s in shape), device=dev).contiguous(memory_format=cl), (u,) * nd) for c, u in chans]
    flops = 2.0 * cin * cout * 3 ** nd * float(np.prod(shape))

    def hand():
        return U._hand_conv(conv, srcs, 1)

    def split():
        os.environ["STARDIST_AMD_CONV"] = "bf16x6"
        try:
            return U._hand_conv(conv, srcs, 1)
        finally:
            os.environ["STARDIST_AMD_CONV"] = "hand"

    def lib():
        xs = [F.interpolate(t, scale_factor=2.0, mode="nearest") if any(u) else t for t, u in srcs]
        x = xs[0] if len(xs) == 1 else torch.cat(xs, 1)
        os.environ["STARDIST_AMD_CONV"] = "miopen"
        try:
            return U._conv_bias_act(conv, x, 1)
        finally:
            os.environ["STARDIST_AMD_CONV"] = "hand"
    with torch.no_grad():
        th = timeit(hand, reps)
        ts = timeit(split, reps) if cin >= 32 and args_split else float("nan")
        try:
            tl = float("nan") if skip_lib else timeit(lib, reps)
        except Exception as e:          # e.g. int32 index limit of the library on the biggest 3D layer
            tl = float("nan"); print("   library path failed:", repr(e)[:100])
        err = float("nan")
        if np.prod(shape) * cout < 2 ** 29 and tl == tl:
            err = float((hand() - lib()).abs().max())
    print("%dD %-16s %-22s -> %3d : hand %8.3f ms %6.1f TF/s | miopen+glue %8.3f ms %6.1f TF/s | x%.2f  maxdiff %.2e | bf16x6 %8.3f ms %6.1f TF/s-equivalent"
          % (nd, "x".join(map(str, shape)), "+".join("%d%s" % (c, "^" if u else "") for c, u in chans), cout, th, flops / th / 1e9, tl, flops / tl / 1e9,
             tl / th, err, ts, flops / ts / 1e9), flush=True)
    return th, tl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--size3d", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-lib", action="store_true", help="skip the MIOpen comparison (saves its find-mode warm-up)")
    a = ap.parse_args()
    global skip_lib
    skip_lib = a.no_lib
    dev = torch.device("cuda:0")
    tot = [0.0, 0.0]
    if a.size:
        s = a.size
        L2 = [((s, s), [(1, 0)], 32), ((s, s), [(32, 0)], 32), ((s // 2,) * 2, [(32, 0)], 64), ((s // 2,) * 2, [(64, 0)], 64),
              ((s // 4,) * 2, [(64, 0)], 128), ((s // 4,) * 2, [(128, 0)], 128), ((s // 8,) * 2, [(128, 0)], 256), ((s // 8,) * 2, [(256, 0)], 128),
              ((s // 4,) * 2, [(128, 1), (128, 0)], 128), ((s // 4,) * 2, [(128, 0)], 64), ((s // 2,) * 2, [(64, 1), (64, 0)], 64),
              ((s // 2,) * 2, [(64, 0)], 32), ((s, s), [(32, 1), (32, 0)], 32), ((s, s), [(32, 0)], 32), ((s, s), [(32, 0)], 128)]
        for shape, ch, co in L2:
            th, tl = layer(2, shape, ch, co, a.reps, dev); tot[0] += th; tot[1] += tl
        print("2D network conv layers: hand %.2f ms, miopen+glue %.2f ms" % tuple(tot), flush=True)
    if a.size3d:
        s = a.size3d
        tot = [0.0, 0.0]
        L3 = [((s,) * 3, [(1, 0)], 32), ((s,) * 3, [(32, 0)], 32), ((s // 2,) * 3, [(32, 0)], 64), ((s // 2,) * 3, [(64, 0)], 64),
              ((s // 4,) * 3, [(64, 0)], 128), ((s // 4,) * 3, [(128, 0)], 64), ((s // 2,) * 3, [(64, 1), (64, 0)], 64), ((s // 2,) * 3, [(64, 0)], 32),
              ((s,) * 3, [(32, 1), (32, 0)], 32), ((s,) * 3, [(32, 0)], 32), ((s,) * 3, [(32, 0)], 128)]
        for shape, ch, co in L3:
            th, tl = layer(3, shape, ch, co, max(2, a.reps // 2), dev); tot[0] += th; tot[1] += tl
            torch.cuda.empty_cache()
        print("3D network conv layers: hand %.2f ms, miopen+glue %.2f ms" % tuple(tot), flush=True)


if __name__ == "__main__":
    main()
