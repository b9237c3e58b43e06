"""Per-layer timing of the hand-written 3x3 / 3x3x3 convolution kernels on every layer shape of the two bench networks: HIP-event
times and algorithmic TFLOP/s (2 x MACs / time) for the exact-f32 kernel (csrc/conv3x3.hip), the six-product bf16 form
(conv3x3_bf16.hip) and the three-product fp16 form (conv3x3_f16.hip) at two and at one workgroup per CU, plus the largest
deviation of each split form from the exact kernel.
usage: python tools/probe_hand_conv.py [--size 2048] [--size3d 256] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: E402,F401  (STARDIST_AMD_PROBE_LIB: a variant build of the library)
import stardist_amd  # noqa: E402,F401
from stardist_amd.lib import _native as N  # noqa: E402
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


MODES = [("hand", "hand", 2), ("bf16x6", "bf16x6", 2), ("f16x3", "f16x3", 2), ("f16x3/1wg", "f16x3", 1), ("f16x3/s16", "f16x3", 2)]
if os.environ.get("PROBE_F16_ONLY") == "1":
    MODES = [m for m in MODES if m[0] in ("f16x3", "f16x3/s16")]


def layer(nd, shape, chans, cout, reps, dev, totals):
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    cin = sum(c for c, _ in chans)
    conv = (torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d)(cin, cout, 3, padding=1).to(dev)
    srcs = [(torch.randn((1, c) + tuple(s >> u for s in shape), device=dev).contiguous(memory_format=cl), (u,) * nd) for c, u in chans]
    flops = 2.0 * cin * cout * 3 ** nd * float(np.prod(shape))
    line = "%dD %-14s %-16s -> %3d :" % (nd, "x".join(map(str, shape)), "+".join("%d%s" % (c, "^" if u else "") for c, u in chans), cout)
    ref = None
    with torch.no_grad():
        for tag, mode, wgs in MODES:
            if cin < 32 and tag not in ("hand", "f16x3/s16"):
                continue
            N.check(N.lib().sd_set_option(b"conv_f16_workgroups_per_cu", wgs))
            use = srcs
            if tag.endswith("/s16"):
                # split16 tensors on both sides (the features layer, c_out 128, writes f32 for the heads)
                if cin < 32:
                    conv.__dict__["_sd_split_out"] = True
                else:
                    use = [(U.split16_pack(t), up) for t, up in srcs]
                    conv.__dict__["_sd_split_out"] = cout != 128
            with U.force_conv_mode(mode):
                t = timeit(lambda: U._hand_conv(conv, use, 1), reps)
                y = U._hand_conv(conv, use, 1)
            conv.__dict__["_sd_split_out"] = False
            y = U.split16_unpack(y)
            del use
            if ref is None:
                ref, dev_ = y, float("nan")
            else:
                dev_ = float((y - ref).abs().max() / ref.abs().max())
            del y
            totals[tag] = totals.get(tag, 0.0) + t
            line += "  %s %7.3f ms %6.1f TF/s" % (tag, t, flops / t / 1e9) + ("" if tag == "hand" else " (d %.1e)" % dev_)
    N.check(N.lib().sd_set_option(b"conv_f16_workgroups_per_cu", 2))
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--size3d", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.size:
        s = a.size
        tot = {}
        L2 = [((s, s), [(1, 0)], 32), ((s, s), [(32, 0)], 32), ((s // 2,) * 2, [(32, 0)], 64), ((s // 2,) * 2, [(64, 0)], 64),
              ((s // 4,) * 2, [(64, 0)], 128), ((s // 4,) * 2, [(128, 0)], 128), ((s // 8,) * 2, [(128, 0)], 256), ((s // 8,) * 2, [(256, 0)], 128),
              ((s // 4,) * 2, [(128, 1), (128, 0)], 128), ((s // 4,) * 2, [(128, 0)], 64), ((s // 2,) * 2, [(64, 1), (64, 0)], 64),
              ((s // 2,) * 2, [(64, 0)], 32), ((s, s), [(32, 1), (32, 0)], 32), ((s, s), [(32, 0)], 32), ((s, s), [(32, 0)], 128)]
        for shape, ch, co in L2:
            layer(2, shape, ch, co, a.reps, dev, tot)
        print("2D network conv layers (the one-channel first layer counted once, under 'hand'): " + ", ".join("%s %.2f ms" % kv for kv in tot.items()), flush=True)
    if a.size3d:
        s = a.size3d
        tot = {}
        L3 = [((s,) * 3, [(1, 0)], 32), ((s,) * 3, [(32, 0)], 32), ((s // 2,) * 3, [(32, 0)], 64), ((s // 2,) * 3, [(64, 0)], 64),
              ((s // 4,) * 3, [(64, 0)], 128), ((s // 4,) * 3, [(128, 0)], 64), ((s // 2,) * 3, [(64, 1), (64, 0)], 64), ((s // 2,) * 3, [(64, 0)], 32),
              ((s,) * 3, [(32, 1), (32, 0)], 32), ((s,) * 3, [(32, 0)], 32), ((s,) * 3, [(32, 0)], 128)]
        for shape, ch, co in L3:
            layer(3, shape, ch, co, max(2, a.reps // 2), dev, tot)
            torch.cuda.empty_cache()
        print("3D network conv layers: " + ", ".join("%s %.2f ms" % kv for kv in tot.items()), flush=True)


if __name__ == "__main__":
    main()
