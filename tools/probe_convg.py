"""Time the layers of the ResNet backbone (3D_demo topology, stardist/models/model3d.py:400-447) and the other general-kernel layers
on the GPU: HIP-event time per launch and TFLOP/s (2 * MACs), csrc/conv_general.hip vs csrc/conv3x3.hip where both apply.
usage: python tools/probe_convg.py [--size 64,256,256] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from stardist_amd.models import unet as U  # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="64,256,256")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    S = tuple(int(v) for v in a.size.split(","))
    dev = torch.device("cuda:0")
    cl3, cl2 = torch.channels_last_3d, torch.channels_last
    half = (S[0], S[1] // 2, S[2] // 2)
    layers = [  # name, nd, c_in, c_out, k, stride, spatial, tf_same, residual
        ("stem 7x7x7 1->32", 3, 1, 32, 7, (1, 1, 1), S, False, False),
        ("stem 3x3x3 32->32 (conv3x3.hip)", 3, 32, 32, 3, (1, 1, 1), S, False, False),
        ("block0 first 3x3x3 s(1,2,2) 32->64", 3, 32, 64, 3, (1, 2, 2), S, True, False),
        ("block0 proj 1x1x1 s(1,2,2) 32->64", 3, 32, 64, 1, (1, 2, 2), S, True, False),
        ("block body 3x3x3 64->64 + res (conv3x3.hip)", 3, 64, 64, 3, (1, 1, 1), half, False, True),
        ("features 3x3x3 64->128 (conv3x3.hip)", 3, 64, 128, 3, (1, 1, 1), half, False, False),
        ("H&E first layer 3x3 3->32 2048^2", 2, 3, 32, 3, (1, 1), (2048, 2048), False, False),
        ("prob_class 1x1 128->4 1024^2", 2, 128, 4, 1, (1, 1), (1024, 1024), False, False),
    ]
    with torch.no_grad():
        for name, nd, ci, co, k, st, sp, tf_same, with_res in layers:
            Conv = torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d
            conv = Conv(ci, co, k, stride=st, padding=0 if tf_same else k // 2).to(dev)
            cl = cl2 if nd == 2 else cl3
            x = torch.randn((1, ci) + tuple(sp), device=dev).contiguous(memory_format=cl)
            osp = tuple(-(-n // s) for n, s in zip(sp, st))
            res = torch.randn((1, co) + osp, device=dev).contiguous(memory_format=cl) if with_res else None
            y = U._hand_conv(conv, [(x, 0)], 1, res=res, tf_same=tf_same)
            assert y is not None, name
            ms = timed(lambda: U._hand_conv(conv, [(x, 0)], 1, res=res, tf_same=tf_same), a.reps)
            flop = 2.0 * np.prod(osp) * co * ci * k ** nd
            line = "%-48s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flop / ms / 1e9)
            if k == 3 and all(s == 1 for s in st) and ci % 32 == 0:      # the same layer forced through the general kernel
                yg = U._general_conv(conv, x, 1, res)
                msg = timed(lambda: U._general_conv(conv, x, 1, res), a.reps)
                line += "   | general kernel %8.3f ms %7.1f TFLOP/s, max |diff| %.2g" % (msg, flop / msg / 1e9, float((yg - y).abs().max()))
            print(line, flush=True)


if __name__ == "__main__":
    main()
