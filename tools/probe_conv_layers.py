"""Per-layer convolution times of the bench networks (HIP events around every Conv module) and which aten convolution
back-end each call takes (torch.profiler with shapes): finds layers that drop to PyTorch's vol2col/im2col + GEMM fallback.
usage: python tools/probe_conv_layers.py [--size3d 256] [--size 2048] [--skip-2d]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stardist_amd  # noqa: E402,F401
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D  # noqa: E402


def layer_times(model, x, reps=3):
    net = model.net
    recs = {}
    hooks = []
    for name, mod in net.named_modules():
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.Conv3d)):
            def pre(m, inp, name=name):
                e = torch.cuda.Event(enable_timing=True); e.record()
                recs.setdefault(name, []).append([e, None, tuple(inp[0].shape)])
            def post(m, inp, out, name=name):
                e = torch.cuda.Event(enable_timing=True); e.record()
                recs[name][-1][1] = e
            hooks.append(mod.register_forward_pre_hook(pre))
            hooks.append(mod.register_forward_hook(post))
    # the fused path calls conv._conv_forward directly (no module hooks): patch it
    import stardist_amd.models.unet as U
    orig = U._conv_nobias
    names = {id(m): n for n, m in net.named_modules()}

    def timed(conv, x):
        n = names.get(id(conv), "?")
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        y = orig(conv, x)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        recs.setdefault(n, []).append([e0, e1, tuple(x.shape)])
        return y
    U._conv_nobias = timed
    try:
        with torch.no_grad():
            for _ in range(reps):
                recs.clear()
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
                model._net_forward(x)
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                torch.cuda.synchronize()
    finally:
        U._conv_nobias = orig
        for h in hooks:
            h.remove()
    tot = e0.elapsed_time(e1)
    rows = []
    for n, lst in recs.items():
        for a, b, shp in lst:
            if b is not None:
                rows.append((n, shp, a.elapsed_time(b)))
    return tot, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--size3d", type=int, default=256)
    ap.add_argument("--skip-2d", action="store_true")
    ap.add_argument("--skip-3d", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    legs = []
    if not a.skip_2d:
        legs.append(("2D", StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0), (a.size, a.size)))
    if not a.skip_3d:
        legs.append(("3D", StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0), (a.size3d,) * 3))
    for tag, model, shape in legs:
        img = np.random.RandomState(0).rand(*shape).astype(np.float32)
        x, *_ = model._predict_setup(img, None, None, None)
        xt = torch.as_tensor(x, device=dev)
        model.use_hip_graph = False
        with torch.no_grad():
            for _ in range(2):
                model._net_forward(xt)
        torch.cuda.synchronize()
        tot, rows = layer_times(model, xt)
        print("== %s network forward %.2f ms; convolutions %.2f ms" % (tag, tot, sum(r[2] for r in rows)))
        for n, shp, ms in rows:
            conv = dict(model.net.named_modules())[n]
            px = int(np.prod(shp[2:])) * shp[0]
            fl = 2.0 * px * conv.in_channels * conv.out_channels * int(np.prod(conv.kernel_size))
            print("  %-28s in %-28s k%s %4d->%-4d %8.3f ms %7.1f TFLOP/s" % (n, shp, "x".join(map(str, conv.kernel_size)), conv.in_channels, conv.out_channels, ms, fl / ms / 1e9))
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            with torch.no_grad():
                model._net_forward(xt)
            torch.cuda.synchronize()
        seen = {}
        for ev in prof.events():
            if any(k in ev.name for k in ("slow_conv", "miopen_convolution", "thnn_conv", "_convolution", "vol2col", "im2col")) and ev.input_shapes:
                key = (ev.name, str(ev.input_shapes[:2]))
                seen[key] = seen.get(key, 0) + 1
        print("  aten convolution back-ends:")
        for (n, s), c in sorted(seen.items()):
            if n in ("aten::_convolution",):
                continue
            print("    %-40s x%d %s" % (n, c, s))


if __name__ == "__main__":
    main()
