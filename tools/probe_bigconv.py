"""Times the full-resolution 64->32 3x3x3 convolution of the 3D U-Net (input exactly 2**30 elements = 4 GiB), whole vs
z-slabs with halo vs split over the input channels (no concat).  usage: python tools/probe_bigconv.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stardist_amd  # noqa
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
cl = torch.channels_last_3d


def t(fn, n=3):
    for _ in range(2):
        y = fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y


g = torch.Generator(device="cpu").manual_seed(0)
D = 256
a = torch.randn(1, 32, D, D, D, generator=g).to(dev).contiguous(memory_format=cl)
b = torch.randn(1, 32, D, D, D, generator=g).to(dev).contiguous(memory_format=cl)
w = (torch.randn(32, 64, 3, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=cl)
with torch.no_grad():
    ms_cat, x = t(lambda: torch.cat([a, b], 1))
    print("concat %.2f ms  %s" % (ms_cat, x.is_contiguous(memory_format=cl)))
    ms, y0 = t(lambda: F.conv3d(x, w, None, 1, 1))
    print("whole            %.2f ms" % ms)
    for ns in (2, 4, 8):
        def slabs():
            out = torch.empty((1, 32, D, D, D), device=dev).contiguous(memory_format=cl)
            step = D // ns
            for z0 in range(0, D, step):
                z1 = z0 + step
                lo, hi = max(0, z0 - 1), min(D, z1 + 1)
                xs = x[:, :, lo:hi]
                pz0, pz1 = (1 if z0 == 0 else 0), (1 if z1 == D else 0)
                if pz0 or pz1:
                    xs = F.pad(xs, (0, 0, 0, 0, pz0, pz1))
                out[:, :, z0:z1] = F.conv3d(xs, w, None, 1, (0, 1, 1))
            return out
        ms, y = t(slabs)
        print("z-slabs x%d       %.2f ms  max|d| %.3g  equal %s" % (ns, ms, (y - y0).abs().max().item(), torch.equal(y, y0)))
    wa, wb = w[:, :32].contiguous(memory_format=cl), w[:, 32:].contiguous(memory_format=cl)
    ms, y = t(lambda: F.conv3d(a, wa, None, 1, 1).add_(F.conv3d(b, wb, None, 1, 1)))
    print("channel split    %.2f ms (+ no concat)  max|d| %.3g" % (ms, (y - y0).abs().max().item()))
    # slabs straight from the two sources (no concat of the whole tensor: concat per slab)
    def slabs_nocat(ns=4):
        out = torch.empty((1, 32, D, D, D), device=dev).contiguous(memory_format=cl)
        step = D // ns
        for z0 in range(0, D, step):
            z1 = z0 + step
            lo, hi = max(0, z0 - 1), min(D, z1 + 1)
            xs = torch.cat([a[:, :, lo:hi], b[:, :, lo:hi]], 1)
            pz0, pz1 = (1 if z0 == 0 else 0), (1 if z1 == D else 0)
            if pz0 or pz1:
                xs = F.pad(xs, (0, 0, 0, 0, pz0, pz1))
            out[:, :, z0:z1] = F.conv3d(xs, w, None, 1, (0, 1, 1))
        return out
    ms, y = t(slabs_nocat)
    print("slabs x4 incl. per-slab concat %.2f ms  equal %s" % (ms, torch.equal(y, y0)))
    # 2D analogue: 64->32 at 2048^2
    a2 = torch.randn(1, 32, 2048, 2048, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    b2 = torch.randn(1, 32, 2048, 2048, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w2 = (torch.randn(32, 64, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    ms_c, x2 = t(lambda: torch.cat([a2, b2], 1), 10)
    ms_w, _ = t(lambda: F.conv2d(x2, w2, None, 1, 1), 10)
    w2a, w2b = w2[:, :32].contiguous(memory_format=torch.channels_last), w2[:, 32:].contiguous(memory_format=torch.channels_last)
    ms_s, _ = t(lambda: F.conv2d(a2, w2a, None, 1, 1).add_(F.conv2d(b2, w2b, None, 1, 1)), 10)
    print("2D 64->32 @2048^2: concat %.3f + conv %.3f ms   vs channel split %.3f ms" % (ms_c, ms_w, ms_s))
