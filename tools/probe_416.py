"""Where does a 416^3 volume (more than 2^31 elements per 32-channel level) go wrong?  One process, stage by stage:
  A  predict_instances three times (instance counts, label sums)
  B  the network alone: eager twice, HIP graph twice -> probability maps compared bit for bit, feature checksums
  C  every layer's output near the two far corners against the same layers on the 256^3 corner crops (beyond the receptive field
     the values must agree)
usage: python tools/probe_416.py [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import synth
from stardist_amd.models import Config3D, StarDist3D, unet

S = int(sys.argv[1]) if len(sys.argv) > 1 else 416
stages = sys.argv[2] if len(sys.argv) > 2 else "ABC"
dev = torch.device("cuda:0")
tile = torch.from_numpy(synth.s3d_nuclei_image(256, seed=0)).to(dev)
m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
m.thresholds = dict(prob=0.5, nms=0.3)
bench.calibrate_heads(m, tile, frac=0.009, radius=8.5, noise=0.03)
vol = tile.repeat(2, 2, 2)[:S, :S, :S].contiguous()
print("volume", tuple(vol.shape), "conv mode", unet.conv_mode(), flush=True)

if "A" in stages:
    for rep in range(3):
        t = time.time()
        labels, res = m.predict_instances(vol)
        torch.cuda.synchronize()
        print("A rep %d: %.2f s  instances %d  label sum %d" % (rep, time.time() - t, len(res["prob"]), int(np.asarray(labels, np.int64).sum())), flush=True)
    m.__dict__.pop("_graphs", None)
    torch.cuda.empty_cache()

xc = vol[None, None].contiguous(memory_format=torch.channels_last_3d)


def forward(graph):
    m.use_hip_graph = graph
    if graph:
        ys = m._net_forward(vol[..., None], sparse_head=True)
        prob, feat = ys[0], ys[1]
    else:
        prob, feat = m._net_eager(xc, sparse_head=True)[:2]
    torch.cuda.synchronize()
    return prob.reshape(-1).clone(), float(torch.sum(feat, dtype=torch.float64))


if "B" in stages:
    ref = None
    for name, graph in (("eager0", False), ("eager1", False), ("graph0", True), ("graph1", True), ("graph2", True)):
        p, fs = forward(graph)
        if ref is None:
            ref = p
        ne = p != ref
        n = int(ne.sum())
        msg = "B %s: prob sum %.6f  feature sum %.6f  differs from eager0 in %d voxels" % (name, float(p.double().sum()), fs, n)
        if n:
            idx = torch.nonzero(ne).reshape(-1)
            z = idx // (S * S); y = (idx // S) % S; x = idx % S
            msg += "  z %d..%d y %d..%d x %d..%d  max |d| %.3g" % (int(z.min()), int(z.max()), int(y.min()), int(y.max()), int(x.min()), int(x.max()),
                                                                    float((p - ref).abs().max()))
        print(msg, flush=True)
        del p
    m.__dict__.pop("_graphs", None)
    del ref
    torch.cuda.empty_cache()

if "C" in stages:
    rec = []
    orig_conv, orig_pool = unet._hand_conv, unet.max_pool
    mode = {}

    def keep(out, tag):
        if out is None or out.dim() != 5:
            return out
        full = mode["full"]
        s = full // out.shape[2]
        w = 96 // s
        for corner in ("near", "far"):
            if mode["which"] in ("big", corner):
                sl = slice(0, w) if corner == "near" else slice(out.shape[2] - w, out.shape[2])
                rec.append((mode["which"], corner, len([r for r in rec if r[0] == mode["which"] and r[1] == corner]), tag, tuple(out.shape),
                            out[:, :, sl, sl, sl].clone()))
        return out

    unet._hand_conv = lambda conv, srcs, kind, res=None, bn=None, tf_same=False: keep(orig_conv(conv, srcs, kind, res, bn, tf_same),
                                                                                     "conv %d->%d" % (conv.in_channels, conv.out_channels))
    unet.max_pool = lambda x, pool: keep(orig_pool(x, pool), "pool")
    m.use_hip_graph = False
    mode.update(which="big", full=S)
    pb = m._net_eager(xc, sparse_head=True)[0]
    torch.cuda.synchronize()
    for which, crop in (("near", vol[:256, :256, :256]), ("far", vol[S - 256:, S - 256:, S - 256:])):
        mode.update(which=which, full=256)
        pc = m._net_eager(crop.contiguous()[None, None].contiguous(memory_format=torch.channels_last_3d), sparse_head=True)[0]
        torch.cuda.synchronize()
        a = pb[0, 0, :96, :96, :96] if which == "near" else pb[0, 0, S - 96:, S - 96:, S - 96:]
        b = pc[0, 0, :96, :96, :96] if which == "near" else pc[0, 0, 160:, 160:, 160:]
        print("C prob %s corner: differing voxels %d of %d, max |d| %.3g" % (which, int((a != b).sum()), a.numel(), float((a - b).abs().max())), flush=True)
    unet._hand_conv, unet.max_pool = orig_conv, orig_pool
    big = {(r[1], r[2]): r for r in rec if r[0] == "big"}
    for r in rec:
        if r[0] == "big":
            continue
        o = big[(r[1], r[2])]
        a, b = o[5], r[5]
        if a.shape != b.shape:
            print("C %s layer %d %s: shapes %s vs %s" % (r[1], r[2], r[3], tuple(a.shape), tuple(b.shape)), flush=True)
            continue
        ne = (a != b)
        n = int(ne.sum())
        msg = "C %s layer %2d %-14s big %s: differing %d of %d" % (r[1], r[2], r[3], o[4], n, a.numel())
        if n:
            idx = torch.nonzero(ne)
            msg += "  max |d| %.3g  z %d..%d y %d..%d x %d..%d (of %d)" % (float((a - b).abs().max()), int(idx[:, 2].min()), int(idx[:, 2].max()),
                                                                           int(idx[:, 3].min()), int(idx[:, 3].max()), int(idx[:, 4].min()),
                                                                           int(idx[:, 4].max()), a.shape[2])
        print(msg, flush=True)
print("done", flush=True)
