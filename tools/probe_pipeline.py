"""Isolate the stages of the sharded bench legs, one per process (a GPU fault in one does not hide the others).
usage: python tools/probe_pipeline.py {2d-pipe|2d-serial|3d-pipe|3d-serial|3d-416|3d-352} [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import synth
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D

what = sys.argv[1]
dev = torch.device("cuda:0")
t0 = time.time()


def heads(m, calibrate):
    """SD_HEADS=<file>: the calibrated head parameters are taken from / saved to that file, so that several processes run ONE model
    (the calibration goes through library 1x1 convolutions and reductions whose results are not repeatable across processes)"""
    import hashlib
    f = os.environ.get("SD_HEADS")
    net = m.net
    ps = [net.prob.bias, net.dist.weight, net.dist.bias]
    if f and os.path.exists(f):
        with torch.no_grad():
            for p, q in zip(ps, torch.load(f)):
                p.copy_(q.to(p.device))
    else:
        calibrate()
        if f:
            torch.save([p.detach().cpu() for p in ps], f)
    print("head parameters sha1", hashlib.sha1(b"".join(p.detach().cpu().numpy().tobytes() for p in ps)).hexdigest()[:16], flush=True)


if what.startswith("2d"):
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
    tile = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
    m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    heads(m, lambda: bench.calibrate_heads(m, tile))
    big = tile.repeat(size // 2048, size // 2048)
    kw = dict(block_size=2048, min_overlap=128, context=128, pipeline=what.endswith("pipe"))
    axes = "YX"
else:
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    tile = torch.from_numpy(synth.s3d_nuclei_image(256, seed=0)).to(dev)
    m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
    m.thresholds = dict(prob=0.5, nms=0.3)
    heads(m, lambda: bench.calibrate_heads(m, tile, frac=0.009, radius=8.5, noise=0.03))
    big = tile.repeat(size // 256, size // 256, size // 256)
    blk = {"3d-416": 416, "3d-352": 352}.get(what, 256)
    kw = dict(block_size=blk, min_overlap=32, context=32, pipeline=not what.endswith("serial"))
    axes = "ZYX"
    if blk != 256:      # one block only: the network + selection + NMS at that block size
        big = big[:blk, :blk, :blk].contiguous()
print(what, "setup %.1f s, input %s" % (time.time() - t0, tuple(big.shape)), flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    labels, res = m.predict_instances_sharded(big, axes, **kw)
    torch.cuda.synchronize(); dt = time.time() - t
    st = m._last_sharded_stats
    print("rep %d: %.3f s  %.1f M/s  instances %d  pipelined %d  phase1 %.3f (predict wait %.3f, nms %.3f) final %.3f  labels sum %d" %
          (rep, dt, big.numel() / dt / 1e6, len(res["prob"]), st["pipelined"], st["t_phase1"], st["t_predict"], st["t_local_nms"], st["t_final"],
           int(np.asarray(labels, np.int64).sum())), flush=True)
    import hashlib
    print("   result sha1: labels %s  points %s  prob %s" % tuple(hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
                                                                for a in (np.asarray(labels), np.asarray(res["points"]), np.asarray(res["prob"]))), flush=True)
