"""2D NMS: the deferral of the enclosure's undecided pairs to the tail batch (option nms2d_defer_undecided = first deferring round) must
not change a single keep flag; time per setting on the bench's candidate set.  usage: python tools/check_defer.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import synth
from stardist_amd import nms
from stardist_amd.lib import _native as N, stardist2d as sd2
from stardist_amd.models import Config2D, StarDist2D
dev = torch.device("cuda:0")
sets = []
for shape, R, thr in (((256, 256), 32, 0.4), ((512, 512), 32, 0.4), ((356, 299), 11, 0.5), ((114, 217), 32, 0.3), ((768, 768), 32, 0.2)):
    d, p, s = synth.s2d_uniform(shape[0], shape[1], n_rays=R)
    sets.append(("uniform %s R=%d thr=%g" % (shape, R, thr), d, p, thr))
img = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
bench.calibrate_heads(m, img)
prob, dist, points = m.predict_sparse(img)
o = nms._argsort_desc(prob)
for thr in (0.4, 0.3, 0.6):
    sets.append(("bench set thr=%g" % thr, np.ascontiguousarray(dist[o]), np.ascontiguousarray(points[o].astype(np.float32)), thr))
ok = True
for name, d, p, thr in sets:
    td, tp = torch.from_numpy(d).to(dev), torch.from_numpy(p).to(dev)
    res = {}
    for opt in (0, 2, 1):
        with N.option("nms2d_defer_undecided", opt):
            ts = []
            for rep in range(3):
                torch.cuda.synchronize(); t = time.time()
                keep = sd2.c_non_max_suppression_inds(td, tp, 1, 1, 0, np.float32(thr))
                torch.cuda.synchronize(); ts.append(time.time() - t)
            st = N.last_stats["nms2d"]
            k = keep.cpu().numpy() if hasattr(keep, "cpu") else np.asarray(keep)
            res[opt] = (k, min(ts), st[4] / 1e6, st[2], st[10])
    same = all(np.array_equal(res[0][0], res[o_][0]) for o_ in (1, 2))
    ok &= same
    print("%-34s N=%7d -> %6d  %s | " % (name, len(d), int(res[0][0].sum()), "SAME" if same else "DIFFERENT") +
          "  ".join("opt %d: %.2f ms (pair %.2f, rounds %d, deferred %d)" % (o_, 1e3 * res[o_][1], res[o_][2], res[o_][3], res[o_][4]) for o_ in (0, 2, 1)), flush=True)
print("ALL SAME" if ok else "MISMATCH")
