"""Shared by tests/test_gpu_bigparity.py and tests/golden/make_sharded_golden.py: the full-size sharded workloads of bench.py
(BASELINE.json configs 4 / 5) and the candidate list of the WHOLE input in the order predict_instances on the whole input would
hand it to the NMS."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG2D = dict(size=16384, tile=2048, block=4416, overlap=128, context=128)           # bench.py --sharded-size / --sharded-block
CFG3D = dict(size=1024, tile=256, block=560, overlap=32, context=32)
CFG3D_REF = dict(size=512, tile=256, block=304, overlap=32, context=32)             # 8 blocks; the compiled reference finishes in minutes


def model_and_input(dim, cfg, dev):
    """the calibrated model and the big input exactly as bench.py builds them (seed-0 tile repeated)"""
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    rep = cfg["size"] // cfg["tile"]
    if dim == 2:
        tile = torch.from_numpy(synth.s2d_nuclei_image(cfg["tile"], cfg["tile"], seed=0)).to(dev)
        model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
        bench.calibrate_heads(model, tile)
        return model, tile.repeat(rep, rep), "YX"
    tile = torch.from_numpy(synth.s3d_nuclei_image(cfg["tile"], seed=0)).to(dev)
    model = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
    model.thresholds = dict(prob=0.5, nms=0.3)
    bench.calibrate_heads(model, tile, frac=0.009, radius=8.5, noise=0.03)
    return model, tile.repeat(rep, rep, rep), "ZYX"


def whole_input_candidates(model, big, axes, cfg):
    """(dist, prob, points) device tensors: every candidate of the whole input exactly once (a pixel belongs to the block whose write
    region exclusively covers it, else to the first block that reports it), ordered as nms._argsort_desc orders the row-major
    candidate list of the whole input -- the order the reference's predict_instances hands to its NMS (stardist/nms.py:163-167)."""
    import torch
    from stardist_amd.big import BlockND
    from stardist_amd.nms import _argsort_desc
    nd = len(axes)
    grid = model._axes_div_by(axes)
    blocks = BlockND.cover(big.shape, axes, (cfg["block"],) * nd, (cfg["overlap"],) * nd, (cfg["context"],) * nd, grid)
    P, D, X = [], [], []
    for block in blocks:
        r = model.predict_sparse_device(block.read(big, axes=axes), axes=axes)
        prob, dist, pts = r[0], r[1], r[-1]
        bl = block.blocks_for_axes(axes)
        start = torch.tensor([t.start for t in bl], device=prob.device, dtype=torch.int64).reshape(1, nd)
        lo = torch.tensor([t.start + t.context_start for t in bl], device=prob.device, dtype=torch.int64).reshape(1, nd)
        hi = torch.tensor([t.end - t.context_end for t in bl], device=prob.device, dtype=torch.int64).reshape(1, nd)
        gp = pts.to(torch.int64) + start
        ins = torch.all((gp >= lo) & (gp < hi), dim=1)
        P.append(prob[ins].clone()); D.append(dist[ins].clone()); X.append(gp[ins])
    prob, dist, pts = torch.cat(P), torch.cat(D), torch.cat(X)
    del P, D, X
    key = pts[:, 0]
    for d in range(1, nd):
        key = key * int(big.shape[d]) + pts[:, d]
    ks, ki = torch.sort(key, stable=True)
    first = torch.ones_like(ks, dtype=torch.bool)
    first[1:] = ks[1:] != ks[:-1]
    sel = ki[first]                                   # row-major, each pixel once
    prob, dist, pts = prob[sel], dist[sel], pts[sel]
    so = _argsort_desc(prob)
    return dist[so].contiguous(), prob[so].contiguous(), pts[so].contiguous(), len(blocks)


def points_digest(points):
    """SHA-256 of the survivors' centres (int32, rows sorted lexicographically): independent of the order they are listed in"""
    p = np.ascontiguousarray(np.asarray(points).astype(np.int32))
    p = p[np.lexsort(p.T[::-1])]
    return hashlib.sha256(p.tobytes()).hexdigest()


def array_digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
