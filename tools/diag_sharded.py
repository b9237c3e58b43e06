"""Diagnostics of 'big == whole' at full size (GPU box): the monolithic HIP NMS over every candidate of the whole input against the
committed reference golden (keep bits), and the block-sharded result against the monolithic one -- where do differing instances lie
relative to the block geometry, and how does the count depend on the context?
usage: python tools/diag_sharded.py 2d|3d512|3d1024 [context ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _bigparity as B  # noqa: E402
from stardist_amd.big import BlockND  # noqa: E402


def edge_distance(points, big, axes, block, cfg, ctx, grid):
    dim = len(axes)
    blocks = BlockND.cover(big.shape, axes, (block,) * dim, (cfg["overlap"],) * dim, (ctx,) * dim, grid)
    edges = [sorted(set(v for b in blocks for t in [b.blocks_for_axes(axes)[a]] for v in (t.start, t.end, t.start + t.context_start, t.end - t.context_end)))
             for a in range(dim)]
    dmin = np.full(len(points), 1e9)
    for a in range(dim):
        e = np.asarray([v for v in edges[a] if 0 < v < big.shape[a]])
        if len(e):
            dmin = np.minimum(dmin, np.abs(points[:, a:a + 1] - e[None]).min(1))
    return dmin


def sweep_2d(model, big, axes, cfg, gold, dev):
    """the monolithic NMS over the 27 M candidates of the whole slide exceeds the HIP kernel's documented capacity (2^31 neighbour
    entries per call), so the whole-slide truth is the committed golden of the compiled reference; the sharded prediction is run with
    growing context until its instances hash to the golden, which also yields the reference's survivor list for locating differences"""
    grid = model._axes_div_by(axes)
    results = {}
    for ctx in [int(v) for v in sys.argv[2:]] or [cfg["context"]]:
        model.__dict__.pop("_graphs", None); torch.cuda.empty_cache()
        block = -(-(cfg["block"] + 2 * (ctx - cfg["context"])) // max(grid)) * max(grid)
        t0 = time.time()
        out = model.predict_instances_sharded(big, axes, block_size=block, min_overlap=cfg["overlap"], context=ctx, return_labels=False)
        st = model._last_sharded_stats
        pts = np.asarray(out[1]["points"]).astype(np.int64)
        same = B.points_digest(pts) == gold["points_sha256"]
        results[ctx] = (block, pts, same)
        print("2d: context %d, block %d (%d blocks, %.1f s): %d instances (reference over all %d candidates: %d) -- instance set %s" % (
            ctx, block, st["blocks"], time.time() - t0, st["instances"], gold["candidates"], gold["survivors"],
            "IDENTICAL to the reference's" if same else "differs"), flush=True)
    exact = [c for c, (_, _, same) in results.items() if same]
    if not exact:
        print("no context reproduced the reference exactly"); return
    ref = results[exact[0]][1]
    key = lambda p: p[:, 0] * int(big.shape[1]) + p[:, 1]
    rk = key(ref)
    for ctx, (block, pts, same) in results.items():
        if same:
            continue
        k = key(pts)
        only_s, only_r = pts[~np.isin(k, rk)], ref[~np.isin(rk, k)]
        dmin = edge_distance(np.concatenate([only_s, only_r]), big, axes, block, cfg, ctx, grid)
        print("   context %d: %d instances only in the sharded result, %d only in the reference's; distance of their centres to the nearest read / write "
              "region edge: min %d, median %d, max %d; histogram (0-32-64-96-128-192-256+): %s" % (
                  ctx, len(only_s), len(only_r), dmin.min(), np.median(dmin), dmin.max(), np.histogram(dmin, [0, 32, 64, 96, 128, 192, 256, 1e9])[0].tolist()), flush=True)


def main():
    which = sys.argv[1]
    dev = torch.device("cuda:0")
    dim = 2 if which == "2d" else 3
    cfg = dict(B.CFG2D if which == "2d" else (B.CFG3D_REF if which == "3d512" else B.CFG3D))
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "sharded_fullsize.json"))).get("2d" if dim == 2 else "3d") if which != "3d1024" else None
    model, big, axes = B.model_and_input(dim, cfg, dev)
    if which == "2d":
        return sweep_2d(model, big, axes, cfg, gold, dev)
    t0 = time.time()
    dist, prob, pts, nb = B.whole_input_candidates(model, big, axes, cfg)
    n = int(prob.numel())
    torch.cuda.synchronize(); t1 = time.time()
    if dim == 2:
        from stardist_amd.lib.stardist2d import c_non_max_suppression_inds
        keep = c_non_max_suppression_inds(dist.float().contiguous(), pts.float().contiguous(), 1, 1, 0, np.float32(model.thresholds.nms))
    else:
        from stardist_amd.lib.stardist3d import c_non_max_suppression_inds
        from stardist_amd.rays3d import rays_from_json
        rays = rays_from_json(model.config.rays_json)
        verts = torch.as_tensor(np.ascontiguousarray(rays.vertices, np.float32), device=dev)
        faces = torch.as_tensor(np.ascontiguousarray(rays.faces, np.int32), device=dev)
        keep = c_non_max_suppression_inds(dist.float().contiguous(), pts.float().contiguous(), verts, faces, prob.float().contiguous(), 1, 1, 0,
                                          np.float32(model.thresholds.nms))
    torch.cuda.synchronize(); t2 = time.time()
    keep = keep.bool()
    kh = keep.cpu().numpy()
    print("%s: %d candidates (%d blocks, %.1f s), monolithic HIP NMS %.2f s -> %d survivors" % (which, n, nb, t1 - t0, t2 - t1, int(kh.sum())), flush=True)
    if gold is not None:
        print("   reference golden: candidates %d, survivors %d, keep bits %s" % (gold["candidates"], gold["survivors"],
              "IDENTICAL to the compiled reference's" if B.array_digest(np.packbits(kh)) == gold["keep_sha256"] else "DIFFER from the compiled reference's"), flush=True)
    mono = pts[keep]
    shape = torch.tensor(list(big.shape), device=dev)

    def lin(p):
        k = p[:, 0]
        for d in range(1, dim):
            k = k * int(big.shape[d]) + p[:, d]
        return k
    mono_key = lin(mono)
    del dist, prob
    torch.cuda.empty_cache()
    for ctx in [int(v) for v in sys.argv[2:]] or [cfg["context"]]:
        model.__dict__.pop("_graphs", None); torch.cuda.empty_cache()
        block = cfg["block"] + 2 * (ctx - cfg["context"])
        grid = model._axes_div_by(axes)
        block = -(-block // max(grid)) * max(grid)
        try:
            out = model.predict_instances_sharded(big, axes, block_size=block, min_overlap=cfg["overlap"], context=ctx, return_labels=False)
        except Exception as e:
            print("   context %d block %d: %r" % (ctx, block, e)); continue
        st = model._last_sharded_stats
        sp = torch.as_tensor(np.asarray(out[1]["points"]), device=dev).to(torch.int64)
        sk = lin(sp)
        only_s = sp[~torch.isin(sk, mono_key)]; only_m = mono[~torch.isin(mono_key, sk)]
        print("   context %d, block %d (%d blocks): sharded %d instances; %d only in the sharded result, %d only in the monolithic one" % (
            ctx, block, st["blocks"], st["instances"], len(only_s), len(only_m)), flush=True)
        if len(only_s) + len(only_m):
            blocks = BlockND.cover(big.shape, axes, (block,) * dim, (cfg["overlap"],) * dim, (ctx,) * dim, grid)
            edges = [sorted(set(v for b in blocks for t in [b.blocks_for_axes(axes)[a]] for v in (t.start, t.end, t.start + t.context_start, t.end - t.context_end)))
                     for a in range(dim)]
            both = torch.cat([only_s, only_m]).cpu().numpy()
            dmin = np.full(len(both), 1e9)
            for a in range(dim):
                e = np.asarray([v for v in edges[a] if 0 < v < big.shape[a]])
                if len(e):
                    dmin = np.minimum(dmin, np.abs(both[:, a:a + 1] - e[None]).min(1))
            print("      distance of the differing centres to the nearest read / write region edge: min %d, median %d, max %d; histogram (0-32-64-96-128-192-256+): %s"
                  % (dmin.min(), np.median(dmin), dmin.max(), np.histogram(dmin, [0, 32, 64, 96, 128, 192, 256, 1e9])[0].tolist()), flush=True)


if __name__ == "__main__":
    main()
