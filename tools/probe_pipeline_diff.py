"""Where do the pipelined and the serialised sharded pass differ?  No synchronisation is added: the per-block survivor records (what
phase 1 hands to the exchange) are captured by wrapping torch.cat in stardist_amd.big and compared after the pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import stardist_amd.big as B
from oracle import synth
from stardist_amd.models import Config2D, StarDist2D
dev = torch.device("cuda:0")
tile = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
bench.calibrate_heads(m, tile)
big = tile.repeat(3, 3)
order = sys.argv[1:] or ["pipe", "serial", "pipe2"]
out = {}
for mode in order:
    labels, res = m.predict_instances_sharded(big, "YX", block_size=2048, min_overlap=128, context=128, pipeline=mode.startswith("pipe"), broadcast_result=True)
    out[mode] = (np.asarray(labels).copy(), res)
    print(mode, "instances", len(res["prob"]), "label sum", int(np.asarray(labels, np.int64).sum()), dict((k, m._last_sharded_stats[k]) for k in ("gathered", "unique", "band", "interior", "pipelined")), flush=True)
ref = out["serial"]
for mode in order:
    if mode == "serial": continue
    l, r = out[mode]
    print("== serial vs", mode, ": labels equal", np.array_equal(ref[0], l), " points equal", np.array_equal(ref[1]["points"], r["points"]),
          " prob equal", np.array_equal(ref[1]["prob"], r["prob"]), " coord equal", np.array_equal(ref[1]["coord"], r["coord"]))
    if not np.array_equal(ref[0], l):
        d = np.argwhere(ref[0] != l)
        print("   differing pixels:", len(d), "bbox", d.min(0), d.max(0), " ids", np.unique(ref[0][ref[0] != l])[:10], np.unique(l[ref[0] != l])[:10])
    if not np.array_equal(ref[1]["prob"], r["prob"]):
        k = np.flatnonzero(ref[1]["prob"] != r["prob"])
        print("   differing instances:", len(k), k[:10], ref[1]["prob"][k[:5]], r["prob"][k[:5]], ref[1]["points"][k[:5]].tolist(), r["points"][k[:5]].tolist())
