"""Time the 3D NMS alone on the BENCH's candidate set (calibrated U-Net on the 256^3 synthetic volume). SD_TRACE=1 prints the cascade counters.
usage: python tools/time_nms3d_bench.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import synth
from stardist_amd import nms
from stardist_amd.lib import _native, stardist3d as sd3
from stardist_amd.models import Config3D, StarDist3D
from stardist_amd.rays3d import rays_from_json
dev = torch.device("cuda:0")
if os.environ.get("SD_SPLIT_EXACT"):
    _native.check(_native.lib().sd_set_option(b"nms3d_split_exact", int(os.environ["SD_SPLIT_EXACT"])))
    print("nms3d_split_exact =", _native.lib().sd_get_option(b"nms3d_split_exact"))
if os.environ.get("SD_TAIL"):
    _native.check(_native.lib().sd_set_option(b"nms3d_tail_batch", int(os.environ["SD_TAIL"])))
for env, name in (("SD_DEFER3", b"nms3d_defer_exact"), ("SD_REUSE", b"nms3d_bounds_reuse")):
    if os.environ.get(env):
        _native.check(_native.lib().sd_set_option(name, int(os.environ[env])))
        print(name.decode(), "=", _native.lib().sd_get_option(name))
if os.environ.get("SD_TRACE"):
    _native.lib().sd_set_option(b"trace", 1)      # per-round counters on stdout
S = int(os.environ.get("SD_SIZE3D", "256"))
vol = torch.from_numpy(synth.s3d_nuclei_image(S, seed=0)).to(dev)
m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
m.thresholds = dict(prob=0.5, nms=0.3)
bench.calibrate_heads(m, vol, frac=0.009, radius=8.5, noise=0.03)
prob, dist, points = m.predict_sparse(vol)
o = nms._argsort_desc(prob)
rays = rays_from_json(m.config.rays_json)
td = torch.from_numpy(np.ascontiguousarray(dist[o])).to(dev); tp = torch.from_numpy(np.ascontiguousarray(points[o].astype(np.float32))).to(dev)
ts = torch.from_numpy(np.ascontiguousarray(prob[o])).to(dev)
tV = torch.from_numpy(np.ascontiguousarray(rays.vertices, np.float32)).to(dev); tF = torch.from_numpy(np.ascontiguousarray(rays.faces, np.int32)).to(dev)
del m, vol
torch.cuda.empty_cache()
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    torch.cuda.synchronize(); t = time.time()
    keep = sd3.c_non_max_suppression_inds(td, tp, tV, tF, ts, 1, 1, 0, np.float32(0.3))
    torch.cuda.synchronize(); dt = time.time() - t
    st = _native.last_stats["nms3d"]
    print(f"rep {r}: N={len(td)} -> {int(keep.sum())}  {dt*1e3:.1f} ms  stage3 {st[8]/1e6:.1f} ms ({st[2]} pairs)  stage4 {st[9]/1e6:.1f} ms ({st[11]} pairs)  "
          f"stage5 {st[10]/1e6:.1f} ms ({st[3]})  rounds {st[4]}  near-threshold exact-volume decisions {st[13]} large-face fallbacks {st[14]}", flush=True)
