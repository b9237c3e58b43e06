"""Time the 2D NMS alone on the BENCH's candidate set (calibrated U-Net on the 2048^2 synthetic tile). usage: python tools/time_nms2d_bench.py [reps]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import synth
from stardist_amd import nms
from stardist_amd.lib import _native, stardist2d as sd2
from stardist_amd.models import Config2D, StarDist2D
dev = torch.device("cuda:0")
if os.environ.get("SD_LIB"):                      # a probe build of the library (SD_BUILD_DEBUG_SWITCHES=1: tuning knobs read from the environment)
    _native.LIB_PATH = os.environ["SD_LIB"]
if os.environ.get("SD_AREA_BOUNDS"):
    _native.check(_native.lib().sd_set_option(b"nms2d_area_bounds", int(os.environ["SD_AREA_BOUNDS"])))
if os.environ.get("SD_PAIR_LANES"):
    _native.check(_native.lib().sd_set_option(b"nms2d_pair_lanes", int(os.environ["SD_PAIR_LANES"])))
    print("nms2d_pair_lanes =", _native.lib().sd_get_option(b"nms2d_pair_lanes"))
for env, opt in (("SD_DEFER_FROM", b"nms2d_defer_undecided"), ("SD_DEFER_MAX", b"nms2d_defer_max")):
    if os.environ.get(env):
        _native.check(_native.lib().sd_set_option(opt, int(os.environ[env])))
for kv in filter(None, os.environ.get("SD_OPTS", "").split(",")):          # SD_OPTS="name=value,name=value": any sd_set_option switch
    k, v = kv.split("=")
    _native.check(_native.lib().sd_set_option(k.encode(), int(v)))
    print(k, "=", _native.lib().sd_get_option(k.encode()))
if os.environ.get("SD_TRACE"):
    _native.lib().sd_set_option(b"trace", 1)      # per-round counters on stdout
img = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
bench.calibrate_heads(m, img)
prob, dist, points = m.predict_sparse(img)
o = nms._argsort_desc(prob)
td = torch.from_numpy(np.ascontiguousarray(dist[o])).to(dev); tp = torch.from_numpy(np.ascontiguousarray(points[o].astype(np.float32))).to(dev)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    torch.cuda.synchronize(); t = time.time()
    keep = sd2.c_non_max_suppression_inds(td, tp, 1, 1, 0, np.float32(0.4))
    torch.cuda.synchronize(); dt = time.time() - t
    st = _native.last_stats["nms2d"]
    crc = zlib.crc32((keep.cpu().numpy() if hasattr(keep, "cpu") else np.asarray(keep)).astype(np.uint8).tobytes())
    print(f"rep {r}: N={len(td)} -> {int(keep.sum())}  {dt*1e3:.1f} ms  pair {st[4]/1e6:.2f} ms ({st[0]} pairs, {st[5]} launches)  general {st[6]/1e6:.2f} ms ({st[1]})  build {st[7]/1e6:.2f} ms  rounds {st[2]}  keep crc {crc:08x}", flush=True)
