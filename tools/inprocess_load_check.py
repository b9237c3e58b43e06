"""The self-check of the decision kernel (tools/patches/nms2d_decide_selfcheck_probe.patch built as libstardist_hip_probe.so) in ONE process while
a second stream of the SAME process keeps the device busy with large matrix products: does in-process load alone make the kernel disagree with
itself, or does it take several processes?  usage: SD_LIB=stardist_amd/csrc/libstardist_hip_probe.so python tools/inprocess_load_check.py REPS [load: 0|1]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from oracle import synth
from stardist_amd import nms
from stardist_amd.lib import _native, stardist2d as sd2
from stardist_amd.models import Config2D, StarDist2D
if os.environ.get("SD_LIB"):
    _native.LIB_PATH = os.path.join(ROOT, os.environ["SD_LIB"]) if not os.path.isabs(os.environ["SD_LIB"]) else os.environ["SD_LIB"]
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
load = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_native.check(_native.lib().sd_set_option(b"probe_tier", 77))
img = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
bench.calibrate_heads(m, img)
prob, dist, points = m.predict_sparse(img)
o = nms._argsort_desc(prob)
td = torch.from_numpy(np.ascontiguousarray(dist[o])).to(dev); tp = torch.from_numpy(np.ascontiguousarray(points[o].astype(np.float32))).to(dev)
stop = False
def burn():
    s2 = torch.cuda.Stream(device=dev)
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    with torch.cuda.stream(s2):
        while not stop:
            for _ in range(8):
                c = a @ b
            s2.synchronize()
th = None
if load:
    th = threading.Thread(target=burn, daemon=True); th.start(); time.sleep(0.5)
crcs = set()
import zlib
t0 = time.time()
for r in range(reps):
    keep = sd2.c_non_max_suppression_inds(td, tp, 1, 1, 0, np.float32(0.4))
    torch.cuda.synchronize()
    crcs.add(zlib.crc32(keep.cpu().numpy().astype(np.uint8).tobytes()))
stop = True
print("in-process load %d: %d NMS calls in %.1f s, distinct keep CRCs: %s" % (load, reps, time.time() - t0, ["%08x" % c for c in crcs]), flush=True)
