"""Workload for the rocprofv3 --pmc passes: 2 x StarDist2D.predict_instances on the 2048^2 bench tile and
2 x StarDist3D.predict_instances on the 256^3 bench volume (same models / calibration as bench.py, no timing).
usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o p -- python tools/pmc_predict.py [--skip-3d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import synth
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D

dev = torch.device("cuda:0")
img = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
m2 = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
m2.thresholds = dict(prob=0.5, nms=0.4)
bench.calibrate_heads(m2, img)
N2 = int(os.environ.get("SD_PMC_STEPS", "2"))
for _ in range(N2):
    lab, res = m2.predict_instances(img)
print("2D:", len(res["prob"]), "instances")
from stardist_amd.lib import _native
st = _native.last_stats["nms2d"]
print("PAIRS_PER_STEP=%d PAIR_LAUNCHES_PER_STEP=%d GENERAL_PATH_PAIRS=%d SIZE=2048" % (st[0], st[5], st[1]), flush=True)
if "--skip-3d" not in sys.argv:
    del m2, img
    torch.cuda.empty_cache()
    vol = torch.from_numpy(synth.s3d_nuclei_image(256, seed=0)).to(dev)
    m3 = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
    m3.thresholds = dict(prob=0.5, nms=0.3)
    bench.calibrate_heads(m3, vol, frac=0.009, radius=8.5, noise=0.03)
    for _ in range(N2):
        lab, res = m3.predict_instances(vol)
    print("3D:", len(res["prob"]), "instances")
