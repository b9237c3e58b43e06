"""A/B of the result-preserving 3D NMS switches on the BENCH's candidate set (calibrated U-Net on the 256^3 synthetic volume), one
process: every combination of option values named in SD_COMBOS ("name=value,name=value;..."; default: "nms3d_defer_exact" 0..4 and
the round-4 form) -> median ms of `reps` calls, the keep flags compared with the first combination's; then the per-round trace of the
first and of the best one.
usage: python tools/time_nms3d_options.py [reps]"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
S = int(os.environ.get("SD_SIZE3D", "256"))
vol = torch.from_numpy(synth.s3d_nuclei_image(S, seed=0)).to(dev)
m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
m.thresholds = dict(prob=0.5, nms=0.3)
bench.calibrate_heads(m, vol, frac=0.009, radius=8.5, noise=0.03)
prob, dist, points = m.predict_sparse(vol)
o = nms._argsort_desc(prob)
rays = rays_from_json(m.config.rays_json)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
td, tp, ts = t(dist[o]), t(points[o].astype(np.float32)), t(prob[o])
tV, tF = t(np.float32(rays.vertices)), t(np.int32(rays.faces))
del m, vol
torch.cuda.empty_cache()
L = _native.lib()


def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    keep = sd3.c_non_max_suppression_inds(td, tp, tV, tF, ts, 1, 1, 0, np.float32(0.3))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, keep


first = None
res = {}
# SD_COMBOS="name=value,name=value;name=value;..." -- one timing per ';'-separated combination (options not named keep their defaults)
combos = os.environ.get("SD_COMBOS", "nms3d_defer_exact=0,nms3d_bounds_reuse=0;nms3d_defer_exact=0;nms3d_defer_exact=1;nms3d_defer_exact=2;nms3d_defer_exact=3;nms3d_defer_exact=4")
defaults = {}
for combo in combos.split(";"):
    kv = [c.split("=") for c in combo.split(",") if c]
    for name, _ in kv:
        defaults.setdefault(name, L.sd_get_option(name.encode()))
    for name, v in defaults.items():
        L.sd_set_option(name.encode(), v)
    for name, v in kv:
        _native.check(L.sd_set_option(name.encode(), int(v)))
    run()
    ms = []
    for _ in range(reps):
        dt, keep = run(); ms.append(dt)
    k = keep.cpu().numpy()
    if first is None:
        first = k
    st = _native.last_stats["nms3d"]
    res[combo] = float(np.median(ms))
    print("%-60s median %.2f ms (min %.2f)  N=%d -> %d  rounds %d  %s" % (
        combo or "(defaults)", np.median(ms), min(ms), len(td), int(k.sum()), st[4], "SAME" if np.array_equal(k, first) else "DIFFERENT KEEP FLAGS"), flush=True)
best = min(res, key=res.get)
print("best:", best, "%.2f ms" % res[best])
L.sd_set_option(b"trace", 1)
for combo in dict.fromkeys([combos.split(";")[0], best]):
    for name, v in defaults.items():
        L.sd_set_option(name.encode(), v)
    for name, v in [c.split("=") for c in combo.split(",") if c]:
        L.sd_set_option(name.encode(), int(v))
    print("---- trace:", combo or "(defaults)", flush=True)
    run()
    sys.stdout.flush()
