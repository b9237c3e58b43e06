"""host-side profile (cProfile) of predict_instances: which Python / torch calls cost wall time between the kernels.
usage: python tools/pyprof_step.py 2d|3d"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import synth
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
dev = torch.device("cuda:0")
if sys.argv[1] == "2d":
    x = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
    m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(m, x)
else:
    x = torch.from_numpy(synth.s3d_nuclei_image(256, seed=0)).to(dev)
    m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
    m.thresholds = dict(prob=0.5, nms=0.3)
    bench.calibrate_heads(m, x, frac=0.009, radius=8.5, noise=0.03)
for _ in range(3):
    m.predict_instances(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    m.predict_instances(x)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
