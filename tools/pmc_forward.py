"""Workload for the rocprofv3 passes that tie the convolution kernel to the bench's HIP-event time of the forward pass: the bench
model of one leg (2d: 2048^2 tile, 3d: 256^3 volume), calibrated, then K forward passes and nothing else.  The LAST K x L dispatches of
the split-fp16 convolution kernel (L = launches per forward: 14 in 2D, 10 in 3D) are the K timed passes.
usage: rocprofv3 --kernel-trace [--pmc FETCH_SIZE] --output-format csv -d <dir> -o p -- python tools/pmc_forward.py 2d|3d [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import synth
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D

dev = torch.device("cuda:0")
which = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if which == "2d":
    x = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
    m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(m, x)
    L = 14
else:
    x = torch.from_numpy(synth.s3d_nuclei_image(256, seed=0)).to(dev)
    m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(m, x, frac=0.009, radius=8.5, noise=0.03)
    L = 10
xin = x[..., None]
m._net_forward(xin, sparse_head=True)          # graph capture of the sparse-head form the timed step uses
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(K):
    m._net_forward(xin, sparse_head=True)
b.record(); torch.cuda.synchronize()
macs = bench.net_macs(m, bench.conv_macs_per_input_pixel(m.net, m.config)) if hasattr(bench, "conv_macs_per_input_pixel") else 0
print("FORWARD which=%s K=%d LAUNCHES_PER_FORWARD=%d HIP_EVENT_MS_PER_FORWARD=%.4f" % (which, K, L, a.elapsed_time(b) / K), flush=True)
