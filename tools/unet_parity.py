"""U-Net on the MI355X (MIOpen/CK fp32 kernels, HIP graph, fused bias+activation epilogue, z-slab head) against the same module on
the CPU in float32 (oneDNN): max |d prob|, max relative |d dist|, run-to-run determinism, and the cost of deterministic mode.
usage: python tools/unet_parity.py"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from oracle import synth
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D

dev = torch.device("cuda:0")


def run(tag, make, img, calib):
    for det in (False, True):
        torch.backends.cudnn.deterministic = det
        m = make(dev)
        bench.calibrate_heads(m, torch.from_numpy(img).to(dev), **calib)
        p1, d1 = m.predict(img)[:2]
        p2, d2 = m.predict(img)[:2]
        same = np.array_equal(p1, p2) and np.array_equal(d1, d2)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(3): m.predict(torch.from_numpy(img).to(dev))
        torch.cuda.synchronize(); dt = (time.time() - t) / 3
        mc = make("cpu")
        mc.net.load_state_dict({k: v.cpu() for k, v in m.net.state_dict().items()})
        pc, dc = mc.predict(img)[:2]
        print("%s deterministic=%s: run-to-run identical=%s  max|dprob|=%.3g  max|ddist|=%.3g  max rel ddist=%.3g  (dist~%.1f)  predict %.1f ms" % (
            tag, det, same, np.abs(p1 - pc).max(), np.abs(d1 - dc).max(), (np.abs(d1 - dc) / np.maximum(np.abs(dc), 1e-3)).max(), np.abs(dc).mean(), dt * 1e3), flush=True)


run("2D unet 512^2", lambda d: StarDist2D(Config2D(n_rays=32), basedir=None, device=d, seed=0), synth.s2d_nuclei_image(512, 512, seed=1), dict())
run("3D unet 64^3", lambda d: StarDist3D(Config3D(rays=96), basedir=None, device=d, seed=0), synth.s3d_nuclei_image(64, seed=1), dict(frac=0.02, radius=8.5, noise=0.03))
run("3D resnet 64^3 grid(1,2,2)", lambda d: StarDist3D(Config3D(rays=96, backbone="resnet", grid=(1, 2, 2)), basedir=None, device=d, seed=0),
    synth.s3d_nuclei_image(64, seed=2), dict(frac=0.02, radius=8.5, noise=0.03))
