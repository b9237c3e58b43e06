"""Which layers of the StarDist network are not run-to-run identical on the GPU?  (eager forward twice, per-module outputs compared)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn as nn
from oracle import synth
from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
dev = torch.device("cuda:0")


def probe(tag, m, x):
    outs = [{}, {}]
    for r in range(2):
        hooks = []
        for name, mod in m.net.named_modules():
            if isinstance(mod, (nn.Conv2d, nn.Conv3d)):
                hooks.append(mod.register_forward_hook(lambda mo, i, o, name=name, r=r: outs[r].__setitem__(name, (o.detach().clone(), tuple(i[0].shape), tuple(mo.weight.shape)))))
        with torch.no_grad(), torch.enable_grad():
            pass
        with torch.enable_grad():      # plain module path so that hooks fire
            m.net(x)
        for h in hooks: h.remove()
    for name in outs[0]:
        a, sh, w = outs[0][name]; b = outs[1][name][0]
        if not torch.equal(a, b):
            print("%s: %-28s in %s w %s  max|d|=%.3g" % (tag, name, sh, w, (a - b).abs().max().item()), flush=True)
    print(tag, "done", flush=True)


m2 = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
x = torch.from_numpy(synth.s2d_nuclei_image(512, 512, seed=1)).to(dev)[None, None].contiguous(memory_format=torch.channels_last)
probe("2D", m2, x)
m3 = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
x3 = torch.from_numpy(synth.s3d_nuclei_image(64, seed=1)).to(dev)[None, None].contiguous(memory_format=torch.channels_last_3d)
probe("3D", m3, x3)
