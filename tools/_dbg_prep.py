import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stardist_amd.lib import _native as N
x = np.array([[56, 56, 55, 55, 56]] * 64, np.int32); y = np.array([[54, 54, 54, 53, 53]] * 64, np.int32)
dev = torch.device("cuda")
tx, ty = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
out = torch.zeros(64 * 400, dtype=torch.uint8, device=dev)
N.check(N.lib().sd_prepare_polys_device(N.tptr(tx), N.tptr(ty), 64, 5, N.tptr(out), 64 * 400, N.current_stream()))
torch.cuda.synchronize()
