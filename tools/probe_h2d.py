"""where does the host-input step spend its extra time?  pieces of stardist_amd.utils.to_device on a 2048^2 float32 image"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stardist_amd.utils import to_device
a = np.random.rand(2048, 2048).astype(np.float32)
dev = torch.device("cuda:0")
def T(f, n=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("pageable  torch.as_tensor(a, device)      %.3f ms" % T(lambda: torch.as_tensor(a, device=dev)))
print("to_device (persistent pinned stage)       %.3f ms" % T(lambda: to_device(a, dev)))
h = torch.empty(a.shape, dtype=torch.float32).pin_memory(); t = torch.from_numpy(a)
print("  copy into pinned                        %.3f ms" % T(lambda: h.copy_(t)))
print("  pinned -> device                        %.3f ms" % T(lambda: h.to(dev, non_blocking=True)))
print("  torch.empty(pin_memory=True) + free     %.3f ms" % T(lambda: torch.empty(a.shape, dtype=torch.float32, pin_memory=True)))
print("torch threads", torch.get_num_threads())
