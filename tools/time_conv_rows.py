"""time sd_conv3_f16x3_rows_device (the sparse path's features on the candidate pixels) at the bench's sizes.  usage: python tools/time_conv_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa
import numpy as np, torch
from stardist_amd.models import unet as U
dev = torch.device("cuda:0")
for nd, S, n in ((2, (2048, 2048), 418577), (3, (256, 256, 256), 165227)):
    cl = torch.channels_last if nd == 2 else torch.channels_last_3d
    conv = (torch.nn.Conv2d if nd == 2 else torch.nn.Conv3d)(32, 128, 3, padding=1).to(dev)
    x = torch.randn((1, 32) + S, device=dev).contiguous(memory_format=cl)
    xs = U.split16_pack(x)
    rows = torch.sort(torch.randperm(int(np.prod(S)), device=dev)[:n])[0]
    for tag, r in (("spatial order", rows), ("random order", rows[torch.randperm(n, device=dev)])):
        with torch.no_grad(), U.force_conv_mode("f16x3"):
            U.conv_rows(conv, xs, 1, r); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                U.conv_rows(conv, xs, 1, r)
            b.record(); torch.cuda.synchronize()
        print("%dD %d rows, %s: %.1f us" % (nd, n, tag, a.elapsed_time(b) / 5 * 1e3), flush=True)
