import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stardist_amd  # sets MIOPEN env
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
for (cin, cout, hw) in [(256, 128, 64), (256, 128, 256), (128, 256, 64), (1, 32, 512), (128, 256, 256)]:
    x = torch.randn(1, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    for det in (False, True):
        torch.backends.cudnn.deterministic = det
        try:
            ys = [F.conv2d(x, w, padding=1) for _ in range(4)]
            torch.cuda.synchronize(); t = time.time()
            for _ in range(10): F.conv2d(x, w, padding=1)
            torch.cuda.synchronize(); dt = (time.time() - t) / 10
            same = all(torch.equal(ys[0], y) for y in ys[1:])
            print("cin=%d cout=%d hw=%d det=%s: identical=%s %.3f ms" % (cin, cout, hw, det, same, dt * 1e3), flush=True)
        except Exception as e:
            print("cin=%d cout=%d hw=%d det=%s: FAILED %s" % (cin, cout, hw, det, str(e)[:80]), flush=True)
