import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stardist_amd
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
def gemm_conv(x, w):
    n, c, h, ww = x.shape
    cols = F.unfold(x, 3, padding=1)                       # (n, c*9, h*w)
    y = torch.matmul(w.reshape(w.shape[0], -1), cols)      # (n, cout, h*w)
    return y.reshape(n, w.shape[0], h, ww)
for (cin, cout, hw) in [(256, 128, 64), (128, 256, 64), (256, 128, 32), (512, 256, 32), (128, 64, 128), (64, 128, 128)]:
    x = torch.randn(1, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev)
    ys = [gemm_conv(x, w) for _ in range(6)]
    ref = F.conv2d(x, w, padding=1)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): gemm_conv(x, w)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    cs = [F.conv2d(x, w, padding=1) for _ in range(6)]
    print("cin=%d cout=%d hw=%d: gemm identical=%s  conv identical=%s  max|gemm-conv|=%.3g (|y|~%.1f)  %.3f ms" % (cin, cout, hw, all(torch.equal(ys[0], y) for y in ys[1:]),
          all(torch.equal(cs[0], y) for y in cs[1:]), (ys[0] - ref).abs().max().item(), ref.abs().mean().item(), dt * 1e3), flush=True)
