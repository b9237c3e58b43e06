"""Device timeline of ONE predict_instances step from a rocprofv3 trace: every kernel and every memory copy of the last complete step
in start order, with the idle gap in front of it -- where the device waits for the host (read-backs between greedy rounds, python glue)
shows as gaps, and the copies (`__amd_rocclr_copyBuffer`, H2D / D2H) are attributed to the launch they follow.

  run (GPU box):  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o p -- python tools/step_timeline.py run 2d|3d [steps]
  summarise:      python tools/step_timeline.py report /tmp/tl 2d|3d > profiles/rNN_step_timeline_2d.txt

A step starts at the first-layer kernel of the network (k_conv3_c1x32: once per forward pass); calibration and warm-up steps precede the
reported one in the trace and are ignored."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(which, steps):
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    dev = torch.device("cuda:0")
    if which == "2d":
        x = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
        m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
        bench.calibrate_heads(m, x)
    else:
        x = torch.from_numpy(synth.s3d_nuclei_image(256, seed=0)).to(dev)
        m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        m.thresholds = dict(prob=0.5, nms=0.3)
        bench.calibrate_heads(m, x, frac=0.009, radius=8.5, noise=0.03)
    for _ in range(3 + steps):
        lab, res = m.predict_instances(x)
    torch.cuda.synchronize()
    print(which, len(res["prob"]), "instances")


def short(n):
    for pre in ("void ", "(anonymous namespace)::", "sd::", "at::native::", "sdarea::"):
        n = n.replace(pre, "")
    n = n.split("(")[0]
    return n if len(n) <= 70 else n[:67] + "..."


def report(d, which):
    recs = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            recs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", ""))))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            recs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", "copy " + r.get("Direction", "?").replace("MEMORY_COPY_", ""), r.get("Stream_Id", "")))
    recs.sort()
    starts = [i for i, r in enumerate(recs) if r[2] == "K" and "k_conv3_c1x32" in r[3]]
    if len(starts) < 3:
        print("no complete step found (%d first-layer launches)" % len(starts)); return
    a, b = starts[-2], starts[-1]                       # the last COMPLETE step: [second-to-last first layer, last first layer)
    step = recs[a:b]
    t0 = step[0][0]
    print("## device timeline of one %s predict_instances step (the last complete one in the trace): %d kernels, %d copies, %.3f ms from the first "
          "kernel of this step to the first kernel of the next" % (which, sum(r[2] == "K" for r in step), sum(r[2] == "C" for r in step), (recs[b][0] - t0) / 1e6))
    print("%9s %9s %9s  %s" % ("start ms", "dur us", "gap us", "what (gap = idle time of the device in front of it; overlapping launches on a second stream show a negative gap)"))
    end_prev = t0
    busy = 0.0
    gaps = {}
    agg = {}
    for s, e, kind, name, stream in step:
        gap = (s - end_prev) / 1e3
        print("%9.3f %9.1f %9.1f  %s%s" % ((s - t0) / 1e6, (e - s) / 1e3, gap, name, "" if kind == "K" else "   <-- copy"))
        if gap > 0:
            gaps[name] = gaps.get(name, 0.0) + gap
        busy += max(0, e - max(s, end_prev)) / 1e3
        end_prev = max(end_prev, e)
        c = agg.setdefault(name, [0, 0.0]); c[0] += 1; c[1] += (e - s) / 1e3
    tail = (recs[b][0] - end_prev) / 1e3
    total = (recs[b][0] - t0) / 1e3
    print("\nbusy %.3f ms, idle %.3f ms (of which %.3f ms after the step's last kernel: labels / dict to numpy, python glue of the next call)" % (busy / 1e3, (total - busy) / 1e3, tail / 1e3))
    print("\n### by kernel (this step)")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%9.1f us %4d x  %s" % (t, c, n))
    print("\n### idle time by the launch it precedes (top 25)")
    for n, g in sorted(gaps.items(), key=lambda kv: -kv[1])[:25]:
        print("%9.1f us  before %s" % (g, n))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    else:
        report(sys.argv[2], sys.argv[3])
