"""Per-round counters of the 2D NMS (option trace) under contention: P processes x REPS calls on the bench's candidates; tallies the distinct
'round k' lines.  usage: python tools/contention_trace.py P REPS"""
import os, sys, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "child":
    P, reps = int(sys.argv[1]), int(sys.argv[2])
    env = dict(os.environ, SD_TRACE="1")        # (SD_LIB / SD_OPTS are passed on to tools/time_nms2d_bench.py)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "time_nms2d_bench.py"), str(reps)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(P)]
    tally = collections.Counter()
    for p in procs:
        out, _ = p.communicate(timeout=900)
        for l in out.splitlines():
            if l.startswith("round") or l.startswith("tail batch"):
                l = l.split(" pair_kernel=")[0]
                tally[l] += 1
            elif l.startswith("probe"):
                if ": 0 pairs with a different flag" in l:
                    tally["probe " + l.split(":")[0].split()[-2] + " " + l.split(":")[0].split()[-1] + ": second evaluation agrees"] += 1
                else:
                    print(l[:400])
            elif l.startswith("rep"):
                tally["keep crc " + l.split("keep crc ")[1]] += 1
    for l, c in sorted(tally.items()):
        print("%5d x  %s" % (c, l))
