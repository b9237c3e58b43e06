"""profiles/r06_area_band_adversary.txt from the outputs of oracle/_ref/area_band_adversary (gpurun_out/adv7/run6_seed*.txt: the run that a
restart of the session cut short -- only its `new worst` lines carry evaluation counts, lower bounds; run7_seed*.txt: progress lines every
2000 restarts and the closing summary).  usage: python tools/adversary_report.py"""
import glob, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", "adv7")
R5 = 5.46e9                                     # evaluations up to the end of round 5 (profiles/r05_area_band_adversary.txt), round-5 band
lines, tot6, tot7, worst, fam = [], 0, 0, 0.0, {}
for f in sorted(glob.glob(os.path.join(D, "run6_seed*.txt"))):
    t = open(f).read()
    ev = [int(x) for x in re.findall(r"(\d+) evaluations\)", t)]
    w = [float(x) for x in re.findall(r"new worst ([0-9.]+)", t)]
    tot6 += max(ev); worst = max(worst, max(w))
    lines.append("run 6 (cut short after 2 h 45 min by a restart of the session) %s: at least %d evaluations (count at its last `new worst` line), worst %.4f" % (os.path.basename(f), max(ev), max(w)))
done = True
for f in sorted(glob.glob(os.path.join(D, "run7_seed*.txt")) + glob.glob(os.path.join(D, "run8_seed*.txt"))):
    tag = "run 8" if "run8_" in f else "run 7"
    t = open(f).read()
    m = re.search(r"seed (\d+): (\d+) evaluations \((\d+) usable\), worst \|A_clipper - A\| / band = ([0-9.]+)", t)
    if m:
        ev, us, w = int(m.group(2)), int(m.group(3)), float(m.group(4))
        lines.append("%s %s: %d evaluations (%d usable), worst %.4f  [complete]" % (tag, os.path.basename(f), ev, us, w))
        for fm in re.findall(r"^  (.+?)\s+star ([0-9.]+) \((\d+)\)\s+free ([0-9.]+) \((\d+)\)", t, re.M):
            a = fam.setdefault(fm[0].strip(), [0.0, 0, 0.0, 0])
            a[0] = max(a[0], float(fm[1])); a[1] += int(fm[2]); a[2] = max(a[2], float(fm[3])); a[3] += int(fm[4])
    else:
        if tag == "run 8": done = False
        p = re.findall(r"progress: seed \d+, (\d+) restarts, (\d+) evaluations \((\d+) usable\), worst ([0-9.]+) \(star ([0-9.]+), free ([0-9.]+)\)", t)
        ev, us, w = (int(p[-1][1]), int(p[-1][2]), float(p[-1][3])) if p else (0, 0, 0.0)
        wl = [float(x) for x in re.findall(r"new worst ([0-9.]+)", t)]
        w = max([w] + wl)
        lines.append("%s %s: %d evaluations at its last progress line (%d usable), worst %.4f  [%s]" % (tag, os.path.basename(f), ev, us, w, ("cut short by a second restart of the session at %s of 36000 restarts" % (p[-1][0] if p else "0")) if tag == "run 7" else ("running: %s of 10000 restarts" % (p[-1][0] if p else "0"))))
    tot7 += ev; worst = max(worst, w)
hdr = """# Adversarial search against the ROUND-6 decision band of the 2D NMS (NEAR_W 0.15 per near edge pair, STRIP_W 0.45 per strip, robustly-simple rule on)
# oracle/_ref/area_band_adversary <seed> <restarts> 20000 2 1 0.15 0.45 : simulated annealing over NMS-realisable star polygons and over free integer
# polygons (alternating), six families of starting configurations, the vendored Clipper as the judge; score = |A_clipper - A| / band (a violation is > 1).
# The band of round 6 contains the band of round 5 (monotone), so round 5's %.2e evaluations (profiles/r05_area_band_adversary.txt) remain evidence.
""" % R5
out = hdr + "\n".join(lines) + "\n\n"
out += "total against the round-6 band: %.3e evaluations%s (run 6: >= %.2e, runs 7 + 8: %.3e); worst score %.4f of the band; with round 5's: %.2e\n" % (
    tot6 + tot7, "" if done else " SO FAR", tot6, tot7, worst, R5 + tot6 + tot7)
if fam:
    out += "\nper family (complete seeds): worst score (evaluations), star | free\n"
    for k, a in fam.items():
        out += "  %-22s star %.4f (%d)   free %.4f (%d)\n" % (k, a[0], a[1], a[2], a[3])
open(os.path.join(ROOT, "profiles", "r06_area_band_adversary.txt"), "w").write(out)
print(out[-900:])
