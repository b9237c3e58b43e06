"""writes the adversary's totals (profiles/r06_area_band_adversary.txt, made by tools/adversary_report.py) into DESIGN.md / README.md: the tokens
ADV6TOTAL / ADV6WORST / ADVALLTOTAL on the first call, the previously written values on later calls (kept in tools/.adversary_fill.json)"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = open(os.path.join(ROOT, "profiles", "r06_area_band_adversary.txt")).read()
m = re.search(r"total against the round-6 band: ([0-9.e+]+) evaluations( SO FAR)? .*worst score ([0-9.]+) of the band; with round 5's: ([0-9.e+]+)", t)
def fmt(x):
    x = float(x); e = int(("%e" % x).split("e")[1]); return "%.2f × 10%s" % (x / 10 ** e, "".join("⁰¹²³⁴⁵⁶⁷⁸⁹"[int(c)] for c in str(e)))
new = {"ADV6TOTAL": ("≥ " if m.group(2) else "") + fmt(m.group(1)) + (" so far" if m.group(2) else ""), "ADV6WORST": m.group(3), "ADVALLTOTAL": ("≥ " if m.group(2) else "") + fmt(m.group(4))}
state = os.path.join(ROOT, "tools", ".adversary_fill.json")
old = json.load(open(state)) if os.path.exists(state) else {k: k for k in new}
for f in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, f); s = open(p).read()
    for k in new:
        s = s.replace(old[k], new[k]) if old[k] in (k,) or len(old[k]) > 6 else s.replace(old[k], new[k])
    open(p, "w").write(s)
json.dump(new, open(state, "w"))
print(new)
