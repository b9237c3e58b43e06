"""Per-kernel mean of one rocprofv3 PMC counter (csv output).  usage: python tools/pmc_summary.py <dir/prefix> COUNTER [unit_div]
Reads <prefix>_counter_collection.csv (+ <prefix>_kernel_trace.csv for durations) and prints a markdown table."""
import collections, csv, re, sys

prefix, counter = sys.argv[1], sys.argv[2]
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0


def nm(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n if len(n) < 80 else n[:77] + "..."


dur = collections.defaultdict(lambda: [0, 0])
try:
    for r in csv.DictReader(open(prefix + "_kernel_trace.csv")):
        d = dur[nm(r["Kernel_Name"])]; d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
except FileNotFoundError:
    pass
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(prefix + "_counter_collection.csv")):
    if r["Counter_Name"] != counter:
        continue
    a = agg[nm(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("| kernel | launches | %s per launch | avg duration us (under the profiler) |\n|---|---|---|---|" % counter)
for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    d = dur.get(k)
    print("| `%s` | %d | %.3f | %s |" % (k, n, tot / n / div, ("%.1f" % (d[1] / d[0] / 1e3)) if d and d[0] else "-"))
