"""Per-kernel means of ALL counters of one rocprofv3 PMC pass (csv output).
usage: python tools/pmc_multi.py <dir/prefix> [kernel-substring]"""
import collections, csv, re, sys

prefix = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""


def nm(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n if len(n) < 80 else n[:77] + "..."


agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(prefix + "_counter_collection.csv")):
    k = nm(r["Kernel_Name"])
    if flt and flt not in k:
        continue
    a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print("### `%s`" % k)
    print("| counter | launches | mean per launch |\n|---|---|---|")
    for c, (n, tot) in sorted(cs.items()):
        print("| %s | %d | %.1f |" % (c, n, tot / n))
