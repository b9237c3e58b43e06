"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table.

usage: python tools/rocpd_summary.py <results.db> [--md]
Equivalent of `rocprofv3 --stats` kernel_stats.csv: name, calls, total ns, avg ns, min, max, %.
"""
import sqlite3
import sys


def summarise(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot) for r in rows], cols


if __name__ == "__main__":
    rows, _ = summarise(sys.argv[1])
    md = "--md" in sys.argv
    if md:
        print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    # --group SUBSTR N: one more line for a kernel launched N times per step with different shapes (the hand-written convolution:
    # 14 launches per 2D forward pass), so that its per-step total can be held against bench.py's HIP-event time of the forward pass
    if "--group" in sys.argv:
        i = sys.argv.index("--group"); sub, per = sys.argv[i + 1], int(sys.argv[i + 2])
        sel = [r for r in rows if sub in r[0]]
        calls, tot = sum(r[1] for r in sel), sum(r[2] for r in sel)
        if calls:
            print("%s`*%s*`: %d launches, %.3f ms in total = %.3f ms per group of %d launches (one forward pass)%s"
                  % ("" if not md else "\n", sub, calls, tot / 1e6, tot / 1e6 / (calls / float(per)), per, "\n" if md else ""))
    for n, calls, t, a, mn, mx, pc in rows:
        n = n if len(n) < 90 else n[:87] + "..."
        if md:
            print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (n, calls, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, pc))
        else:
            print("%-90s %6d %10.3f ms %10.1f us %6.1f%%" % (n, calls, t / 1e6, a / 1e3, pc))
