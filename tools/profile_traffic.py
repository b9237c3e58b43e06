"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass) of tools/pmc_predict.py into
profiles/pair_kernel_traffic.json, the file bench.py reads `roofline.traffic` from, and a markdown table of every hand-written kernel.

usage: python tools/profile_traffic.py <fetch csv prefix> <write csv prefix> <pmc_predict log> <out.json> <out.md>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide
(16 B/lane) streaming reads by 2x; this kernel's loads are 1-8 byte gathers, for which the counter is uncalibrated -- the raw value
is stored, and 2x raw as the upper bound."""
import collections, csv, json, re, sys

fpre, wpre, log, out_json, out_md = sys.argv[1:6]
tag = sys.argv[6] if len(sys.argv) > 6 else "rNN"


def nm(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def load(prefix, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(prefix + "_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            a = agg[nm(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


fe, wr = load(fpre, "FETCH_SIZE"), load(wpre, "WRITE_SIZE")
m = re.search(r"PAIRS_PER_STEP=(\d+) PAIR_LAUNCHES_PER_STEP=(\d+) GENERAL_PATH_PAIRS=(\d+) SIZE=(\d+)", open(log).read())
pairs, launches, general, size = (int(v) for v in m.groups())
key = [k for k in fe if k.startswith("k_pairs_beam<32, 8")][0]
f_kib = fe[key][1] / fe[key][0]; w_kib = wr[key][1] / wr[key][0]
doc = {"kernel": key, "size": size, "pairs_per_step": pairs, "pair_launches_per_step": launches, "general_path_pairs_per_step": general,
       "fetch_bytes_per_launch_raw": f_kib * 1024, "write_bytes_per_launch_raw": w_kib * 1024,
       "bytes_per_launch": (f_kib + w_kib) * 1024, "bytes_per_launch_upper": (2 * f_kib + w_kib) * 1024,
       "algorithmic_bytes_per_launch": 272.0 * pairs / launches,
       "traffic_over_algorithmic": (f_kib + w_kib) * 1024 / (272.0 * pairs / launches),
       "traffic_over_algorithmic_upper": (2 * f_kib + w_kib) * 1024 / (272.0 * pairs / launches),
       "source": "profiles/%s_pmc_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_predict.py, raw counters)" % tag}
json.dump(doc, open(out_json, "w"), indent=1)
with open(out_md, "w") as fh:
    fh.write("| kernel | launches | FETCH_SIZE MiB / launch (raw) | WRITE_SIZE MiB / launch |\n|---|---|---|---|\n")
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [1, 0])[1] + wr.get(k, [1, 0])[1])):
        if not (k.startswith("k_") or "sd" in k):
            continue
        a, b = fe.get(k, [0, 0.0]), wr.get(k, [0, 0.0])
        fh.write("| `%s` | %d | %.3f | %.3f |\n" % (k[:90], max(a[0], b[0]), a[1] / max(1, a[0]) / 1024, b[1] / max(1, b[0]) / 1024))
    fh.write("\npair kernel: %d pairs / step in %d launches; algorithmic 272 B/pair = %.1f MiB / launch; measured (raw) %.1f MiB / launch = %.1fx "
             "(upper bound with the 2x FETCH correction: %.1fx)\n" % (pairs, launches, doc["algorithmic_bytes_per_launch"] / 2**20, doc["bytes_per_launch"] / 2**20,
                                                                      doc["traffic_over_algorithmic"], doc["traffic_over_algorithmic_upper"]))
print(json.dumps(doc))
