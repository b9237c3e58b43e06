"""Per-forward duration and HBM traffic of the split-fp16 convolution kernel from rocprofv3 runs of tools/pmc_forward.py:
the last K x L dispatches of k_conv3_f16 are the K timed forward passes (L launches each).

usage: python tools/conv_traffic.py <which 2d|3d> <kernel-trace csv prefix> <fetch csv prefix> <write csv prefix> <pmc_forward log> <out.json> <out.md>
Writes / updates <out.json> (profiles/conv_kernel_traffic.json: bench.py reads roofline_convs.traffic from it) and appends the per-layer
table to <out.md>.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads
(MI355X_MICROARCH.md, HBM section): the kernel's loads are exactly those, so fetch is doubled ("corrected"), raw kept beside it."""
import csv, json, os, re, sys

which, ktpre, fpre, wpre, log, out_json, out_md = sys.argv[1:8]
m = re.search(r"FORWARD which=(\S+) K=(\d+) LAUNCHES_PER_FORWARD=(\d+) HIP_EVENT_MS_PER_FORWARD=([0-9.]+)", open(log).read())
K, L, ev_ms = int(m.group(2)), int(m.group(3)), float(m.group(4))


def rows(path, want_counter=None):
    out = []
    for r in csv.DictReader(open(path)):
        if "k_conv3_f16" not in r["Kernel_Name"]:
            continue
        if want_counter and r.get("Counter_Name") != want_counter:
            continue
        out.append(r)
    out.sort(key=lambda r: int(r["Dispatch_Id"]))
    return out[-K * L:]


kt = rows(ktpre + "_kernel_trace.csv")
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in kt]          # ms
fe = [float(r["Counter_Value"]) * 1024 for r in rows(fpre + "_counter_collection.csv", "FETCH_SIZE")]
wr = [float(r["Counter_Value"]) * 1024 for r in rows(wpre + "_counter_collection.csv", "WRITE_SIZE")]
assert len(dur) == len(fe) == len(wr) == K * L, (len(dur), len(fe), len(wr), K, L)
per_layer = []
for l in range(L):
    d = sum(dur[k * L + l] for k in range(K)) / K
    f = sum(fe[k * L + l] for k in range(K)) / K
    w = sum(wr[k * L + l] for k in range(K)) / K
    per_layer.append((d, f, w))
doc = json.load(open(out_json)) if os.path.exists(out_json) else {}
doc[which] = {"kernel": "k_conv3_f16", "launches_per_forward": L, "forwards_profiled": K,
              "kernel_ms_per_forward": round(sum(d for d, _, _ in per_layer), 4), "hip_event_ms_per_forward_unprofiled_run": ev_ms,
              "fetch_bytes_per_forward_raw": sum(f for _, f, _ in per_layer), "write_bytes_per_forward_raw": sum(w for _, _, w in per_layer),
              "bytes_per_forward": sum(2 * f + w for _, f, w in per_layer),
              "source": "rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/pmc_forward.py %s; fetch doubled "
                        "(gfx950: FETCH_SIZE counts 64 B per 128-B request of wide streaming reads)" % which}
json.dump(doc, open(out_json, "w"), indent=1)
with open(out_md, "a") as fh:
    fh.write("\n## k_conv3_f16, %s forward pass (%d launches; mean of the last %d passes of tools/pmc_forward.py %s)\n\n" % (which, L, K, which))
    fh.write("| launch | duration ms | FETCH_SIZE MiB (raw) | WRITE_SIZE MiB | corrected traffic MiB | GB/s |\n|---|---|---|---|---|---|\n")
    for l, (d, f, w) in enumerate(per_layer):
        fh.write("| %d | %.3f | %.1f | %.1f | %.1f | %.0f |\n" % (l, d, f / 2 ** 20, w / 2 ** 20, (2 * f + w) / 2 ** 20, (2 * f + w) / (d * 1e-3) / 1e9))
    fh.write("| all %d | **%.3f** | %.1f | %.1f | %.1f | |\n" % (L, doc[which]["kernel_ms_per_forward"], doc[which]["fetch_bytes_per_forward_raw"] / 2 ** 20,
                                                                doc[which]["write_bytes_per_forward_raw"] / 2 ** 20, doc[which]["bytes_per_forward"] / 2 ** 20))
    fh.write("\nHIP-event time of one forward pass (whole graph: + first layer, pools, head pass) in the unprofiled part of the same run: %.3f ms\n" % ev_ms)
print(json.dumps(doc[which]))
