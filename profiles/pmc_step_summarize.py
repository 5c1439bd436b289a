"""Per-kernel HBM traffic of the PRODUCT step from two rocprofv3 PMC passes of bench.py (one per counter: the TCC block
cannot hold FETCH_SIZE and WRITE_SIZE together, MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -- python bench.py ...
    python profiles/pmc_step_summarize.py A/*/*counter_collection.csv B/*/*counter_collection.csv > profiles/r02_pmc_product_kernels.md

FETCH_SIZE is doubled (gfx950 reports 1/2 of the bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is
taken as reported (KB).  Counter values are summed over XCDs per dispatch, medians over the dispatches of a kernel."""
import collections, csv, statistics, sys

def load(path, counter):
    per = collections.defaultdict(float); name = {}; dur = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = r["Dispatch_Id"]
        per[d] += float(r["Counter_Value"])
        name[d] = r["Kernel_Name"]
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by = collections.defaultdict(list); du = collections.defaultdict(list)
    for d, v in per.items():
        by[name[d]].append(v)
        if d in dur:
            du[name[d]].append(dur[d])
    return by, du

fetch, dur_f = load(sys.argv[1], "FETCH_SIZE")
write, dur_w = load(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches per run | median duration under PMC (us) | FETCH_SIZE x2 (MB) | WRITE_SIZE (MB) | HBM traffic (MB) | traffic / duration (GB/s) |")
print("|---|---|---|---|---|---|---|")
rows = []
for k in fetch:
    if not (k.startswith("prcnn::") or k.startswith("void prcnn::")):
        continue
    f = statistics.median(fetch[k]) * 2 * 1024 / 1e6
    w = statistics.median(write.get(k, [0.0])) * 1024 / 1e6
    d = statistics.median(dur_f[k]) if dur_f.get(k) else float("nan")
    rows.append((f + w, k, len(fetch[k]), d, f, w))
for tot, k, n, d, f, w in sorted(rows, reverse=True)[:40]:
    print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.0f |" % (k[:90], n, d, f, w, tot, tot / d * 1e3 if d == d and d > 0 else 0))
