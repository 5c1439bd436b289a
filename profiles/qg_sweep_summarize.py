"""Per-kernel durations (rocprofv3 --kernel-trace) and HBM traffic (--pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950, both counters in KB) of profiles/qg_sweep.py, per configuration.
usage: python profiles/qg_sweep_summarize.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> [reps] [json out] [scene kind]
The JSON (profiles/r06_pmc_query_and_group_{uniform,lidar}.json; round 4: r04_pmc_query_and_group.json) holds the B = 8, C = 128,
r = 0.2 configurations at nsample = 32 (top level: bench.py's roofline_reference_op reads its traffic from it) and nsample = 64
("ns64")."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
CONFIGS = [(B, R, NS, C) for B in (1, 8) for R in (0.1, 0.2, 0.4) for NS in (32, 64) for C in (0, 1, 128)]
N, M = 16384, 4096
alg = lambda B, C, NS: B * (12 * N + 12 * M + 4 * C * N + 4 * M * NS + 4 * (3 + C) * M * NS)
OURS = ("dense_build", "dense_query_kernel", "group_cat", "ball_query_kernel")


def short(name):
    for k in ("dense_build_reg_kernel", "dense_build_kernel", "dense_query_kernel", "group_cat_lds_kernel", "group_cat_kernel", "ball_query_kernel"):
        if k in name:
            return k
    return name[:40]


def calls_of(rows, start_key="Start_Timestamp"):
    """rows of OUR kernels after the set-up (FPS etc.), in dispatch order, cut into calls: a call starts at its ball-query build"""
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in OURS)]
    rows.sort(key=lambda r: int(r[start_key]) if r.get(start_key) else int(r["Dispatch_Id"]))
    calls = []
    for r in rows:
        if "dense_build" in r["Kernel_Name"]:
            calls.append([])
        if calls:
            calls[-1].append(r)
    return calls


trace = list(csv.DictReader(open(sys.argv[1])))
calls = calls_of(trace)
per = 1 + reps
assert len(calls) == per * len(CONFIGS), (len(calls), per * len(CONFIGS))


def pmc_calls(path, counter):
    per_d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not any(k in r["Kernel_Name"] for k in OURS):
            continue
        d = per_d.setdefault(r["Dispatch_Id"], {"Kernel_Name": r["Kernel_Name"], "Dispatch_Id": r["Dispatch_Id"], "v": 0.0,
                                                 "Start_Timestamp": r.get("Start_Timestamp", "")})
        d["v"] += float(r["Counter_Value"])
    return calls_of(list(per_d.values()))


fetch = pmc_calls(sys.argv[2], "FETCH_SIZE") if len(sys.argv) > 2 and os.path.exists(sys.argv[2]) else None
write = pmc_calls(sys.argv[3], "WRITE_SIZE") if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else None
print("| B | r | nsample | C | kernel | avg duration (us) | FETCH_SIZE x2 (MB) | WRITE_SIZE (MB) | HBM traffic (MB) |")
print("|---|---|---|---|---|---|---|---|---|")
for k, (B, R, NS, C) in enumerate(CONFIGS):
    mine = calls[k * per + 1:(k + 1) * per]
    names = [short(r["Kernel_Name"]) for r in mine[0]]
    tot_d = tot_t = 0.0
    detail = []
    for j, nm in enumerate(names):
        d = sum((int(c[j]["End_Timestamp"]) - int(c[j]["Start_Timestamp"])) / 1e3 for c in mine) / len(mine)
        f = w = None
        if fetch and len(fetch) == len(calls):
            f = sum(c[j]["v"] for c in fetch[k * per + 1:(k + 1) * per]) / reps * 2 * 1024 / 1e6
        if write and len(write) == len(calls):
            w = sum(c[j]["v"] for c in write[k * per + 1:(k + 1) * per]) / reps * 1024 / 1e6
        tot_d += d
        tot_t += (f or 0) + (w or 0)
        detail.append({"kernel": nm, "avg_us": round(d, 1), "fetch_MB_x2": None if f is None else round(f, 2), "write_MB": None if w is None else round(w, 2)})
        print("| %d | %g | %d | %d | `%s` | %.1f | %s | %s | %s |" % (B, R, NS, C, nm, d, "%.2f" % f if f is not None else "-",
                                                                 "%.2f" % w if w is not None else "-", "%.2f" % (f + w) if f is not None and w is not None else "-"))
    a = alg(B, C, NS)
    print("| %d | %g | %d | %d | **sum** (algorithmic %.2f MB) | **%.1f** = %.0f GB/s = %.3f of 8 TB/s | | | %s |" % (
        B, R, NS, C, a / 1e6, tot_d, a / tot_d / 1e3, a / tot_d / 1e3 / 8000, ("**%.2f** = %.2f x algorithmic" % (tot_t, tot_t * 1e6 / a)) if tot_t else "-"))
    if (B, R, C) == (8, 0.2, 128) and len(sys.argv) > 5 and tot_t:
        full = {"dense_build_reg_kernel": "dense_build_reg_kernel<16>", "group_cat_lds_kernel": "group_cat_lds_kernel<1, true>"}
        kind = sys.argv[6] if len(sys.argv) > 6 else "uniform"
        rec = {"what": "prcnn_query_and_group, %s scenes, B = 8, N = 16384, M = 4096, C = 128, nsample = %d, r = 0.2: rocprofv3 --kernel-trace averages and "
                       "--pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE passes over profiles/qg_sweep.py (profiles/measure_r06.sh qg)" % (kind, NS),
               "scene": kind, "kernels": " + ".join(full.get(n, n) for n in names), "algorithmic_bytes_per_launch": a,
               "hbm_traffic_bytes_per_launch": int(tot_t * 1e6), "sum_of_kernel_durations_us": round(tot_d, 1), "per_kernel": detail}
        if NS == 32:
            OUT = rec
        else:
            OUT["ns64"] = rec
            json.dump(OUT, open(sys.argv[5], "w"), indent=1)
