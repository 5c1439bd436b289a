"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE counter_collection csv -> a markdown table of EVERY
kernel of the trace that issued f32 MFMAs: launches, average duration, counted GFLOP (MOPS x 512), TFLOP/s, fraction of the 157.3-TFLOP/s
dense f32 peak, MfmaUtil = MFMA-busy cycles / (active cycles x 1024 SIMD slots), effective clock.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
The first two launches of each kernel are dropped (warm-up).   usage: python profiles/pmc_mfma_table.py <counter_collection.csv>"""
import collections, csv, sys
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").replace("prcnn::", "").split("(")[0]
    per[name][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("| kernel | launches | avg us | GFLOP (counted) | TFLOP/s | of 157.3 | MfmaUtil | clock GHz |\n|---|---|---|---|---|---|---|---|")
rows = []
for name, d in per.items():
    ids = sorted(d, key=int)[2:] or sorted(d, key=int)
    avg = lambda k: sum(d[i][k] for i in ids) / len(ids)
    mops = avg("SQ_INSTS_VALU_MFMA_MOPS_F32")
    if mops <= 0:
        continue
    us = sum(dur[name][i] for i in ids) / len(ids)
    gui = avg("GRBM_GUI_ACTIVE") / 8.0
    gflop = mops * 512.0 / 1e9
    rows.append((gflop / us * 1e3, "| `%s` | %d | %.1f | %.2f | %.1f | %.2f | %.3f | %.2f |" % (
        name[:70], len(ids), us, gflop, gflop / us * 1e3, gflop / us * 1e3 / 157.3, avg("SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 1024.0), gui / us / 1e3)))
for _, ln in sorted(rows, key=lambda x: -float(x[1].split("|")[4])):
    print(ln)
