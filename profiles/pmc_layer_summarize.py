"""rocprofv3 --pmc csv of profiles/layer_probe.py -> per-shape MFMA utilisation and effective clock of packed_layer_kernel.
usage: python profiles/pmc_layer_summarize.py <counter_collection.csv>"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "packed_layer_kernel" in r["Kernel_Name"]]
per = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
for r in rows:
    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[r["Dispatch_Id"]] = (r["Grid_Size"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
groups = collections.OrderedDict()
for i in sorted(per, key=int):
    groups.setdefault(meta[i][0], []).append(i)
names = sorted(per[next(iter(per))])
print("| grid (threads) | launches | us | " + " | ".join(names) + " | derived |"); print("|---|---|---|" + "---|" * (len(names) + 1))
for gsz, ids in groups.items():
    ids = ids[3:] if len(ids) > 3 else ids
    d = sum(meta[i][1] for i in ids) / len(ids)
    avg = {k: sum(per[i][k] for i in ids) / len(ids) for k in names}
    der = ""
    if "GRBM_GUI_ACTIVE" in avg:
        gui = avg["GRBM_GUI_ACTIVE"] / 8.0
        der = "clock %.2f GHz" % (gui / d / 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            der += ", MfmaUtil %.1f %%" % (100.0 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024))
    print("| %s | %d | %.1f | " % (gsz, len(ids), d) + " | ".join("%.4g" % avg[k] for k in names) + " | " + der + " |")
