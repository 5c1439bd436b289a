"""Average rocprofv3 --pmc counters per launch of kernels whose name contains a substring.
usage: python profiles/pmc_generic.py <counter_collection.csv> <substring>"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for r in rows:
    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(per, key=int)[2:]
print("launches", len(ids), "avg dur us %.1f" % (sum(dur[i] for i in ids) / len(ids)))
for k in per[ids[0]]:
    print("%-28s %.4g" % (k, sum(per[i][k] for i in ids) / len(ids)))
