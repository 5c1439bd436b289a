"""rocprofv3 kernel trace (csv) of profiles/pmc_step_probe.py -- the product step's kernels on ONE stream, nothing beside them -- ->
average duration per kernel over the last two thirds of its launches: what a kernel takes ALONE, to set against its duration in the
pipelined step (profiles/r04_bench_step_kernel_stats.md).
usage: rocprofv3 --kernel-trace --output-format csv -d out -- python profiles/pmc_step_probe.py 6; python profiles/solo_kernel_times.py out/*/*kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("prcnn::", "").split("(")[0][:48]
    by[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | launches counted | average us alone |\n|---|---|---|")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    v = v[len(v) // 3:]
    print("| `%s` | %d | %.1f |" % (k, len(v), sum(v) / len(v)))
