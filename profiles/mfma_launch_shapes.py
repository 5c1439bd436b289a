"""rocprofv3 kernel trace (csv) -> the MFMA layer kernels' launches grouped by grid shape: calls, average duration, and -- for the plain
layer kernels, whose grid is (row tiles, N / 128, problems) -- nothing else is known from the trace, so the table is read next to the
engine's shapes.  usage: python profiles/mfma_launch_shapes.py <kernel_trace.csv>"""
import collections
import csv
import os
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = rows[len(rows) // 2:]
steps = max(1, int(os.environ.get("PRCNN_PAIR", "2")) * sum(1 for r in half if "rpn_tail" in r["Kernel_Name"]))    # a tail launch = PRCNN_PAIR batches
by = collections.defaultdict(lambda: [0, 0.0])
for r in half:
    name = r["Kernel_Name"]
    if not any(k in name for k in ("packed_layer", "sa_packed", "sa_wide", "rcnn_entrance", "rpn_tail", "rcnn_point")):
        continue
    key = (name.replace("void ", "").replace("prcnn::", "").split("(")[0][:44], r.get("Stream_Id", "?"),
           "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), r["Workgroup_Size_X"])
    by[key][0] += 1
    by[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("second half of the trace = %d steps\n\n| kernel | stream | grid (threads) | wg | launches per step | avg us | us per step |\n|---|---|---|---|---|---|---|" % steps)
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("| `%s` | %s | %s | %s | %.2f | %.1f | %.1f |" % (k[0], k[1], k[2], k[3], v[0] / steps, v[1] / v[0], v[1] / steps))
