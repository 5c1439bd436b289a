"""rocprofv3 kernel trace (csv) -> markdown table of one steady-state window of the bench + per-stream busy time.
The window runs between two consecutive SA1 FPS launches (fps_pruned_kernel<16>): since round 2 a geometry chain covers a
GROUP of batches (PRCNN_GEO_GROUP, default 4), so the window holds that many steps; figures are divided by the number of
RPN stages (sa_xyz_mlp_kernel<32,...> launches) found in it.
usage: python profiles/summarize_step.py <kernel_trace.csv> <title> > out.md"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_pruned_kernel<16" in r["Kernel_Name"]]   # one per step (SA1 FPS)
# a late window that holds a whole group of steps (at the cold start of a closed timed run two chains are launched back to back)
pairs = [(a, b) for a, b in zip(marks[:-1], marks[1:])
         if sum(1 for r in rows[a:b] if "sa_xyz_mlp" in r["Kernel_Name"] and "<32" in r["Kernel_Name"]) >= 3]
a, b = pairs[-1]
sel = rows[a:b]
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
by = collections.defaultdict(lambda: [0, 0.0])
stream = collections.defaultdict(float)
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by[r["Kernel_Name"][:100]][0] += 1
    by[r["Kernel_Name"][:100]][1] += d
    stream[r.get("Stream_Id", "?")] += d
steps = max(1, sum(1 for r in sel if "sa_xyz_mlp" in r["Kernel_Name"] and "<32" in r["Kernel_Name"]))
print("# %s\n" % sys.argv[2])
print("Window between two consecutive SA1 FPS launches = one geometry group = %d steps (batches of 8 scenes).  PER STEP: wall "
      "%.2f ms under the profiler (the profiler makes the run host-bound; unprofiled step time is in the bench line), sum of "
      "kernel durations %.2f ms, %.0f launches; busy time per stream: %s.\n"
      % (steps, wall / 1e3 / steps, sum(v[1] for v in by.values()) / 1e3 / steps, len(sel) / steps,
         ", ".join("stream %s %.2f ms" % (k, v / 1e3 / steps) for k, v in sorted(stream.items()))))
print("| kernel | calls per step | total us per step | avg us per call |\n|---|---|---|---|")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:50]:
    print("| `%s` | %.2f | %.1f | %.1f |" % (k, v[0] / steps, v[1] / steps, v[1] / v[0]))
