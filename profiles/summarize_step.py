"""rocprofv3 kernel trace (csv) -> markdown table of ONE steady-state bench step + per-stream busy time.
usage: python profiles/summarize_step.py <kernel_trace.csv> <title> > out.md"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_pruned_kernel<16>" in r["Kernel_Name"]]   # one per step (SA1 FPS)
a, b = marks[-4], marks[-3]                                                               # a late, steady-state step
sel = rows[a:b]
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
by = collections.defaultdict(lambda: [0, 0.0])
stream = collections.defaultdict(float)
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by[r["Kernel_Name"][:100]][0] += 1
    by[r["Kernel_Name"][:100]][1] += d
    stream[r.get("Stream_Id", "?")] += d
print("# %s\n" % sys.argv[2])
print("One steady-state step (batch of 8 scenes; feature stream + geometry side stream overlapping) = window between two "
      "consecutive SA1 FPS launches: wall %.2f ms under the profiler, sum of kernel durations %.2f ms, %d launches; "
      "busy time per stream: %s.\n" % (wall / 1e3, sum(v[1] for v in by.values()) / 1e3, len(sel),
                                       ", ".join("stream %s %.2f ms" % (k, v / 1e3) for k, v in sorted(stream.items()))))
print("| kernel | calls | total us | avg us |\n|---|---|---|---|")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:45]:
    print("| `%s` | %d | %.1f | %.1f |" % (k, v[0], v[1], v[1] / v[0]))
