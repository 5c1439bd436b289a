"""rocprofv3 kernel trace (csv) -> markdown table of one steady-state window of the bench + per-stream busy time.
The window runs between SA1 FPS launches (fps_pruned_kernel<16>: one per geometry GROUP of batches, PRCNN_GEO_GROUP = 4) in
the second half of the trace; figures are divided by the number of steps = rpn_tail_kernel launches found in it.
usage: python profiles/summarize_step.py <kernel_trace.csv> <title> > out.md"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_pruned_kernel<16" in r["Kernel_Name"] or "fps_spec_kernel<16" in r["Kernel_Name"]
         or "fps_spec2_kernel" in r["Kernel_Name"]]            # (fps_spec2_kernel: double.yaml's 32768-point clouds)   # one per geometry group (SA1 FPS)
# steady-state window: from the second sampling launch after the middle of the trace to the last one; a STEP = one launch of
# the fused RPN tail (one per batch).  (Round 2 counted sa_xyz_mlp<32> launches, which moved into the group chain -- one per
# group -- at the end of that round: the window search found nothing and the committed table was empty.)
STEP_KERNEL = "rpn_tail"          # rpn_tail_kernel or rpn_tail_lin_kernel
assert len(marks) >= 3, "need at least three geometry groups in the trace"
a, b = marks[max(1, len(marks) // 2)], marks[-1]
sel = rows[a:b]
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
by = collections.defaultdict(lambda: [0, 0.0])
stream = collections.defaultdict(float)
per_stream = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by[r["Kernel_Name"][:100]][0] += 1
    by[r["Kernel_Name"][:100]][1] += d
    stream[r.get("Stream_Id", "?")] += d
    e = per_stream[r.get("Stream_Id", "?")][r["Kernel_Name"].replace("void ", "").replace("prcnn::", "").split("(")[0][:60]]
    e[0] += 1; e[1] += d
# (second session of round 4: PAIRS of batches share a launch of every stage behind the geometry -- one tail launch = PRCNN_PAIR steps)
import os
PAIR = int(os.environ.get("PRCNN_PAIR", "2"))
steps = max(1, PAIR * sum(1 for r in sel if STEP_KERNEL in r["Kernel_Name"]))
print("# %s\n" % sys.argv[2])
head = ("Window between SA1 FPS launches (whole geometry groups) = %%d steps (batches of 8 scenes; %d batches per launch of the stages "
        "behind the geometry).  PER STEP: wall %%.2f ms under the profiler (the profiler makes the run host-bound; unprofiled step time is "
        "in the bench line), sum of kernel durations %%.2f ms, %%.0f launches; busy time per stream: %%s.\n" % PAIR)
print(head % (steps, wall / 1e3 / steps, sum(v[1] for v in by.values()) / 1e3 / steps, len(sel) / steps,
              ", ".join("stream %s %.2f ms" % (k, v / 1e3 / steps) for k, v in sorted(stream.items()))))
print("| kernel | calls per step | total us per step | avg us per call |\n|---|---|---|---|")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:50]:
    print("| `%s` | %.2f | %.1f | %.1f |" % (k, v[0] / steps, v[1] / steps, v[1] / v[0]))
print("\n## The same window by stream (kernels of at least 10 us per step, and every launch that is not one of the library's)\n")
for sid, ks in sorted(per_stream.items()):
    print("* stream %s, %.2f ms per step: %s" % (sid, stream[sid] / 1e3 / steps, ", ".join(
        "`%s` %.0f us (%.2f x)" % (k, v[1] / steps, v[0] / steps) for k, v in sorted(ks.items(), key=lambda kv: -kv[1][1]) if v[1] / steps >= 10 or "at::" in k or "rocclr" in k)))
