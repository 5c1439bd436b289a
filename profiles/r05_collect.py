"""gpurun_out/r05/* (written by profiles/measure_r05.sh on the GPU box) -> the tracked profiles/r05_* artefacts.
  python profiles/r05_collect.py"""
import os, re, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r05"), os.path.join(ROOT, "profiles")
COPY = {"bench_k20.json": "r05_bench_line.json", "bench_k100.json": "r05_bench_line_k100.json",
        "step_uniform.md": "r05_bench_step_kernel_stats_uniform.md", "step_lidar.md": "r05_bench_step_kernel_stats_lidar.md",
        "kernel_stats_uniform.csv": "r05_bench_kernel_stats.csv", "kernel_stats_lidar.csv": "r05_bench_kernel_stats_lidar.csv",
        "bench_k20_kernel_stats.csv": "r05_bench_k20_kernel_stats.csv", "mfma_shapes_uniform.md": "r05_mfma_launch_shapes_uniform.md",
        "mfma_shapes_lidar.md": "r05_mfma_launch_shapes_lidar.md", "pmc_product_kernels.md": "r05_pmc_product_kernels.md",
        "qg_sweep_uniform.md": "r05_qg_sweep_uniform.md", "kernel_probes.txt": "r05_kernel_probes.txt", "sensitivity.txt": "r05_sensitivity.txt",
        "fps_stamps.txt": "r05_fps_round_stamps.txt", "occupancy.md": "r05_occupancy.md", "solo_kernel_times.md": "r05_solo_kernel_times.md",
        "ball_pack_probe.md": "r05_ball_pack_probe.md"}
for a, b in COPY.items():
    shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))

# MFMA counters: the table from the raw per-kernel averages (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
raw = open(os.path.join(SRC, "pmc_mfma_product_kernels.txt")).read()
rows = []
for blk in raw.split("## ")[1:]:
    name = blk.splitlines()[0].strip()
    m = re.search(r"launches (\d+) avg dur us ([\d.]+)", blk)
    if not m:
        continue
    val = {k: float(v) for k, v in re.findall(r"^(\w+)\s+([\d.e+]+)$", blk, re.M)}
    n, dur = int(m.group(1)), float(m.group(2))
    gui = val["GRBM_GUI_ACTIVE"] / 8.0
    gflop = val["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0 / 1e9
    tf = gflop / dur * 1e-3 * 1e3 / 1e3 * 1e3                                  # GFLOP / us = PFLOP/s -> TFLOP/s
    rows.append("| `%s` | %d | %.1f | %.2f | %.1f | %.2f | %.3f | %.2f |" % (name, n, dur, gflop, gflop / dur * 1e3, gflop / dur * 1e3 / 157.3,
                                                                        val["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0), gui / dur / 1e3))
old = open(os.path.join(DST, "r05_pmc_mfma_product_kernels.md")).read()
head = old[:old.index("| kernel | launches |")]
note = old[old.index("Against the first session's table"):old.index("## raw counters")] if "Against the first session's table" in old else "\n"
open(os.path.join(DST, "r05_pmc_mfma_product_kernels.md"), "w").write(
    head + "| kernel | launches | avg us | GFLOP (counted) | TFLOP/s | of 157.3 | MfmaUtil | clock GHz |\n|---|---|---|---|---|---|---|---|\n" +
    "\n".join(rows) + "\n\n" + note + "## raw counters\n\n```\n" + raw + "```\n")

# RoI geometry stamps: the first section of the file is the kernel as it is
old = open(os.path.join(DST, "r05_roi_geometry_stamps.md")).read()
a = old.index("## the kernel at the end of the round")
a = old.index("\n", a) + 1
b = old.index("## the kernel at the start of the session")
open(os.path.join(DST, "r05_roi_geometry_stamps.md"), "w").write(old[:a] + "\n```\n" + open(os.path.join(SRC, "roi_geometry_stamps.md")).read() + "```\n\n" + old[b:])
print("collected %d files + the MFMA table + the RoI stamps" % len(COPY))
