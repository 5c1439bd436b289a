"""gpurun_out/r06/* (written by profiles/measure_r06.sh on the GPU box) -> the tracked profiles/r06_* artefacts.
  python profiles/r06_collect.py"""
import os, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles")
COPY = {"bench_k20.json": "r06_bench_line.json", "bench_k100.json": "r06_bench_line_k100.json",
        "step_uniform.md": "r06_bench_step_kernel_stats_uniform.md", "step_lidar.md": "r06_bench_step_kernel_stats_lidar.md",
        "kernel_stats_uniform.csv": "r06_bench_kernel_stats.csv", "kernel_stats_lidar.csv": "r06_bench_kernel_stats_lidar.csv",
        "bench_k20_kernel_stats.csv": "r06_bench_k20_kernel_stats.csv", "mfma_shapes_uniform.md": "r06_mfma_launch_shapes_uniform.md",
        "mfma_shapes_lidar.md": "r06_mfma_launch_shapes_lidar.md", "pmc_product_kernels.md": "r06_pmc_product_kernels.md",
        "pmc_mfma_uniform.md": "r06_pmc_mfma_uniform.md", "pmc_mfma_lidar.md": "r06_pmc_mfma_lidar.md",
        "solo_kernel_times.md": "r06_solo_kernel_times.md",
        "qg_sweep_uniform.md": "r06_qg_sweep_uniform.md", "qg_sweep_lidar.md": "r06_qg_sweep_lidar.md",
        "qg_kernels_uniform.md": "r06_qg_kernels_uniform.md", "qg_kernels_lidar.md": "r06_qg_kernels_lidar.md",
        "pmc_query_and_group_uniform.json": "r06_pmc_query_and_group_uniform.json", "pmc_query_and_group_lidar.json": "r06_pmc_query_and_group_lidar.json",
        "double_step_uniform.md": "r06_double_yaml_step_uniform.md", "double_step_lidar.md": "r06_double_yaml_step_lidar.md",
        "double_bench_uniform.json": "r06_double_yaml_bench_uniform.json", "double_bench_lidar.json": "r06_double_yaml_bench_lidar.json",
        "double_probe.txt": "r06_double_yaml_probe.txt", "fps_stamps.txt": "r06_fps_round_stamps.txt"}
n = 0
for a, b in COPY.items():
    if os.path.exists(os.path.join(SRC, a)) and os.path.getsize(os.path.join(SRC, a)) > 0:
        shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))
        n += 1
    else:
        print("missing:", a)
print("collected %d files" % n)
