# per-step kernel tables of the bench, uniform and LiDAR-shaped scenes (run on the GPU box from the repo root: bash profiles/step_tables.sh <tag>)
TAG=${1:-r05}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for sc in uniform lidar; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$sc -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar > $O/kt_$sc.log 2>&1
  f=$(ls $O/kt_$sc/*/*kernel_trace.csv | head -1)
  python profiles/summarize_step.py $f "$TAG, $sc scenes (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar)" > $O/step_$sc.md
  head -70 $(ls $O/kt_$sc/*/*kernel_stats.csv | head -1) > $O/kernel_stats_$sc.csv
  python profiles/mfma_launch_shapes.py $f > $O/mfma_shapes_$sc.md
  rm -rf $O/kt_$sc
done
