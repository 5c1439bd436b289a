# Round 6: one-box sweep of the launch-geometry knobs around their defaults (bench.py, uniform + LiDAR-shaped, K = 20 median of 5 / K = 100)
# usage (GPU box, repo root): bash profiles/knob_sweep6.sh > gpurun_out/r06/knob_sweep.md
run() {
  env "$@" python -W ignore bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline --no-roofline --no-driver 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; l=c['lidar_like']
print('| %s | %.0f | %.0f | %.0f | %.0f |' % (' '.join(sys.argv[1:]) or 'defaults', d['value'], l['scenes_per_s'], 0, l['scenes_per_s_k100']))" "$@"
}
echo "| setting | uniform K = 20 | LiDAR-shaped K = 20 | - | LiDAR-shaped K = 100 |"
echo "|---|---|---|---|---|"
run
run PRCNN_TAIL_GRID=224
run PRCNN_MFMA_GRID=384
run PRCNN_MFMA_GRID=768
run PRCNN_SA_GRID=384
run PRCNN_SA_GRID=768
run PRCNN_PL_STREAM_CAP=384
run PRCNN_PL_STREAM_CAP=768
run PRCNN_PL_PERSIST_MIN=128
run PRCNN_PL_STREAM_MIN=256
run
run PRCNN_FPS_LDS_PAD=0
run PRCNN_GEO_DEPTH=16
run PRCNN_GRAPH_SLOTS=5
run PRCNN_EARLY_FP=3
run PRCNN_EARLY_FP=1
run PRCNN_FINAL_ON_FEATURE=1
run PRCNN_BENCH_LAG=5
run
