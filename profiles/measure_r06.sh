# The command set behind the r06_* artefacts of profiles/ (run on the GPU box from the repo root: bash profiles/measure_r06.sh [part ...])
# parts: qg (BASELINE's second metric, both scene kinds, with PMC traffic) | double (double.yaml step tables) | step (per-step kernel
# tables + MFMA counters of the bench, both scene kinds) | driver (whole-driver rates) | bench (the driver's own command)
O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
PARTS="${@:-qg double step driver bench}"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }

if has qg; then
# BASELINE's second metric, ball_query + group (prcnn_query_and_group), on uniform AND LiDAR-shaped scenes: HIP-event medians,
# per-kernel trace averages, per-kernel HBM traffic (separate --pmc passes, FETCH_SIZE doubled for gfx950)
for sc in uniform lidar; do
  python profiles/qg_sweep.py $sc 10 2>/dev/null > $O/qg_sweep_$sc.md
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/qg_kt_$sc -- python profiles/qg_sweep.py $sc 5 > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/qg_${c}_$sc -- python profiles/qg_sweep.py $sc 5 > /dev/null 2>&1
  done
  python profiles/qg_sweep_summarize.py $(ls $O/qg_kt_$sc/*/*kernel_trace.csv | head -1) $(ls $O/qg_FETCH_SIZE_$sc/*/*counter_collection.csv | head -1) $(ls $O/qg_WRITE_SIZE_$sc/*/*counter_collection.csv | head -1) 5 $O/pmc_query_and_group_$sc.json $sc > $O/qg_kernels_$sc.md
  rm -rf $O/qg_kt_$sc $O/qg_FETCH_SIZE_$sc $O/qg_WRITE_SIZE_$sc
done
fi
