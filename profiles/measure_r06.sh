# The command set behind the r06_* artefacts of profiles/ (run on the GPU box from the repo root: bash profiles/measure_r06.sh [part ...])
# parts: fps (a sampling round by phase) | qg (BASELINE's second metric, both scene kinds, with PMC traffic) | double (double.yaml step tables) | step (per-step kernel
# tables + MFMA counters of the bench, both scene kinds) | driver (whole-driver rates) | bench (the driver's own command)
O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
PARTS="${@:-qg double step driver bench fps}"
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

if has double; then
# tools/cfgs/double.yaml (NUM_POINTS 32768): per-step kernel tables of the pipelined bench at that size, uniform and LiDAR-shaped scenes
# (VERDICT r5 "missing 3"), and the sampling kernel alone
for sc in uniform lidar; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32_$sc -- python bench.py --points 32768 --scene $sc --steps 40 --warmup 8 --prewarm 8 --windows 1 --no-cpu-baseline --no-roofline --no-driver --no-lidar > $O/kt32_$sc.log 2>&1
  f=$(ls $O/kt32_$sc/*/*kernel_trace.csv | head -1)
  python profiles/summarize_step.py $f "round 6, tools/cfgs/double.yaml (32768 points per scene), $sc scenes (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --points 32768 --scene $sc --steps 40 --warmup 8 --prewarm 8 --windows 1 --no-cpu-baseline --no-roofline --no-driver --no-lidar)" > $O/double_step_$sc.md
  rm -rf $O/kt32_$sc
  timeout 600 python bench.py --points 32768 --scene $sc --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-roofline --no-driver --no-lidar > $O/double_bench_$sc.json 2> $O/double_bench_$sc.err
done
( echo '## profiles/double_probe.py 24 8'; python profiles/double_probe.py 24 8 ) 2>&1 | grep -v amdgpu.ids > $O/double_probe.txt
fi

if has step; then
# per-step kernel tables of the bench, uniform and LiDAR-shaped scenes (bench.py reads launch_ms_in_step from them), launch shapes,
# and rocprofv3's own statistics
for sc in uniform lidar; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$sc -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --windows 1 --no-cpu-baseline --no-roofline --no-driver --no-lidar > $O/kt_$sc.log 2>&1
  f=$(ls $O/kt_$sc/*/*kernel_trace.csv | head -1)
  python profiles/summarize_step.py $f "round 6, $sc scenes (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --windows 1 --no-cpu-baseline --no-roofline --no-driver --no-lidar)" > $O/step_$sc.md
  head -70 $(ls $O/kt_$sc/*/*kernel_stats.csv | head -1) > $O/kernel_stats_$sc.csv
  python profiles/mfma_launch_shapes.py $f > $O/mfma_shapes_$sc.md
  rm -rf $O/kt_$sc
done
# rocprofv3 statistics of the DRIVER's own command (K = 20)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_k20 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-driver --no-lidar --no-roofline > $O/kt_k20.log 2>&1
head -60 $(ls $O/kt_k20/*/*kernel_stats.csv | head -1) > $O/bench_k20_kernel_stats.csv; rm -rf $O/kt_k20
# HBM traffic per kernel of the product step: one counter per pass (the TCC block cannot hold both), single stream
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python profiles/pmc_step_probe.py 4 > $O/pmc_$c.log 2>&1
done
python profiles/pmc_step_summarize.py $(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) > $O/pmc_product_kernels.md
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# MFMA counters of the step's MFMA kernels, both regimes (one pair of batches = 16 scenes per launch, single stream)
for sc in uniform lidar; do
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$sc -- python profiles/pmc_step_probe.py 6 16 $sc > $O/pmc_mfma_$sc.log 2>&1
  python profiles/pmc_mfma_table.py $(ls $O/pmc_mfma_$sc/*/*counter_collection.csv | head -1) > $O/pmc_mfma_$sc.md
  rm -rf $O/pmc_mfma_$sc
done
# solo times of every kernel of the step (single stream) beside the in-step tables
rm -rf /tmp/kt_occ; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_occ -- python profiles/pmc_step_probe.py 3 > /dev/null 2>&1
python profiles/solo_kernel_times.py /tmp/kt_occ/*/*kernel_trace.csv 40 > $O/solo_kernel_times.md
fi

if has driver; then
timeout 1500 python profiles/driver_probe6.py > $O/driver.md 2> $O/driver.err
fi

if has bench; then
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
timeout 900 python bench.py --steps 100 --warmup 8 --windows 3 --no-cpu-baseline --no-roofline --no-driver > $O/bench_k100.json 2> $O/bench_k100.err
cut -c1-300 $O/bench_k20.json
fi

if has fps; then
# a round of fps_spec_kernel by phase (s_memtime stamps in an instrumented copy: `python profiles/fps_stamps.py build` in the build container first)
for sc in uniform lidar; do timeout 300 python -W ignore profiles/fps_stamps.py run $sc 2>&1 | grep -v amdgpu.ids; done > $O/fps_stamps.txt
fi
