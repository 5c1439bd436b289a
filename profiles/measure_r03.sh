# The command set behind the r03_* artefacts of profiles/ (run on the GPU box from the repo root: bash profiles/measure_r03.sh)
O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-driver > $O/bench_k20.json 2>/dev/null
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-driver --no-lidar > $O/bench_k100.json 2>/dev/null
( echo '## profiles/geo_probe.py both'; python profiles/geo_probe.py both; for s in uniform lidar; do echo; echo "## profiles/stage_probe.py $s"; python profiles/stage_probe.py $s; done; for s in uniform lidar; do echo; echo "## profiles/call_probe.py $s rcnn"; python profiles/call_probe.py $s rcnn; done; echo; echo '## profiles/host_bound_probe.py'; python profiles/host_bound_probe.py; echo; echo '## profiles/fps_probe.py'; python profiles/fps_probe.py; echo; echo '## profiles/graph_probe.py'; timeout 300 python profiles/graph_probe.py; echo; echo '## PRCNN_GRAPHS=0 profiles/host_bound_probe.py'; PRCNN_GRAPHS=0 python profiles/host_bound_probe.py; echo; echo '## tests/ref_kernels_probe.py'; timeout 300 python tests/ref_kernels_probe.py ) 2>&1 | grep -v amdgpu.ids > $O/microbench.txt
for sc in uniform lidar; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$sc -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar > $O/kt_$sc.log 2>&1
  f=$(ls $O/kt_$sc/*/*kernel_trace.csv | head -1)
  python profiles/summarize_step.py $f "round 3, $sc scenes (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar)" > $O/step_$sc.md
  head -70 $(ls $O/kt_$sc/*/*kernel_stats.csv | head -1) > $O/kernel_stats_$sc.csv
  rm -rf $O/kt_$sc
done
# HBM traffic per kernel: one counter per pass (the TCC block cannot hold both), single-stream product step
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python profiles/pmc_step_probe.py 4 > $O/pmc_$c.log 2>&1
done
python profiles/pmc_step_summarize.py $(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) > $O/pmc_product_kernels.md
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cut -c1-400 $O/bench_default.json; head -12 $O/pmc_product_kernels.md
