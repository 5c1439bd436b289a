mkdir -p gpurun_out/r2h; O=gpurun_out/r2h; export TMPDIR=/tmp
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2>/dev/null
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_k100.json 2>/dev/null
for p in "packed_probe.py 128" "packed_probe.py 256" "layer_probe.py" "tail_probe.py" "roipool_probe.py" "fps_probe.py" "stage_probe.py" "gap_probe.py" "host_bound_probe.py" "cold_start_probe.py"; do echo "## $p" >> $O/micro.txt; timeout 300 python profiles/$p 2>&1 | grep -v "amdgpu.ids" | tail -16 >> $O/micro.txt; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 16 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver > $O/kt.log 2>&1
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python profiles/summarize_step.py $f "round 2 final (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 16 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver)" > $O/step.md
head -62 $(ls $O/kt/*/*kernel_stats.csv | head -1) > $O/kernel_stats_top.csv
rm -rf $O/kt
cat $O/bench_default.json | cut -c1-600
