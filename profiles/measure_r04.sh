# The command set behind the r04_* artefacts of profiles/ (run on the GPU box from the repo root: bash profiles/measure_r04.sh)
O=gpurun_out/r04; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-driver > $O/bench_k20.json 2>/dev/null
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-driver --no-lidar > $O/bench_k100.json 2>/dev/null
( echo '## profiles/geo_probe.py both'; python profiles/geo_probe.py both; for s in uniform lidar; do echo; echo "## profiles/stage_probe.py $s"; python profiles/stage_probe.py $s; done; echo; echo '## profiles/fps_probe.py'; python profiles/fps_probe.py; echo; echo '## PRCNN_FPS_SEQUENTIAL=1 profiles/fps_probe.py'; PRCNN_FPS_SEQUENTIAL=1 python profiles/fps_probe.py; echo; echo '## profiles/host_bound_probe.py'; python profiles/host_bound_probe.py; echo; echo '## profiles/layer_k_sweep.py (persistent layer kernels)'; python profiles/layer_k_sweep.py; echo; echo '## PRCNN_PL_PERSIST=0 profiles/layer_k_sweep.py (one tile per workgroup, round 3)'; PRCNN_PL_PERSIST=0 python profiles/layer_k_sweep.py ) 2>&1 | grep -v amdgpu.ids > $O/microbench.txt
( for s in uniform lidar; do python profiles/dropin_ops_probe.py $s; echo; done; echo '## profiles/dropin_path_probe.py 4'; python profiles/dropin_path_probe.py 4; echo; echo '## tests/ref_kernels_probe.py'; timeout 300 python tests/ref_kernels_probe.py ) 2>&1 | grep -v amdgpu.ids > $O/dropin_ops.md
# the drop-in path proper under a kernel trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt_dropin -- python profiles/dropin_path_probe.py 4 > /dev/null 2>&1
python profiles/dropin_path_probe.py summarize $(ls $O/kt_dropin/*/*kernel_trace.csv | head -1) 4 > $O/dropin_path_kernels.md; rm -rf $O/kt_dropin
# per-step kernel tables of the bench, uniform and LiDAR-shaped scenes
for sc in uniform lidar; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$sc -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar > $O/kt_$sc.log 2>&1
  f=$(ls $O/kt_$sc/*/*kernel_trace.csv | head -1)
  python profiles/summarize_step.py $f "round 4, $sc scenes (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar)" > $O/step_$sc.md
  head -70 $(ls $O/kt_$sc/*/*kernel_stats.csv | head -1) > $O/kernel_stats_$sc.csv
  python profiles/mfma_launch_shapes.py $f > $O/mfma_shapes_$sc.md
  rm -rf $O/kt_$sc
done
# HBM traffic per kernel of the product step: one counter per pass (the TCC block cannot hold both), single stream
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python profiles/pmc_step_probe.py 4 > $O/pmc_$c.log 2>&1
done
python profiles/pmc_step_summarize.py $(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) > $O/pmc_product_kernels.md
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# BASELINE's second metric: the query_and_group sweep -- HIP-event medians, per-kernel trace averages, per-kernel HBM traffic
python profiles/qg_sweep.py uniform 10 2>/dev/null > $O/qg_sweep_uniform.md
python profiles/qg_sweep.py lidar 10 2>/dev/null > $O/qg_sweep_lidar.md
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/qg_kt -- python profiles/qg_sweep.py uniform 5 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/qg_$c -- python profiles/qg_sweep.py uniform 5 > /dev/null 2>&1
done
python profiles/qg_sweep_summarize.py $(ls $O/qg_kt/*/*kernel_trace.csv | head -1) $(ls $O/qg_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $O/qg_WRITE_SIZE/*/*counter_collection.csv | head -1) 5 $O/pmc_query_and_group.json > $O/qg_kernels.md
rm -rf $O/qg_kt $O/qg_FETCH_SIZE $O/qg_WRITE_SIZE
# the drop-in operators' traffic (group_points, three_interpolate, three_nn, ball_query)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/ops_$c -- python profiles/dropin_pmc_probe.py > /dev/null 2>&1
done
python profiles/pmc_step_summarize.py $(ls $O/ops_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $O/ops_WRITE_SIZE/*/*counter_collection.csv | head -1) > $O/pmc_dropin_ops.md
rm -rf $O/ops_FETCH_SIZE $O/ops_WRITE_SIZE
# MFMA counters of the step's MFMA kernels
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python profiles/pmc_step_probe.py 6 > $O/pmc_mfma.log 2>&1
for k in rpn_tail_lin_kernel rcnn_entrance_kernel sa_wide3_kernel sa_packed_mlp256_kernel sa_packed_mlp128_kernel "packed_layer_pipe_kernel<false>" "packed_layer_persist_kernel<false>" packed_layer_stream_kernel; do echo "## $k"; python profiles/pmc_generic.py $(ls $O/pmc_mfma/*/*counter_collection.csv | head -1) "$k"; done > $O/pmc_mfma_product_kernels.txt 2>&1
rm -rf $O/pmc_mfma
# second session: what shares a SIMD with an MFMA stream; instruction mix of the MFMA kernels; RoI pooling and the fused tail alone
( ./profiles/_exp/coexec_probe ) > $O/coexec_probe.md 2>/dev/null      # built here: hipcc --offload-arch=gfx950 -O3 -o profiles/_exp/coexec_probe profiles/coexec_probe.hip
bash profiles/pmc_inst_mix.sh > /dev/null 2>&1; cp gpurun_out/pmc_inst_mix.txt $O/pmc_inst_mix.txt
( echo '## profiles/roipool_probe.py'; python profiles/roipool_probe.py; echo; echo '## profiles/tail_probe.py'; python profiles/tail_probe.py ) 2>&1 | grep -v amdgpu.ids > $O/kernel_probes.txt
cut -c1-400 $O/bench_default.json; head -12 $O/pmc_product_kernels.md
