# The command set behind the r05_* artefacts of profiles/ (run on the GPU box from the repo root: bash profiles/measure_r05.sh)
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
# per-step kernel tables of the bench, uniform and LiDAR-shaped scenes
for sc in uniform lidar; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$sc -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --windows 1 --no-cpu-baseline --no-roofline --no-driver --no-lidar > $O/kt_$sc.log 2>&1
  f=$(ls $O/kt_$sc/*/*kernel_trace.csv | head -1)
  python profiles/summarize_step.py $f "round 5, $sc scenes (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --scene $sc --steps 40 --warmup 8 --prewarm 8 --windows 1 --no-cpu-baseline --no-roofline --no-driver --no-lidar)" > $O/step_$sc.md
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
# MFMA counters of the step's MFMA kernels
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python profiles/pmc_step_probe.py 6 > $O/pmc_mfma.log 2>&1
for k in rpn_tail_lin_kernel rcnn_entrance_kernel sa_wide3_kernel sa_packed_mlp256_kernel sa_packed_mlp128_kernel "packed_layer_pipe_kernel<false>" "packed_layer_persist_kernel<false>" packed_layer_stream_kernel; do echo "## $k"; python profiles/pmc_generic.py $(ls $O/pmc_mfma/*/*counter_collection.csv | head -1) "$k"; done > $O/pmc_mfma_product_kernels.txt 2>&1
rm -rf $O/pmc_mfma
# BASELINE's second metric: the query_and_group sweep
python profiles/qg_sweep.py uniform 10 2>/dev/null > $O/qg_sweep_uniform.md
# double.yaml (32768 points) and the fused tail alone
( echo '## profiles/double_probe.py 24 8'; python profiles/double_probe.py 24 8; echo; echo '## PRCNN_FPS_NO_PAIR=1 profiles/double_probe.py 24 8'; PRCNN_FPS_NO_PAIR=1 python profiles/double_probe.py 24 8; echo; echo '## profiles/tail_probe.py'; python profiles/tail_probe.py; echo '## PRCNN_TAIL_NARROW=0 profiles/tail_probe.py'; PRCNN_TAIL_NARROW=0 python profiles/tail_probe.py; echo '## profiles/nms_probe.py'; python profiles/nms_probe.py ) 2>&1 | grep -v amdgpu.ids > $O/kernel_probes.txt
# second session of the round: what a kernel's microsecond costs the step, the RoI chain and the FPS rounds by phase, occupancy, the pack calls alone
for sc in uniform lidar; do echo "## $sc"; timeout 1100 python -W ignore profiles/sensitivity_probe.py $sc 100 2>&1 | grep -v amdgpu.ids; done > $O/sensitivity.txt
python profiles/roi_geometry_stamps.py build > /dev/null 2>&1; python profiles/fps_stamps.py build > /dev/null 2>&1
for sc in uniform lidar; do timeout 300 python -W ignore profiles/roi_geometry_stamps.py run $sc 2>&1 | grep -v amdgpu.ids; done > $O/roi_geometry_stamps.md
for sc in uniform lidar; do timeout 300 python -W ignore profiles/fps_stamps.py run $sc 2>&1 | grep -v amdgpu.ids; done > $O/fps_stamps.txt
rm -rf /tmp/kt_occ; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_occ -- python profiles/pmc_step_probe.py 3 > /dev/null 2>&1
python profiles/occupancy_table.py /tmp/kt_occ/*/*kernel_trace.csv > $O/occupancy.md; python profiles/solo_kernel_times.py /tmp/kt_occ/*/*kernel_trace.csv 40 > $O/solo_kernel_times.md
for sc in uniform lidar; do timeout 250 python -W ignore profiles/ball_pack_probe.py $sc 2>&1 | grep -v amdgpu.ids; done > $O/ball_pack_probe.md
timeout 900 python bench.py --steps 100 --warmup 8 --windows 3 --no-cpu-baseline --no-roofline --no-driver > $O/bench_k100.json 2> $O/bench_k100.err
cut -c1-300 $O/bench_k20.json; head -12 $O/pmc_product_kernels.md
