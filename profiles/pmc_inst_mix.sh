#!/bin/bash
# Instruction mix of the product step's MFMA kernels (single stream): VALU / MFMA / LDS / vector-memory / scalar instructions per launch
# -> profiles/r04_pmc_inst_mix.md: with the price list of r04_coexec_probe.md, what besides the MFMAs a kernel's time is made of
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/mix; mkdir -p $O
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/a -- python profiles/pmc_step_probe.py 6 > $O/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/b -- python profiles/pmc_step_probe.py 6 > $O/b.log 2>&1
for k in rpn_tail_lin_kernel rcnn_entrance_kernel sa_wide3_kernel sa_packed_mlp256_kernel sa_packed_mlp128_kernel "packed_layer_pipe_kernel<false>" "packed_layer_persist_kernel<false>" packed_layer_stream_kernel packed_layer_pipe32_kernel; do
  echo "## $k"; python profiles/pmc_generic.py $(ls $O/a/*/*counter_collection.csv | head -1) "$k"; python profiles/pmc_generic.py $(ls $O/b/*/*counter_collection.csv | head -1) "$k" | tail -n +2
done > gpurun_out/pmc_inst_mix.txt 2>&1
rm -rf $O
