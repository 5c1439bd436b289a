# counted MFMA work of RPN SA2's launch (16 LiDAR-shaped scenes, single stream) with and without the zero padding's MFMAs
cd /root/repo; export TMPDIR=/tmp
for NW in 1 0; do
  rm -rf /tmp/pmc_nw; PRCNN_SA_NARROW=$NW timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_nw -- python profiles/pmc_step_probe.py 6 16 lidar > /dev/null 2>&1
  echo "## PRCNN_SA_NARROW=$NW: sa_packed_mlp128_kernel<2, ...>"; python profiles/pmc_generic.py $(ls /tmp/pmc_nw/*/*counter_collection.csv | head -1) "sa_packed_mlp128_kernel<2"
  echo "## PRCNN_SA_NARROW=$NW: packed_layer_pipe_kernel<true> / stream (the level's per-point parts among them)"; python profiles/pmc_generic.py $(ls /tmp/pmc_nw/*/*counter_collection.csv | head -1) "packed_layer_stream_kernel"
done
