cd /root/repo; export TMPDIR=/tmp
rm -rf /tmp/kt_l; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/kt_l -- python profiles/pmc_step_probe.py 6 16 lidar > /dev/null 2>&1
for k in "sa_packed_mlp128_kernel<1" sa_packed_mlp256_kernel rcnn_entrance_kernel sa_wide3_kernel rpn_tail_lin_kernel sa_xyz_mlp_packed_mfma_kernel "packed_layer_persist_kernel<false>"; do echo "## $k"; python profiles/pmc_generic.py $(ls /tmp/kt_l/*/*counter_collection.csv | head -1) "$k"; done
