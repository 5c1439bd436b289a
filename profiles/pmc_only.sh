O=gpurun_out/pmc_x; mkdir -p $O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python profiles/pmc_step_probe.py 4 > $O/pmc_$c.log 2>&1
done
python profiles/pmc_step_summarize.py $(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) | head -5 | cut -c1-200
rm -rf $O
