run() { for rep in 1 2; do echo "$(env PRCNN_LOADER_WORKERS=$1 PRCNN_WRITER_PROCS=$2 python profiles/driver_probe.py run 2>&1 | tail -1)"; done; }
run 4 2; run 6 2; run 8 2; run 8 3; run 10 3; run 6 1; run 8 1
