run() { # cores loaders writers
  if [ "$1" = "all" ]; then PRE=""; else PRE="taskset -c 0-$(($1-1))"; fi
  for rep in 1 2; do echo "cores $1: $($PRE env PRCNN_LOADER_WORKERS=$2 PRCNN_WRITER_PROCS=$3 python profiles/driver_probe.py run 2>&1 | tail -1)"; done
}
run 16 12 3; run 16 8 3; run 24 16 4; run 32 16 6; run 32 12 3; run 64 16 6; run all 16 6; run all 12 3; run all 8 2
