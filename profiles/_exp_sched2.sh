# third session of round 5: the early-FP split again, now that RPN SA2 (side streams) is lighter
cd /root/repo; export TMPDIR=/tmp
one() { timeout 300 python bench.py --scene $SC --steps $K --warmup 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for rep in 1 2; do for EF in 2 3 1; do export PRCNN_EARLY_FP=$EF; for SC in uniform lidar; do for K in 100 20; do echo "early_fp=$EF $SC K=$K $(one)"; done; done; done; done
unset PRCNN_EARLY_FP
for rep in 1 2; do for FF in 0 1; do export PRCNN_FINAL_ON_FEATURE=$FF; for SC in uniform lidar; do for K in 20; do echo "final_on_feature=$FF $SC K=$K $(one)"; done; done; done; done
