# the narrow RPN SA2 kernel against the padded one: tests, then K = 100 / K = 20 on both scene kinds, alternating
cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "narrow_scales or shadow or packed_kernel" 2>&1 | tail -3
one() { timeout 300 python bench.py --scene $SC --steps $K --warmup 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for rep in 1 2 3; do for NW in 1 0; do export PRCNN_SA_NARROW=$NW; for SC in lidar uniform; do for K in 100 20; do echo "narrow=$NW $SC K=$K $(one)"; done; done; done; done
