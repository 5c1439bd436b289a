# quick check of a kernel change: the tests named in $TESTS, solo kernel times of the product step under rocprofv3, bench values
cd /root/repo; export TMPDIR=/tmp
[ -n "$TESTS" ] && timeout 900 python -m pytest tests -m gpu -x -q -k "$TESTS" 2>&1 | tail -3
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python profiles/pmc_step_probe.py 6 > /dev/null 2>&1
echo "## solo kernel times"; python profiles/solo_kernel_times.py /tmp/kt/*/*kernel_trace.csv ${TOP:-16}
one() { timeout 300 python bench.py --scene $SC --steps $K --warmup 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for rep in 1 2; do for SC in uniform lidar; do for K in 100 20; do echo "$SC K=$K $(one)"; done; done; done
