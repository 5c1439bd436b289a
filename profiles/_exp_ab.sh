# usage: bash profiles/_exp_ab.sh VAR  -> bench value at K=20/K=100, uniform + lidar, VAR=1 vs VAR=0, twice alternating
V=$1
one() { env "$@" python bench.py --scene $SC --steps $K --warmup 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for rep in 1 2; do for val in ${VALS:-1 0}; do for SC in uniform lidar; do for K in 20 100; do if [ "$val" = "unset" ]; then echo "$V unset $SC K=$K $(one X=1)"; else echo "$V=$val $SC K=$K $(one $V=$val)"; fi; done; done; done; done
