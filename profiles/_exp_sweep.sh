# usage: bash profiles/_exp_sweep.sh VAR v1 v2 ...  -> bench value at K=100 / K=20, uniform + lidar, for every value of VAR (twice)
cd /root/repo
V=$1; shift
one() { env "$@" python -W ignore bench.py --scene $SC --steps $K --warmup 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for rep in 1 2; do for val in "$@"; do for SC in uniform lidar; do for K in 100 20; do echo "$V=$val $SC K=$K $(one $V=$val)"; done; done; done; done
