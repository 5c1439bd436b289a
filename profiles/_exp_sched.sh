run() { # name, env...
  for sc in uniform lidar; do
    for k in 20 100; do
      v=$(env "${@:2}" python bench.py --scene $sc --steps $k --warmup 8 --no-cpu-baseline --no-roofline --no-driver --no-lidar 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
      echo "$1 $sc K=$k $v"
    done
  done
}
run base X=1
run fp2 PRCNN_EARLY_FP=2
run fp1 PRCNN_EARLY_FP=1
run fp0 PRCNN_EARLY_FP=0
run g0off PRCNN_EARLY_G0=0
run base2 X=1
