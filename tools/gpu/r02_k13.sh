#!/bin/bash
# 13 rows per lane on the 8- and 16-lane groups: 3 waves per SIMD with small spills (product) against 2 waves per SIMD without (build_w2), same box
L=$PWD/aligntools/c_amd
for v in "" _w2; do
for spec in "C2 100 100 225000" "C2 200 200 56250" "C3 100 100 225000" "C3 200 200 56250" "C4 100 400 100000" "C4 200 500 50000"; do
  set -- $spec
  AT_LIB_PATH=$L/libaligntools_hip$v.so timeout -k 10 200 python3 bench.py --workload $1 --l1 $2 --l2 $3 --pairs $4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 x $3 lib=$v', round(d['value'],1), d['config']['kernel_config'][13:110])"
done
done
