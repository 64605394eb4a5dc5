#!/bin/bash
# rendering kernel: four pairs per wavefront (product) against one (AT_RENDER_GROUP=64), same box
set -e
mkdir -p gpurun_out/r02c
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "render or strings or cli or golden or host_entry or surface or chunked or edge" > gpurun_out/r02c/pytest6.log 2>&1 || { tail -30 gpurun_out/r02c/pytest6.log; exit 1; }
tail -1 gpurun_out/r02c/pytest6.log
for spec in "C2 36 36" "C2 50 50" "C2 75 75" "C2 100 100" "C2 150 150" "C3 150 150" "C4 150 500" "C3 1024 1024"; do
  set -- $spec
  for rg in 16 64; do
    if [ $2 = 1024 ]; then P=10000; elif [ $3 = 500 ]; then P=100000; else P=$((2250000000 / $2 / $3)); fi
    AT_RENDER_GROUP=$rg timeout -k 10 200 python3 bench.py --workload $1 --l1 $2 --l2 $3 --pairs $P --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 x $3 render group=$rg', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
