#!/bin/bash
for spec in "C2 36 36" "C2 75 75" "C2 150 150" "C3 150 150"; do
  set -- $spec
  for rg in 8 16 32; do
    P=$((2250000000 / $2 / $3))
    AT_RENDER_GROUP=$rg timeout -k 10 200 python3 bench.py --workload $1 --l1 $2 --l2 $3 --pairs $P --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 x $3 render group=$rg', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
for rg in 8 16; do
AT_RENDER_GROUP=$rg timeout -k 10 200 python3 bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 60 steps render group=$rg', round(d['value'],1))"
done
