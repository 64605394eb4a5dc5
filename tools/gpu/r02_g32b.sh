#!/bin/bash
# reads of 321..608 bases: two 32-lane groups x 12 / 16 / 19 rows (product) against the round-1 classes (AT_GROUP=32: 13 rows up to 416, 64-lane strips beyond), same box
set -e
mkdir -p gpurun_out/r02c
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "32_lane" > gpurun_out/r02c/pytest2.log 2>&1 || { tail -30 gpurun_out/r02c/pytest2.log; exit 1; }
tail -1 gpurun_out/r02c/pytest2.log
for mode in C2 C3; do
for L in 350 384 450 512 560 608; do
  for grp in 0 32; do
    AT_GROUP=$grp timeout -k 10 200 python3 bench.py --workload $mode --l1 $L --l2 $L --pairs $((4500000000 / L / L)) --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode $L x $L AT_GROUP=$grp', round(d['value'],1), d['config']['kernel_config'][:100])"
  done
done
done
for L in 350 500 600; do
for grp in 0 32; do
AT_GROUP=$grp timeout -k 10 200 python3 bench.py --workload C4 --l1 $L --l2 $((L+300)) --pairs 30000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fit -s $L AT_GROUP=$grp', round(d['value'],1), d['config']['kernel_config'][:100])"
done
done
