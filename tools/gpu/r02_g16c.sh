#!/bin/bash
# 250- and 300-base reads: four 16-lane groups x 16 / 19 rows (product) against two 32-lane groups (AT_GROUP=32), same box
set -e
mkdir -p gpurun_out/r02c
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "32_lane or lane_groups or sliver" > gpurun_out/r02c/pytest.log 2>&1 || { tail -30 gpurun_out/r02c/pytest.log; exit 1; }
tail -1 gpurun_out/r02c/pytest.log
for mode in C2 C3; do
for L in 220 250 300 304; do
  for grp in 0 32; do
    AT_GROUP=$grp timeout -k 10 200 python3 bench.py --workload $mode --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode $L x $L AT_GROUP=$grp', round(d['value'],1), d['config']['kernel_config'][13:75])"
  done
done
done
for grp in 0 32; do
AT_GROUP=$grp timeout -k 10 200 python3 bench.py --workload C4 --l1 250 --l2 600 --pairs 50000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fit -s 250 x 600 AT_GROUP=$grp', round(d['value'],1), d['config']['kernel_config'][13:75])"
done
