#!/bin/bash
# reads of up to 52 bases: sixteen groups of 4 lanes (product) against eight groups of 8 lanes (AT_GROUP=8), same box
set -e
mkdir -p gpurun_out/r02c
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "8_lane or uniform_batches or golden or oracle_ops or sliver or byte_alphabets" > gpurun_out/r02c/pytest5.log 2>&1 || { tail -30 gpurun_out/r02c/pytest5.log; exit 1; }
tail -1 gpurun_out/r02c/pytest5.log
for mode in C2 C3; do
for L in 56 64 70 75 76; do
  for grp in 0 8; do
    AT_GROUP=$grp timeout -k 10 200 python3 bench.py --workload $mode --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode $L x $L AT_GROUP=$grp', round(d['value'],1), d['config']['kernel_config'][13:100])"
  done
done
done
for grp in 0 8; do
AT_GROUP=$grp timeout -k 10 200 python3 bench.py --workload C4 --l1 75 --l2 500 --pairs 200000 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fit -s 75 x 500 AT_GROUP=$grp', round(d['value'],1), d['config']['kernel_config'][13:100])"
done
AT_GROUP=0 timeout -k 10 200 python3 bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['value'],1))"
