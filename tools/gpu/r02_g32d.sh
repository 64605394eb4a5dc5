#!/bin/bash
set -e
mkdir -p gpurun_out/r02c
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "32_lane or uniform_batches or baseline_shapes" > gpurun_out/r02c/pytest3.log 2>&1 || { tail -30 gpurun_out/r02c/pytest3.log; exit 1; }
tail -1 gpurun_out/r02c/pytest3.log
for mode in C2 C3; do
for L in 305 320 400 416; do
    timeout -k 10 200 python3 bench.py --workload $mode --l1 $L --l2 $L --pairs $((4500000000 / L / L)) --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode $L x $L', round(d['value'],1), d['config']['kernel_config'][:100])"
done
done
for L in 305 320; do
timeout -k 10 200 python3 bench.py --workload C4 --no-jump --l1 $L --l2 $((L+200)) --pairs 30000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fit $L', round(d['value'],1), d['config']['kernel_config'][:100])"
timeout -k 10 200 python3 bench.py --workload C4 --l1 $L --l2 $((L+200)) --pairs 30000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fit -s $L', round(d['value'],1), d['config']['kernel_config'][:100])"
done
