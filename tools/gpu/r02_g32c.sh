#!/bin/bash
for L in 350 384 450 512 600; do
for grp in 0 32; do
AT_GROUP=$grp timeout -k 10 200 python3 bench.py --workload C4 --no-jump --l1 $L --l2 $((L+200)) --pairs 30000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fit $L AT_GROUP=$grp', round(d['value'],1), d['config']['kernel_config'][:100])"
done
done
