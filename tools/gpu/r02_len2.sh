#!/bin/bash
for L in 200 208 304 320 350 416 420 450 500 512 600 800 1000; do
  timeout -k 10 200 python3 bench.py --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('local $L x $L', round(d['value'],1), d['config']['kernel_config'][:110])"
done
