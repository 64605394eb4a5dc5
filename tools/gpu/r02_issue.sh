#!/bin/bash
for rep in 1 2; do
for w in 5 10 20 50 100; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('steps 20 warmup $w', round(d['value'],1), round(d['ms_per_step'],4), 'kmin', round(r['kernel_min_ms'],3))"
done
done
