#!/bin/bash
# does the 20-step driver-style run lose to the 60-step one through the ramp of the pipeline or through the clocks?
for rep in 1 2; do
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 200" "--steps 60 --warmup 5" "--steps 200 --warmup 5"; do
  timeout -k 10 200 python3 bench.py --gpus 1 $args --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', round(d['value'],1), round(d['ms_per_step'],4), d['roofline'].get('kernel_min_ms'), d['roofline'].get('kernel_avg_ms'))"
done
done
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with cpu baseline, 20/5', round(d['value'],1))"
