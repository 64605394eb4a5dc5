#!/bin/bash
set -e
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "render or strings or cli or golden or long_sequences or baseline_shapes" 2>&1 | tail -1
for rep in 1 2; do
timeout -k 10 300 python3 bench.py --workload C3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', round(d['value'],1))"
AT_RENDER_GROUP=16 timeout -k 10 300 python3 bench.py --workload C3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 render group 16', round(d['value'],1))"
done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 driver-style', round(d['value'],1))"
