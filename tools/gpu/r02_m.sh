#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02m
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "8_lane or ragged or fuzz or golden" > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
: > $O/length_sweep.jsonl
for L in 36 40 48 50 56 64 75; do
  timeout -k 10 200 python3 bench.py --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --steps 20 --warmup 5 --no-cpu-baseline >> $O/length_sweep.jsonl 2>> $O/bench.err || true
done
python3 -c "
import json
print([(json.loads(l)['config']['l1'], round(json.loads(l)['value']), json.loads(l)['config']['kernel_config'].split('rows/lane=')[1].split()[0]) for l in open('$O/length_sweep.jsonl')])"
