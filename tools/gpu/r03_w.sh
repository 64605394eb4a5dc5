#!/bin/bash
# round 3, call W: the ramp overlap scan in four instructions per step: parity (scores-only campaigns, all-vs-all tests), C5all rate
set -e
export TMPDIR=/tmp
O=gpurun_out/r03w
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap or allpairs or all_vs_all or all_pairs or long" 2>&1 | tail -2
AT_FUZZ_MODES=overlap AT_FUZZ_TB=0 python3 tests/fuzz_parity.py 80000 61 | cut -c1-300
for i in 1 2; do
python3 bench.py --workload C5all --steps 6 --warmup 2 --streams 1 --no-render --no-traceback --no-cpu-baseline > $O/C5all.json 2> $O/C5all.err
python3 -c "import json; d=json.load(open('$O/C5all.json')); print('C5all', round(d['value'],1), 'GCUPS', d['config']['kernel_config'][:70])"
done
