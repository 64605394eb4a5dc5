#!/bin/bash
# round 3, call O: bit-parallel edit distance with one alignment per lane (5 / 8 / 16 / 32 words): parity, then the rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03o
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edit or myers or bit_parallel" 2>&1 | tail -3
AT_FUZZ_MODES=edit AT_FUZZ_EDIT_UNIT=1 AT_MYERS_LANE_MIN_PAIRS=1 python3 tests/fuzz_parity.py 40000 31 | cut -c1-600
AT_FUZZ_MODES=edit AT_FUZZ_EDIT_UNIT=1 python3 tests/fuzz_parity.py 20000 32 | cut -c1-600
for wl in E150 E1k; do
  python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/$wl.json 2> $O/$wl.err
  python3 -c "import json; d=json.load(open('$O/$wl.json')); print('$wl', round(d['value'],1), 'GCUPS', d['config']['kernel_config'][:70])"
done
