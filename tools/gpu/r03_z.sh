#!/bin/bash
# round 3, call Z: bit-parallel edit distance, one alignment per lane with 2 / 3 / 4 words for reads of up to 64 / 96 / 128 bases: parity, rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "edit or bit_parallel or aimed" 2>&1 | tail -3
AT_FUZZ_MODES=edit AT_FUZZ_EDIT_UNIT=1 AT_MYERS_LANE_MIN_PAIRS=1 python3 tests/fuzz_parity.py 60000 81 | cut -c1-500
for L in 50 75 100 125; do
  python3 bench.py --workload E150 --l1 $L --l2 $L --pairs $((9000000000 / L / L)) --steps 10 --warmup 3 --no-cpu-baseline > $O/E_$L.json 2> $O/err.txt
  python3 -c "import json; d=json.load(open('$O/E_$L.json')); print('$L x $L', round(d['value'],1), 'GCUPS', d['config']['kernel_config'][:60])"
done
