#!/bin/bash
# round 3, call Z: bit-parallel edit distance up to 32 768 bases: parity, rate on 10 kbp pairs
set -e
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edit or bit_parallel" 2>&1 | tail -3
for m in 1 0; do
  AT_MYERS=$m python3 bench.py --workload E1k --l1 10000 --l2 10000 --pairs 4096 --steps 5 --warmup 2 --no-cpu-baseline > $O/E10k_$m.json 2> $O/err.txt
  python3 -c "import json; d=json.load(open('$O/E10k_$m.json')); print('AT_MYERS=$m 4096 x 10000 x 10000', round(d['value'],1), 'GCUPS', d['config']['kernel_config'][:80])"
done
AT_MYERS=1 python3 bench.py --workload E1k --l1 30000 --l2 30000 --pairs 2048 --steps 3 --warmup 1 --no-cpu-baseline > $O/E30k.json 2>> $O/err.txt
python3 -c "import json; d=json.load(open('$O/E30k.json')); print('2048 x 30000 x 30000', round(d['value'],1), 'GCUPS', d['config']['kernel_config'][:80])"
