#!/bin/bash
# round 4, call L: the overlap filter -- parity, then C5 all-vs-all with and without a threshold
set -e
export TMPDIR=/tmp
O=gpurun_out/r04l
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap_threshold or all_vs_all" > $O/filter.log 2>&1 || { tail -40 $O/filter.log; exit 1; }
tail -3 $O/filter.log
for T in none 30 60; do
  F=""; [ $T != none ] && F="--min-score $T"
  timeout -k 10 300 python3 bench.py --workload C5all --steps 10 --warmup 2 --no-cpu-baseline $F > $O/C5all_$T.json 2> $O/err_$T.txt
  python3 -c "import json; d=json.load(open('$O/C5all_$T.json')); print('C5all min-score $T', round(d['value'],1), 'GCUPS', round(d['ms_per_step'],3), 'ms/step', d['config']['kernel_config'][:120])"
done
