#!/bin/bash
# round 4, call A: first run of the two-pass traceback kernels -- parity against the one-pass kernels and the oracle, then rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout -k 10 300 python3 tools/tp_check.py 0.25 > $O/tp_check.txt 2>&1 || { tail -30 $O/tp_check.txt; exit 1; }
tail -40 $O/tp_check.txt
for W in C2 C3 C4; do
  for TP in 1 0; do
    AT_TWO_PASS=$TP timeout -k 10 200 python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > $O/${W}_tp$TP.json 2> $O/err_${W}_$TP.txt
    python3 -c "import json; d=json.load(open('$O/${W}_tp$TP.json')); print('$W two_pass=$TP', round(d['value'],1), 'GCUPS', d['config']['kernel_config'][:110])"
  done
done
