#!/bin/bash
# round 3: a sweep grid that leaves a wave slot per CU free (AT_WAVES_PER_CU=7) so that the rendering kernel need not wait for a launch to end
set -e
export TMPDIR=/tmp
O=gpurun_out/r03aa
mkdir -p $O
for w in 8 7 6 7 8; do
  for s in 3 2; do
    for k in 20 60; do
      AT_WAVES_PER_CU=$w python3 bench.py --steps $k --warmup 5 --streams $s --no-cpu-baseline > $O/w${w}_s${s}_k$k.json 2> $O/err.txt
      python3 -c "import json; d=json.load(open('$O/w${w}_s${s}_k$k.json')); print('waves/cu<=$w streams $s steps $k', round(d['value'],1), 'GCUPS')"
    done
  done
done
