#!/bin/bash
# round 3: launches in flight (bench.py --streams) against the length of the timed region
set -e
export TMPDIR=/tmp
O=gpurun_out/r03aa
mkdir -p $O
for rep in 1 2; do
for s in 3 4 5 6 2; do
  for k in 20 60; do
    python3 bench.py --steps $k --warmup 5 --streams $s --no-cpu-baseline > $O/s${s}_k$k.json 2> $O/err.txt
    python3 -c "import json; d=json.load(open('$O/s${s}_k$k.json')); print('streams $s steps $k', round(d['value'],1), 'GCUPS')"
  done
done
done
