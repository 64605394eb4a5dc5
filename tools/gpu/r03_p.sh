#!/bin/bash
# round 3, call P: is the slower start of a timed region the chip's clocks (a longer warm-up cures it) or the launches' lockstep (it does not)?
set -e
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
for w in 5 100 5 100 400; do
  for k in 20; do
    python3 bench.py --steps $k --warmup $w --no-cpu-baseline > $O/wu${w}_k$k.json 2> $O/wu${w}_k$k.err
    python3 -c "import json; d=json.load(open('$O/wu${w}_k$k.json')); print('warmup $w steps $k', round(d['ms_per_step'],4), 'ms/step', round(d['value'],1), 'GCUPS', 'alone', d['roofline'].get('kernel_alone_ms'))"
  done
done
