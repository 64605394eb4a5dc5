#!/bin/bash
# round 3, call V: which waves share a SIMD?  stagger by wave slot bit 0 / workgroup parity / second half of the grid / slot bit 1 / workgroup bit 3
set -e
export TMPDIR=/tmp
O=gpurun_out/r03v
mkdir -p $O
for st in 0 12 268 524 780 1036 280 536; do
  AT_STAGGER=$st python3 bench.py --streams 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/s1_$st.json 2> $O/err.txt
  AT_STAGGER=$st python3 bench.py --workload C4 --streams 1 --steps 30 --warmup 3 --no-cpu-baseline > $O/c4s1_$st.json 2>> $O/err.txt
  python3 -c "
import json
a=json.load(open('$O/s1_$st.json')); b=json.load(open('$O/c4s1_$st.json'))
print('stagger mode', $st >> 8, 'n', $st & 255, ': C2 one at a time', round(a['value'],1), 'alone_ms', round(a['roofline'].get('kernel_alone_ms') or 0,4), '| C4 one at a time', round(b['value'],1))"
done
