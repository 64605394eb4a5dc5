#!/bin/bash
# round 3, call Y: the 64-lane group with 8 rows per lane for long reads with tracebacks (C3): parity, rates A/B
set -e
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
for k8 in 1 0; do
  AT_G64_K8=$k8 python3 bench.py --workload C3 --steps 40 --warmup 5 --no-cpu-baseline > $O/C3_$k8.json 2> $O/err.txt
  AT_G64_K8=$k8 python3 bench.py --workload C3 --steps 20 --warmup 3 --streams 1 --no-cpu-baseline > $O/C3s1_$k8.json 2>> $O/err.txt
  python3 -c "
import json
a=json.load(open('$O/C3_$k8.json')); b=json.load(open('$O/C3s1_$k8.json'))
print('AT_G64_K8=$k8 C3', round(a['value'],1), 'GCUPS | one launch at a time', round(b['value'],1), '|', a['config']['kernel_config'][:110])"
done
for L in 1500; do
  for k8 in 1 0; do
    AT_G64_K8=$k8 python3 bench.py --workload C2 --l1 $L --l2 $L --pairs $((20000000000 / L / L)) --steps 20 --warmup 3 --no-cpu-baseline > $O/L${L}_$k8.json 2>> $O/err.txt
    python3 -c "
import json
a=json.load(open('$O/L${L}_$k8.json'))
print('AT_G64_K8=$k8 local $L x $L', round(a['value'],1), 'GCUPS |', a['config']['kernel_config'][:100])"
  done
done
for k8 in 1 0; do
  AT_G64_K8=$k8 python3 bench.py --workload C3 --l1 1024 --l2 1024 --pairs 40000 --steps 20 --warmup 3 --no-cpu-baseline > $O/C3big_$k8.json 2>> $O/err.txt
  python3 -c "
import json
a=json.load(open('$O/C3big_$k8.json'))
print('AT_G64_K8=$k8 C3 with 40000 pairs', round(a['value'],1), 'GCUPS |', a['config']['kernel_config'][:110])"
done
