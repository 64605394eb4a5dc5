#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
for W in C2 C3 C4; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 60 --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err
  python3 tools/bl.py $W < $O/bench_$W.json || true
done
for W in C2 C4; do
  AT_GROUP=16 timeout -k 10 300 python3 bench.py --workload $W --steps 60 --no-cpu-baseline > $O/bench_${W}_g16.json 2> $O/bench_${W}_g16.err
  python3 tools/bl.py $W g16 < $O/bench_${W}_g16.json || true
  timeout -k 10 300 python3 bench.py --workload $W --steps 60 --no-cpu-baseline --no-traceback > $O/bench_${W}_scores.json 2> $O/bench_${W}_scores.err
  python3 tools/bl.py $W scores < $O/bench_${W}_scores.json || true
done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C2_driverstyle.json 2>/dev/null
python3 tools/bl.py driverstyle < $O/bench_C2_driverstyle.json || true
for L in 36 48 64 100; do
  timeout -k 10 300 python3 bench.py --l1 $L --l2 $L --pairs 400000 --steps 40 --no-cpu-baseline > $O/bench_len$L.json 2>/dev/null
  python3 tools/bl.py len$L < $O/bench_len$L.json || true
done
tools/bin/valu_issue $O/valu_issue.json > $O/valu_issue.txt
