#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
for W in C2 C3 C4; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 60 --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err
  python3 tools/bl.py < $O/bench_$W.json || true
done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C2_driverstyle.json 2>/dev/null
python3 tools/bl.py < $O/bench_C2_driverstyle.json || true
