#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02h
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1 || true
cat $O/ragged_rate.txt
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
run C2 python3 bench.py --workload C2 --steps 60 --no-cpu-baseline
run C3 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C4 python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
