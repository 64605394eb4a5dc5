#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
run C4 python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_g16 AT_GROUP=16 python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C2 python3 bench.py --workload C2 --steps 60 --no-cpu-baseline
run C3 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C4_b python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
tools/bin/valu_issue $O/valu_issue.json > $O/valu_issue.txt
