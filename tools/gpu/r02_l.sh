#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02l
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap or scale or golden or all_vs_all or ops_and_end" > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
run C5 python3 bench.py --workload C5 --steps 60 --no-cpu-baseline
run C5_int32 AT_NO_PACKED_OVERLAP=1 python3 bench.py --workload C5 --steps 60 --no-cpu-baseline
run C5_k4 AT_ROWS_PER_LANE=4 python3 bench.py --workload C5 --steps 60 --no-cpu-baseline
run C5_b python3 bench.py --workload C5 --steps 60 --no-cpu-baseline
