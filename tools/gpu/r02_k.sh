#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02k2
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "fit or ragged or golden or packed or fuzz or scale" > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
run C4 python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_scores python3 bench.py --workload C4 --steps 60 --no-cpu-baseline --no-traceback
run C4_b python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_g16 AT_GROUP=16 python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
