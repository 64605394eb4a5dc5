#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
NP=$PWD/aligntools/c_amd/exp/libaligntools_hip_nopipe.so
for rep in a b; do
run C4_pipe_g8_$rep python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_nopipe_g8_$rep AT_LIB_PATH=$NP python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_pipe_g16_$rep AT_GROUP=16 python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_nopipe_g16_$rep AT_GROUP=16 AT_LIB_PATH=$NP python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
done
run C4_scores python3 bench.py --workload C4 --steps 60 --no-cpu-baseline --no-traceback
