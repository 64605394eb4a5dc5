#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
E=$PWD/aligntools/c_amd/exp
for rep in a b; do
run C4_ahead4_$rep python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_ahead8_$rep AT_LIB_PATH=$E/libaligntools_hip_ahead8.so python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_ahead2_$rep AT_LIB_PATH=$E/libaligntools_hip_a2.so python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
run C4_prio_$rep AT_LIB_PATH=$E/libaligntools_hip_prio.so python3 bench.py --workload C4 --steps 60 --no-cpu-baseline
done
run C2_prio AT_LIB_PATH=$E/libaligntools_hip_prio.so python3 bench.py --workload C2 --steps 60 --no-cpu-baseline
run C2 python3 bench.py --workload C2 --steps 60 --no-cpu-baseline
run C3_prio AT_LIB_PATH=$E/libaligntools_hip_prio.so python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C3 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
