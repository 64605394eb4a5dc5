#!/bin/bash
# round 3, call D: sliver items fused into the launch + walk loads unified: parity, then A/B against the round-2 formats (same box)
set -e
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
AT_FUZZ_MODES=fitj,overlap,local,global timeout -k 10 600 python3 tests/fuzz_parity.py 40000 302 > $O/fuzz.txt 2>&1 || { tail -30 $O/fuzz.txt; exit 1; }
tail -1 $O/fuzz.txt
run() {  # name, lib (new|old), tail mode, workload, extra args
  if [ $2 = old ]; then export AT_LIB_PATH=$PWD/aligntools/c_amd/exp/libaligntools_hip_old.so; else unset AT_LIB_PATH; fi
  export AT_TAIL_SPLIT=$3
  python3 bench.py --workload $4 --steps 30 --warmup 8 --no-cpu-baseline $5 > $O/$1.json 2> $O/$1.err
  python3 -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],1), 'GCUPS', round(d['ms_per_step'],3), 'ms/step; alone', round(d['roofline']['gcups_one_launch_at_a_time'],1))"
}
for w in C2 C4 C3; do
  run ${w}_old old 0 $w
  run ${w}_new_tail0 new 0 $w
  run ${w}_new_tail1 new 1 $w
  run ${w}_new_tail0_s1 new 0 $w "--streams 1"
  run ${w}_new_tail1_s1 new 1 $w "--streams 1"
done
unset AT_LIB_PATH AT_TAIL_SPLIT
python3 tools/batch_cli_rate.py > $O/batch_cli_rate.txt 2>&1 && cat $O/batch_cli_rate.txt
AT_HOST_TRACE=1 python3 tools/host_path_rate.py > $O/host_path_rate.txt 2> $O/host_trace.txt && cat $O/host_path_rate.txt
