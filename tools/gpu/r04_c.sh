#!/bin/bash
# round 4, call C: two-pass kernels, batched stage loads + branch-light tile walk: parity, statistics, rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout -k 10 300 python3 tools/tp_check.py 0.25 > $O/tp_check.txt 2>&1 || { tail -30 $O/tp_check.txt; exit 1; }
tail -1 $O/tp_check.txt
AT_LIB_PATH=aligntools/c_amd/exp/libaligntools_hip_st.so timeout -k 10 200 python3 tools/tp_stats.py C2 C3 C4 2>&1 | tee $O/stats.txt
for W in C2 C3 C4; do
  for TP in 1 0; do
    AT_TWO_PASS=$TP timeout -k 10 200 python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > $O/${W}_tp$TP.json 2> $O/err_${W}_$TP.txt
    python3 -c "import json; d=json.load(open('$O/${W}_tp$TP.json')); print('$W two_pass=$TP', round(d['value'],1), 'GCUPS', round(d['roofline']['gcups_one_launch_at_a_time'],1), 'alone', d['config']['kernel_config'][:100])"
  done
done
