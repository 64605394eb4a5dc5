#!/bin/bash
# round 4, call B: two-pass kernels with staged row entries and tile walks -- parity, statistics, rates (inline vs call)
set -e
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout -k 10 300 python3 tools/tp_check.py 0.25 > $O/tp_check.txt 2>&1 || { tail -30 $O/tp_check.txt; exit 1; }
tail -3 $O/tp_check.txt
AT_LIB_PATH=aligntools/c_amd/exp/libaligntools_hip_inl.so timeout -k 10 300 python3 tools/tp_check.py 0.25 > $O/tp_check_inl.txt 2>&1 || { tail -30 $O/tp_check_inl.txt; exit 1; }
tail -1 $O/tp_check_inl.txt
AT_LIB_PATH=aligntools/c_amd/exp/libaligntools_hip_st.so timeout -k 10 200 python3 tools/tp_stats.py C2 C3 C4 2>&1 | tee $O/stats_call.txt
AT_LIB_PATH=aligntools/c_amd/exp/libaligntools_hip_stinl.so timeout -k 10 200 python3 tools/tp_stats.py C2 C3 C4 2>&1 | tee $O/stats_inl.txt
for W in C2 C3 C4; do
  for V in main inl; do
    L=aligntools/c_amd/libaligntools_hip.so; [ $V = inl ] && L=aligntools/c_amd/exp/libaligntools_hip_inl.so
    AT_LIB_PATH=$L timeout -k 10 200 python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > $O/${W}_$V.json 2> $O/err_${W}_$V.txt
    python3 -c "import json; d=json.load(open('$O/${W}_$V.json')); print('$W $V', round(d['value'],1), 'GCUPS', d['config']['kernel_config'][:100])"
  done
done
