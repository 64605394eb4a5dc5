#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r04e
mkdir -p $O
AT_LIB_PATH=aligntools/c_amd/exp/libaligntools_hip_st.so timeout -k 10 200 python3 tools/tp_stats.py C2 C3 C4 2>&1 | tee $O/stats.txt
