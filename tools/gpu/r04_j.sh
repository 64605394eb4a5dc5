#!/bin/bash
# round 4, call J: two-pass parity on the 64-lane shapes (packed kernels forced for small batches), both the default routing and AT_TWO_PASS=2
set -e
export TMPDIR=/tmp
O=gpurun_out/r04j
mkdir -p $O
AT_PACKED_MIN_ROUNDS=0 timeout -k 10 500 python3 tools/tp_check.py 0.3 > $O/tp_check.txt 2>&1 || { tail -30 $O/tp_check.txt; exit 1; }
grep -E "1024|tp_check|MISMATCH" $O/tp_check.txt | cut -c1-150
