#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02i
mkdir -p $O
timeout -k 10 1150 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1 || true
cat $O/ragged_rate.txt
