#!/bin/bash
set -e
mkdir -p gpurun_out/r02c
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged" > gpurun_out/r02c/pytest4.log 2>&1 || { tail -30 gpurun_out/r02c/pytest4.log; exit 1; }
tail -1 gpurun_out/r02c/pytest4.log
timeout -k 10 300 python3 tools/ragged_rate.py 250 300 2>&1 | grep -v amdgpu
echo "AT_RAGGED_PACKED=0:"
AT_RAGGED_PACKED=0 timeout -k 10 300 python3 tools/ragged_rate.py 250 300 2>&1 | grep -v amdgpu
