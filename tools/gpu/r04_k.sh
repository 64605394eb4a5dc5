#!/bin/bash
# round 4, call K: the new two-pass tests, then the whole GPU suite
set -e
export TMPDIR=/tmp
O=gpurun_out/r04k
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_pass" > $O/two_pass.log 2>&1 || { tail -40 $O/two_pass.log; exit 1; }
tail -3 $O/two_pass.log
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
