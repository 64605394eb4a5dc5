#!/bin/bash
# full GPU suite, then a randomised campaign with fresh seeds (the walks of local and global changed: look-ahead)
export TMPDIR=/tmp
O=gpurun_out/r02fuzz3
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
: > $O/fuzz_parity.txt
for seed in 501 502 503 504 505 506; do
  timeout -k 10 150 python3 tests/fuzz_parity.py 100000 $seed 2>&1 | grep -v amdgpu >> $O/fuzz_parity.txt || echo "seed $seed: time limit or failure" >> $O/fuzz_parity.txt
  tail -1 $O/fuzz_parity.txt
done
