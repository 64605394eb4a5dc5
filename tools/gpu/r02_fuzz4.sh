#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02fuzz4
mkdir -p $O
: > $O/fuzz_parity.txt
for seed in 1301 1302 1303; do
  timeout -k 10 200 python3 tests/fuzz_parity.py 100000 $seed 2>&1 | grep -v amdgpu >> $O/fuzz_parity.txt || echo "seed $seed: time limit or failure" >> $O/fuzz_parity.txt
  tail -1 $O/fuzz_parity.txt
done
timeout -k 10 300 python3 -m pytest tests/test_fuzz.py -x -q -m gpu 2>&1 | tail -1
