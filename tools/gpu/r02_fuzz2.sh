#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02fuzz2
mkdir -p $O
: > $O/fuzz_parity.txt
for seed in 401 402 403 404 405 406 407 408 409 410; do
  timeout -k 10 150 python3 tests/fuzz_parity.py 100000 $seed 2>&1 | grep -v amdgpu >> $O/fuzz_parity.txt || echo "seed $seed: time limit or failure" >> $O/fuzz_parity.txt
  tail -1 $O/fuzz_parity.txt
done
