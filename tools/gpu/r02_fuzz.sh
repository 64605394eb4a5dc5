#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02fuzz
mkdir -p $O
: > $O/fuzz_parity.txt
for seed in 201 202 203 204 205 206; do
  timeout -k 10 170 python3 tests/fuzz_parity.py 60000 $seed >> $O/fuzz_parity.txt 2>&1 || echo "seed $seed: stopped by the time limit or failed (rc $?)" >> $O/fuzz_parity.txt
  tail -1 $O/fuzz_parity.txt
done
