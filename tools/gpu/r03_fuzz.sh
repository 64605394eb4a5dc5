#!/bin/bash
# a longer randomised campaign on the final build (appended to profiles/r03/fuzz_parity.txt)
set -e
export TMPDIR=/tmp
O=gpurun_out/r03fz
rm -rf $O; mkdir -p $O
: > $O/fuzz_more.txt
for seed in 3501 3502 3503 3504 3505 3506 3507 3508 3509 3510; do
  timeout -k 10 400 python3 tests/fuzz_parity.py 60000 $seed >> $O/fuzz_more.txt 2>&1 || { tail -30 $O/fuzz_more.txt; exit 1; }
  grep -h "fuzz parity" $O/fuzz_more.txt | tail -1
done
