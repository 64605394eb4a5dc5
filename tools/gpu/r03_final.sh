#!/bin/bash
# profiles/r03/, last step: the -m gpu suite and one more randomised campaign on the build as committed
set -e
export TMPDIR=/tmp
O=gpurun_out/r03f
rm -rf $O; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu_final.log 2>&1 || { tail -60 $O/pytest_gpu_final.log; exit 1; }
tail -4 $O/pytest_gpu_final.log
: > $O/fuzz_parity_final.txt
for seed in ${AT_FINAL_SEEDS:-3701 3702 3703 3704 3705}; do
  timeout -k 10 900 python3 tests/fuzz_parity.py 100000 $seed >> $O/fuzz_parity_final.txt 2>&1 || { tail -30 $O/fuzz_parity_final.txt; exit 1; }
  echo "seed $seed done"
done
AT_FUZZ_MODES=fitj,overlap timeout -k 10 900 python3 tests/fuzz_parity.py 100000 3706 >> $O/fuzz_parity_final.txt 2>&1
AT_FUZZ_MODES=overlap,edit AT_FUZZ_TB=0 timeout -k 10 900 python3 tests/fuzz_parity.py 100000 3707 >> $O/fuzz_parity_final.txt 2>&1
AT_FUZZ_MODES=edit AT_FUZZ_EDIT_UNIT=1 AT_MYERS_LANE_MIN_PAIRS=1 timeout -k 10 900 python3 tests/fuzz_parity.py 100000 3708 >> $O/fuzz_parity_final.txt 2>&1
grep -h "fuzz parity" $O/fuzz_parity_final.txt
