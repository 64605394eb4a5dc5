#!/bin/bash
# round 3, call F: ragged 32-lane frames + ragged packed overlap; sliver as 32-lane items; the BASELINE configs at full size
set -e
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 600 python3 tests/fuzz_parity.py 30000 304 > $O/fuzz.txt 2>&1 || { tail -30 $O/fuzz.txt; exit 1; }
tail -1 $O/fuzz.txt
timeout -k 10 1100 python3 -m pytest tests/test_full_size.py -x -q -m gpu --durations=5 > $O/pytest_full.log 2>&1 || { tail -60 $O/pytest_full.log; exit 1; }
tail -8 $O/pytest_full.log
for w in C2 C4; do
  for t in 0 1; do
    AT_TAIL_SPLIT=$t python3 bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline > $O/${w}_tail$t.json 2> $O/${w}_tail$t.err
    python3 -c "import json; d=json.load(open('$O/${w}_tail$t.json')); print('$w tail$t', round(d['value'],1), 'GCUPS', round(d['ms_per_step'],3), 'ms/step; alone', round(d['roofline']['gcups_one_launch_at_a_time'],1))"
  done
done
python3 tools/ragged_rate.py 500 1000 > $O/ragged_rate_long.txt 2>&1 && cat $O/ragged_rate_long.txt
python3 tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 && cat $O/pcie_rate.txt
