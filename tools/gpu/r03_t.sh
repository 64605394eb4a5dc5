#!/bin/bash
# round 3, call T: ragged batches with at most two launches per class and chunk (AT_RAGGED_MIN_BUCKET 16384): parity, rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03t
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "ragged or frames or fuzz" 2>&1 | tail -3
python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1
cut -c1-110 $O/ragged_rate.txt
