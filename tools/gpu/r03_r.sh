#!/bin/bash
# round 3, call R: ragged batches plan their frames while the uploads cross the link -- parity of the ragged paths, then the rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03r
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "ragged or frames or fuzz or alphabet or protein or byte" 2>&1 | tail -3
python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1
cut -c1-130 $O/ragged_rate.txt
python3 tools/host_path_rate.py 2>&1 | cut -c1-100
