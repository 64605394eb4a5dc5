#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu_a.log 2>&1 || { tail -30 $O/pytest_gpu_a.log; exit 1; }
tail -3 $O/pytest_gpu_a.log
python3 tools/collect_traffic.py --out $O/pmc C2 C3 C4 C5
python3 tools/collect_traffic.py --out $O/pmc_scores --no-traceback C2 C3 C4 C5
