#!/bin/bash
# round 3, call M: the pack kernel with aligned dword loads: parity (every length / alignment / alphabet the suite and the campaign draw), host rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03m
rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_cli.py tests/test_default_routing.py -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 600 python3 tests/fuzz_parity.py 60000 3401 > $O/fuzz.txt 2>&1 || { tail -30 $O/fuzz.txt; exit 1; }
tail -2 $O/fuzz.txt | cut -c1-200
python3 tools/host_path_rate.py > $O/host_path_rate.txt 2>&1 && cat $O/host_path_rate.txt | cut -c1-100
python3 tools/host_path_pinned.py > $O/host_path_pinned.txt 2>&1 && cat $O/host_path_pinned.txt
python3 tools/batch_cli_rate.py > $O/batch_cli_rate.txt 2>&1 && cat $O/batch_cli_rate.txt
python3 tools/alphabet_rate.py > $O/alphabet_rate.txt 2>&1 && cat $O/alphabet_rate.txt | cut -c1-140
