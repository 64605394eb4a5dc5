#!/bin/bash
# round 3, call A: the GPU suite on the streamed all-pairs / block-wise reader / pipelined CLI; host-inclusive rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python3 tools/batch_cli_rate.py > $O/batch_cli_rate.txt 2>&1 && cat $O/batch_cli_rate.txt
python3 tools/host_path_rate.py > $O/host_path_rate.txt 2>&1 && cat $O/host_path_rate.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err && cat $O/bench_c2.json
