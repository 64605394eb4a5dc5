#!/bin/bash
# round 3, call G: host entry with ordered uploads; fuzz-seed scan for kernel-class coverage
set -e
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chunked or host_entry or golden or ragged" > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
for o in 1 0; do
  AT_HOST_ORDERED_UPLOADS=$o AT_HOST_TRACE=1 python3 tools/host_path_rate.py > $O/host_path_rate_ordered$o.txt 2> $O/host_trace_ordered$o.txt && cat $O/host_path_rate_ordered$o.txt
done
for c in 6 8 12; do
  echo "AT_HOST_CHUNKS=$c"; AT_HOST_CHUNKS=$c AT_HOST_CHUNK_MIN=4096 python3 tools/host_path_rate.py 2>/dev/null | cut -c1-90
done
python3 tools/ragged_rate.py 150 > $O/ragged_rate.txt 2>&1 && cut -c1-110 $O/ragged_rate.txt
for seed in 21 22 23 24 25 26 27 28; do
  timeout -k 10 300 python3 tests/fuzz_parity.py 1500 $seed > $O/fuzz_seed$seed.txt 2>&1 || { tail -30 $O/fuzz_seed$seed.txt; exit 1; }
done
grep -h "kernel classes" $O/fuzz_seed*.txt | cut -c1-1500
