#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "chunked or host_entry or golden or cli_batch" > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
echo "equal shares"; python3 tools/host_path_rate.py 2>/dev/null | cut -c1-90
for sh in "8,22,25,22,15,8" "25,25,20,15,10,5" "5,10,15,20,25,25" "10,30,30,20,10" "12,30,30,28"; do
  n=$(echo $sh | tr ',' '\n' | wc -l)
  echo "shares $sh"; AT_HOST_CHUNKS=$n AT_HOST_CHUNK_MIN=2048 AT_HOST_CHUNK_SHARES=$sh python3 tools/host_path_rate.py 2>/dev/null | cut -c1-90
done
