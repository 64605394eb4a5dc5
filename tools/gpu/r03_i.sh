#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
for o in 2 1 0; do
  echo "ordered=$o"; AT_HOST_ORDERED_UPLOADS=$o python3 tools/host_path_rate.py 2>/dev/null | cut -c1-90
done
for q in 8 16; do
  for o in 2 1; do
    echo "GPU_MAX_HW_QUEUES=$q ordered=$o"; GPU_MAX_HW_QUEUES=$q AT_HOST_ORDERED_UPLOADS=$o python3 tools/host_path_rate.py 2>/dev/null | cut -c1-90
  done
done
echo "GPU_MAX_HW_QUEUES=8 chunks 8"; GPU_MAX_HW_QUEUES=8 AT_HOST_CHUNKS=8 AT_HOST_CHUNK_MIN=8192 python3 tools/host_path_rate.py 2>/dev/null | cut -c1-90
AT_HOST_TRACE=1 python3 tools/host_path_rate.py > $O/host_path_rate.txt 2> $O/host_trace.txt
python3 tools/host_path_pinned.py > $O/host_path_pinned.txt 2>&1 && cat $O/host_path_pinned.txt
