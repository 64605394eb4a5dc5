#!/bin/bash
# round 3, call H: host entry -- chunks x ordered uploads
set -e
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
for c in 3 4 5 6 8; do
  for o in 1 0; do
    echo "AT_HOST_CHUNKS=$c ordered=$o"; AT_HOST_ORDERED_UPLOADS=$o AT_HOST_CHUNKS=$c AT_HOST_CHUNK_MIN=8192 python3 tools/host_path_rate.py 2>/dev/null | cut -c1-90
  done
done
AT_HOST_TRACE=1 python3 tools/host_path_rate.py > $O/host_path_rate.txt 2> $O/host_trace.txt && cat $O/host_path_rate.txt
AT_HOST_CHUNKS=8 AT_HOST_CHUNK_MIN=8192 AT_HOST_TRACE=1 python3 tools/host_path_rate.py > $O/host_path_rate8.txt 2> $O/host_trace8.txt && cat $O/host_path_rate8.txt
