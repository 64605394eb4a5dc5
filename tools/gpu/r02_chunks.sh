#!/bin/bash
for c in 1 2 3 4 6 8; do
  echo "AT_HOST_CHUNKS=$c"
  AT_HOST_CHUNKS=$c AT_HOST_CHUNK_MIN=8192 timeout -k 10 200 python3 tools/host_path_rate.py 2>&1 | grep "host path" | cut -c1-100
done
