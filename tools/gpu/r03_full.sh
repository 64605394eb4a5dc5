#!/bin/bash
# profiles/r03/full_size_bench.jsonl: BASELINE configs[3] and [4] at their stated sizes on the one GPU, on the sources as committed
set -e
export TMPDIR=/tmp
O=gpurun_out/r03g
rm -rf $O; mkdir -p $O
: > $O/full_size_bench.jsonl
timeout -k 10 600 python3 bench.py --workload C4full --steps 5 --warmup 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
echo "C4full done"
timeout -k 10 900 python3 bench.py --workload C5full --steps 1 --warmup 0 --streams 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
cut -c1-200 $O/full_size_bench.jsonl
