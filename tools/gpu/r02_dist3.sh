#!/bin/bash
O=gpurun_out/r02dist3
mkdir -p $O
port=29700
for args in "--streams 3" "--streams 4" "--streams 5" "--streams 3 --no-cigar-gather" "--streams 3 --no-gather" "--streams 3 --no-render"; do
  for steps in 20 60; do
  port=$((port+1))
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps $steps --warmup 5 --no-cpu-baseline $args > $O/x.json 2> $O/x.err
  python3 -c "import sys,json; d=json.loads([l for l in open('$O/x.json') if l.startswith('{')][-1]); print('nccl world 1 $args steps $steps:', round(d['value'],1), round(d['ms_per_step'],4), 'kavg', round(d['roofline']['kernel_avg_ms'],3))" || tail -3 $O/x.err
  done
done
