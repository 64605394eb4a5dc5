#!/bin/bash
# which kernels does the CIGAR gather add?  kernel trace of the RCCL world-1 run (environment of torch.distributed.run set by hand: no launcher under the profiler)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02dist5
mkdir -p $O
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29731
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/cig -o cig --output-format csv -- python3 $R/bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/cig.json 2> $O/cig.err
export MASTER_PORT=29732
rocprofv3 --kernel-trace --stats -d $O/nocig -o nocig --output-format csv -- python3 $R/bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-cigar-gather > $O/nocig.json 2> $O/nocig.err
cd $R
for v in cig nocig; do
  echo "== $v"; python3 -c "import json; d=json.loads([l for l in open('$O/$v.json') if l.startswith('{')][-1]); print(round(d['value'],1), d['ms_per_step'])"
  f=$(find $O/$v -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -14
done
