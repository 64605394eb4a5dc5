#!/bin/bash
# Everything profiles/r02/ is made of, on one MI355X (through gpurun from the repo root).
set -e
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r02p
rm -rf $O; mkdir -p $O/pmc $O/pmc_scores
tools/bin/valu_issue $O/valu_issue.json > $O/valu_issue.txt
echo "valu_issue done"
# counters of the sweep kernel of every workload (separate --pmc passes, no trace flags beside them)
python3 tools/collect_traffic.py --out $O/pmc C2 C3 C4 C5 > $O/collect.log 2>&1
python3 tools/collect_traffic.py --out $O/pmc_scores --no-traceback C2 C3 C4 > $O/collect_scores.log 2>&1
echo "pmc done"
# bench lines with the CPU baseline, priced with the counters just collected
mkdir -p profiles/r02
cp $O/valu_issue.json profiles/r02/valu_issue.json
cp $O/traffic_*.json profiles/r02/
: > $O/workloads_bench.jsonl
for W in C2 C3 C4 C5 C5all E1k E150; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 60 >> $O/workloads_bench.jsonl 2>> $O/bench.err
  echo "bench $W done"
done
for W in C2 C3 C4; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 60 --no-traceback --no-cpu-baseline >> $O/workloads_bench.jsonl 2>> $O/bench.err
done
# A/B lines: C5 on the int32 kernel, C2 / C4 on the 16-lane groups
AT_NO_PACKED_OVERLAP=1 timeout -k 10 300 python3 bench.py --workload C5 --steps 60 --no-cpu-baseline > $O/ab_C5_int32_bench.json 2>> $O/bench.err
AT_GROUP=16 timeout -k 10 300 python3 bench.py --workload C2 --steps 60 --no-cpu-baseline > $O/ab_C2_group16_bench.json 2>> $O/bench.err
AT_GROUP=16 timeout -k 10 300 python3 bench.py --workload C4 --steps 60 --no-cpu-baseline > $O/ab_C4_group16_bench.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/c2_driver_style_bench.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --streams 1 --steps 60 --no-cpu-baseline > $O/c2_streams1_bench.json 2>> $O/bench.err
# rocprofv3 kernel stats of the default command
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats -o c2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/$O/c2_under_rocprof_bench.json 2> $R/$O/rocprof.err) || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats1 -o c2s1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $R/$O/c2_streams1_under_rocprof_bench.json 2>> $R/$O/rocprof.err) || true
find $O/stats $O/stats1 -name "*kernel_stats.csv" -exec cp {} $O/ \; 2>/dev/null || true
echo "rocprof done"
# read-length sweep (driver-style 20 steps) and ragged host-path rates
: > $O/length_sweep.jsonl
for L in 36 48 50 64 75 100 125 150 152 200 250 300 350 416 512 608; do
  timeout -k 10 200 python3 bench.py --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --steps 20 --warmup 5 --no-cpu-baseline >> $O/length_sweep.jsonl 2>> $O/bench.err || true
done
timeout -k 10 300 python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1 || true
timeout -k 10 300 python3 tools/host_path_rate.py > $O/host_path_rate.txt 2>&1 || true
echo "sweeps done"
# N > 1 pipeline rehearsals on the one card: RCCL world of 1 under torch.distributed.run, gloo world of 2 sharing the GPU
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/dist_nccl1.json 2> $O/dist_nccl1.err || true
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --workload C4 --pairs 20000 > $O/dist_gloo2_C4.json 2> $O/dist_gloo2_C4.err || true
echo "rehearsals done"
