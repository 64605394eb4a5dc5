#!/bin/bash
# Everything profiles/r03/ is made of, on one MI355X (through gpurun from the repo root; ~15 minutes).
#   part 1 (this script):  counters, bench lines, rocprofv3 kernel stats, host-inclusive rates, N > 1 rehearsals
#   part 2 (r03_profiles2.sh): the -m gpu suite, the full-size workloads, the randomised campaign
set -e
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r03p
rm -rf $O; mkdir -p $O/pmc $O/pmc_scores
[ -x tools/bin/valu_issue ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/bin/valu_issue
tools/bin/valu_issue $O/valu_issue.json > $O/valu_issue.txt
echo "valu_issue done"
# counters of the sweep kernel of every workload (separate --pmc passes, no trace flags beside them)
python3 tools/collect_traffic.py --out $O/pmc C2 C3 C4 C5 > $O/collect.log 2>&1
python3 tools/collect_traffic.py --out $O/pmc_scores --no-traceback C2 C3 C4 C5all E1k E150 > $O/collect_scores.log 2>&1
# (the raw rocprofv3 output directories are large; their rows are in pmc*/<W>_<pass>.csv and the traffic_*.json files)
find $O/pmc $O/pmc_scores -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo "pmc done"; du -sh $O | tail -1
# bench lines with the CPU baseline, priced with the counters just collected
mkdir -p profiles/r03
cp $O/valu_issue.json profiles/r03/valu_issue.json
cp $O/traffic_*.json profiles/r03/
: > $O/workloads_bench.jsonl
for W in C2 C3 C4 C5 C5all E1k E150; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 60 >> $O/workloads_bench.jsonl 2>> $O/bench.err
  echo "bench $W done"
done
for W in C2 C3 C4; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 60 --no-traceback --no-cpu-baseline >> $O/workloads_bench.jsonl 2>> $O/bench.err
done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/c2_driver_style_bench.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --streams 1 --steps 60 --no-cpu-baseline > $O/c2_streams1_bench.json 2>> $O/bench.err
AT_TAIL_SPLIT=0 timeout -k 10 300 python3 bench.py --streams 1 --steps 60 --no-cpu-baseline > $O/ab_c2_streams1_no_sliver_items.json 2>> $O/bench.err
AT_TAIL_SPLIT=0 timeout -k 10 300 python3 bench.py --steps 60 --no-cpu-baseline > $O/ab_c2_no_sliver_items.json 2>> $O/bench.err
echo "bench done"
# rocprofv3 kernel stats of the default command and of one launch at a time
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats -o c2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/$O/c2_under_rocprof_bench.json 2> $R/$O/rocprof.err) || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats1 -o c2s1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $R/$O/c2_streams1_under_rocprof_bench.json 2>> $R/$O/rocprof.err) || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats4 -o c4 --output-format csv -- python3 $R/bench.py --workload C4 --no-cpu-baseline --steps 60 > $R/$O/c4_under_rocprof_bench.json 2>> $R/$O/rocprof.err) || true
find $O/stats $O/stats1 $O/stats4 -name "*kernel_stats.csv" -exec cp {} $O/ \; 2>/dev/null || true
rm -rf $O/stats $O/stats1 $O/stats4
echo "rocprof done"; du -sh $O | tail -1
# host-inclusive rates
timeout -k 10 300 python3 tools/host_path_rate.py > $O/host_path_rate.txt 2>&1 || true
timeout -k 10 300 python3 tools/batch_cli_rate.py > $O/batch_cli_rate.txt 2>&1 || true
timeout -k 10 600 python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1 || true
timeout -k 10 120 python3 tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 || true
timeout -k 10 300 python3 tools/cli_latency.py > $O/cli_latency.txt 2>&1 || true
echo "host rates done"
# read-length sweep (driver-style 20 steps)
: > $O/length_sweep.jsonl
for L in 36 50 75 100 150 200 250 300 416 512 608; do
  timeout -k 10 200 python3 bench.py --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --steps 20 --warmup 5 --no-cpu-baseline >> $O/length_sweep.jsonl 2>> $O/bench.err || true
done
echo "sweeps done"
# N > 1 pipeline rehearsals on the one card: RCCL world of 1 under torch.distributed.run; gloo world of 2 sharing the GPU, started the way
# the driver may start it -- plain `python3 bench.py --gpus 2` (bench.py launches its own ranks)
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/dist_nccl1.json 2> $O/dist_nccl1.err || true
timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --workload C4 --pairs 20000 > $O/dist_gloo2_C4_self_launched.json 2> $O/dist_gloo2_C4.err || true
echo "rehearsals done"; du -sh $O | tail -1
