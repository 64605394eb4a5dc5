#!/bin/bash
# profiles.sh ROUND [main|ab|all] -- everything profiles/<ROUND>/ is made of, on one MI355X (through gpurun from the repo root; ~15 minutes):
# the issue-rate microbenchmark, PMC counters of every workload's sweep kernel (separate --pmc passes), bench lines priced with
# them (CPU baseline included), rocprofv3 kernel stats, the two-pass A/B, host-inclusive rates, N > 1 rehearsals.
# Results land in gpurun_out/<ROUND>p; tools/copy_profiles.sh ROUND copies what is to be committed into profiles/<ROUND>/.
set -e
export TMPDIR=/tmp
R=$PWD
RD=${1:-r04}
PART=${2:-all}     # main: counters, bench lines, rocprofv3 stats, host rates, rehearsals; ab: the two-pass A/B and the walk kernel's stats (gpurun's limit is 20 minutes a call)
O=gpurun_out/${RD}p
[ $PART != ab ] && rm -rf $O
mkdir -p $O/pmc $O/pmc_scores profiles/$RD
if [ $PART != ab ]; then
[ -x tools/bin/valu_issue ] || { mkdir -p tools/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/bin/valu_issue; }
tools/bin/valu_issue $O/valu_issue.json > $O/valu_issue.txt
echo "valu_issue done"
python3 tools/collect_traffic.py --round ${RD#r} --out $O/pmc C2 C3 C4 C5 > $O/collect.log 2>&1
python3 tools/collect_traffic.py --round ${RD#r} --out $O/pmc_scores --no-traceback C2 C3 C4 C5all E1k E150 > $O/collect_scores.log 2>&1
find $O/pmc $O/pmc_scores -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo "pmc done"; du -sh $O | tail -1
cp $O/valu_issue.json profiles/$RD/valu_issue.json
cp $O/traffic_*.json profiles/$RD/
: > $O/workloads_bench.jsonl
for W in C2 C3 C4 C5 C5all E1k E150; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 60 >> $O/workloads_bench.jsonl 2>> $O/bench.err
  echo "bench $W done"
done
for W in C2 C3 C4; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 60 --no-traceback --no-cpu-baseline >> $O/workloads_bench.jsonl 2>> $O/bench.err
done
timeout -k 10 300 python3 bench.py --workload C5all --steps 60 --no-cpu-baseline --min-score 30 >> $O/workloads_bench.jsonl 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/c2_driver_style_bench.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --streams 1 --steps 60 --no-cpu-baseline > $O/c2_streams1_bench.json 2>> $O/bench.err
fi
if [ $PART != main ]; then
# two-pass tracebacks against the one-pass kernels, same box: launches in flight (3, 4) and one launch at a time.  AT_TWO_PASS: 0 never,
# 1 default routing -- the 64-lane groups x 16 rows --, 2 wherever a CK kernel exists; AT_TP_SPLIT=1: pass 2 as a kernel of its own
# (at_walk16.hip.h); AT_DIAG_NO_WALK_KERNEL=1: the sweep with checkpoints alone (no pass 2: what a free pass 2 would reach)
: > $O/two_pass_ab.jsonl
for W in C2 C3 C4; do for V in "0 0 0" "1 0 0" "2 0 0" "2 1 0" "2 1 1"; do set -- $V; for S in 3 4 1; do
  AT_TWO_PASS=$1 AT_TP_SPLIT=$2 AT_DIAG_NO_WALK_KERNEL=$3 timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 --streams $S --no-cpu-baseline | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'workload':'$W','AT_TWO_PASS':$1,'AT_TP_SPLIT':$2,'sweep_only':$3,'streams':$S,'gcups':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],4),'kernel_config':d['config']['kernel_config']}))" >> $O/two_pass_ab.jsonl 2>> $O/bench.err
done; done; done
# where a launch of the walk-kernel form spends its time: rocprofv3 kernel stats one launch at a time, the start / end of every launch with three in flight
bash tools/gpu/kstats.sh ${RD}p AT_TWO_PASS=2,AT_TP_SPLIT=1 "C2 C3 C4" --steps 20 > $O/walk_kernel_stats_rocprof.txt 2>> $O/bench.err || true
for W in C2 C3; do bash tools/gpu/ktrace.sh ${RD}p AT_TWO_PASS=2,AT_TP_SPLIT=1 $W --steps 20 > /dev/null 2>> $O/bench.err || true; done
echo "two-pass A/B done"
fi
[ $PART = ab ] && exit 0
echo "bench done"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats -o c2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/$O/c2_under_rocprof_bench.json 2> $R/$O/rocprof.err) || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats1 -o c2s1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $R/$O/c2_streams1_under_rocprof_bench.json 2>> $R/$O/rocprof.err) || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats3 -o c3 --output-format csv -- python3 $R/bench.py --workload C3 --no-cpu-baseline --steps 60 > $R/$O/c3_under_rocprof_bench.json 2>> $R/$O/rocprof.err) || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats4 -o c4 --output-format csv -- python3 $R/bench.py --workload C4 --no-cpu-baseline --steps 60 > $R/$O/c4_under_rocprof_bench.json 2>> $R/$O/rocprof.err) || true
find $O/stats $O/stats1 $O/stats3 $O/stats4 -name "*kernel_stats.csv" -exec cp {} $O/ \; 2>/dev/null || true
rm -rf $O/stats $O/stats1 $O/stats3 $O/stats4
echo "rocprof done"; du -sh $O | tail -1
timeout -k 10 300 python3 tools/host_path_rate.py > $O/host_path_rate.txt 2>&1 || true
timeout -k 10 300 python3 tools/batch_cli_rate.py > $O/batch_cli_rate.txt 2>&1 || true
timeout -k 10 600 python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1 || true
timeout -k 10 120 python3 tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 || true
timeout -k 10 300 python3 tools/cli_latency.py > $O/cli_latency.txt 2>&1 || true
echo "host rates done"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/dist_nccl1.json 2> $O/dist_nccl1.err || true
timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --workload C4 --pairs 20000 > $O/dist_gloo2_C4_self_launched.json 2> $O/dist_gloo2_C4.err || true
echo "rehearsals done"; du -sh $O | tail -1
