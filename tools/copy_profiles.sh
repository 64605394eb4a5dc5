#!/bin/bash
# copy_profiles.sh ROUND -- copies what tools/gpu/profiles.sh and tools/gpu/suite.sh left under gpurun_out/<ROUND>p and <ROUND>q into
# profiles/<ROUND>/ (rocprofv3 CSVs reduced to this library's kernels)
set -e
RD=${1:-r04}
S=gpurun_out/${RD}p Q=gpurun_out/${RD}q D=profiles/$RD
mkdir -p $D/pmc $D/pmc_scores
cp $S/traffic_*.json $S/valu_issue.json $S/valu_issue.txt $D/ 2>/dev/null || true
for f in workloads_bench.jsonl two_pass_ab.jsonl c2_driver_style_bench.json c2_streams1_bench.json c2_under_rocprof_bench.json c2_streams1_under_rocprof_bench.json \
         c3_under_rocprof_bench.json c4_under_rocprof_bench.json host_path_rate.txt batch_cli_rate.txt ragged_rate.txt pcie_rate.txt cli_latency.txt \
         dist_nccl1.json dist_gloo2_C4_self_launched.json walk_kernel_stats_rocprof.txt C2_trace.txt C3_trace.txt; do [ -f $S/$f ] && cp $S/$f $D/; done
for f in $S/*kernel_stats.csv; do [ -f "$f" ] && (head -1 $f; grep "at::at_" $f) > $D/$(basename $f); done
for sub in pmc pmc_scores; do for f in $S/$sub/*.csv; do [ -f "$f" ] && (head -1 $f; grep "at::at_" $f) > $D/$sub/$(basename $f); done; done
for f in pytest_gpu_final.log smoke.log full_size_bench.jsonl fuzz_parity.txt; do [ -f $Q/$f ] && cp $Q/$f $D/; done
ls $D | head -60
