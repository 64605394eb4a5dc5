#!/bin/bash
# copies what tools/gpu/r03_profiles.sh left under gpurun_out/r03p into profiles/r03 (rocprofv3 CSVs reduced to this library's kernels)
set -e
S=gpurun_out/r03p D=profiles/r03
cp $S/traffic_*.json $S/valu_issue.json $S/valu_issue.txt $D/
cp $S/workloads_bench.jsonl $S/c2_driver_style_bench.json $S/c2_streams1_bench.json $S/ab_c2_streams1_no_sliver_items.json $S/ab_c2_no_sliver_items.json \
   $S/c2_under_rocprof_bench.json $S/c2_streams1_under_rocprof_bench.json $S/c4_under_rocprof_bench.json $D/
cp $S/host_path_rate.txt $S/batch_cli_rate.txt $S/ragged_rate.txt $S/pcie_rate.txt $S/length_sweep.jsonl $S/dist_nccl1.json $S/dist_gloo2_C4_self_launched.json $D/
for f in $S/*kernel_stats.csv; do (head -1 $f; grep "at::at_" $f) > $D/$(basename $f); done
for sub in pmc pmc_scores; do for f in $S/$sub/*.csv; do (head -1 $f; grep "at::at_" $f) > $D/$sub/$(basename $f); done; done
python3 - <<'PY'
old = open('profiles/r03/cli_latency.txt').read()
note = old[old.index('# what the platform charges'):] if '# what the platform charges' in old else ''
new = open('gpurun_out/r03p/cli_latency.txt').read()
open('profiles/r03/cli_latency.txt', 'w').write(new.rstrip('\n') + '\n' + note)
PY
