#!/bin/bash
# round 3, call J: after the last kernel edits (padding repeats of ragged frames store nothing; walks bounded by the pair's own slot):
# parity, then the counters again (they carry the fingerprint of the kernel sources)
set -e
export TMPDIR=/tmp
O=gpurun_out/r03j
rm -rf $O; mkdir -p $O/pmc $O/pmc_scores
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 600 python3 tests/fuzz_parity.py 40000 3301 > $O/fuzz.txt 2>&1 || { tail -30 $O/fuzz.txt; exit 1; }
tail -2 $O/fuzz.txt | cut -c1-400
python3 tools/collect_traffic.py --out $O/pmc C2 C3 C4 C5 > $O/collect.log 2>&1
python3 tools/collect_traffic.py --out $O/pmc_scores --no-traceback C2 C3 C4 C5all E1k E150 > $O/collect_scores.log 2>&1
find $O/pmc $O/pmc_scores -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
mkdir -p profiles/r03; cp $O/traffic_*.json profiles/r03/
: > $O/workloads_bench.jsonl
for W in C2 C3 C4 C5; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 60 >> $O/workloads_bench.jsonl 2>> $O/bench.err
done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/c2_driver_style_bench.json 2>> $O/bench.err
python3 - <<'PY'
import json
for ln in open('gpurun_out/r03j/workloads_bench.jsonl'):
    d=json.loads(ln); r=d['roofline']
    print(d['config']['workload'][:12], round(d['value'],1), 'frac', r.get('frac') and round(r['frac'],3), 'ipa', r.get('insts_per_alignment') and round(r['insts_per_alignment']), 'alone', round(r['gcups_one_launch_at_a_time'],1), r.get('stale'))
d=json.load(open('gpurun_out/r03j/c2_driver_style_bench.json')); print('driver style', round(d['value'],1), d['roofline'].get('frac'))
PY
du -sh $O | tail -1
