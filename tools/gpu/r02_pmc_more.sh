#!/bin/bash
# counters of the score-only workloads that had no roofline line yet: C5 all-vs-all (int32 overlap kernel), edit distance (bit-parallel kernel)
export TMPDIR=/tmp
O=gpurun_out/r02pm
rm -rf $O; mkdir -p $O/pmc_scores
python3 tools/collect_traffic.py --out $O/pmc_scores --no-traceback C5all E1k E150 > $O/collect.log 2>&1
tail -3 $O/collect.log
mkdir -p profiles/r02
cp $O/traffic_*_scores.json profiles/r02/ 2>/dev/null
: > $O/bench.jsonl
for W in C5all E1k E150; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 60 --no-cpu-baseline >> $O/bench.jsonl 2>> $O/bench.err
done
python3 - <<PY
import json
for l in open("$O/bench.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(d["config"]["workload"][:40], round(d["value"]), r.get("frac"), r.get("valu_insts_per_launch"), r.get("cycles_per_inst"), r.get("traffic"))
PY
