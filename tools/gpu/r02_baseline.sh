#!/bin/bash
# round-2 baseline measurements (run through gpurun from the repo root)
set -e
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
tools/bin/valu_issue $O/valu_issue.json > $O/valu_issue.txt
echo "valu_issue done"
for W in C2 C3 C4 C5; do
  python3 bench.py --workload $W --steps 60 > $O/bench_$W.json 2> $O/bench_$W.err
  echo "bench $W done"
done
python3 bench.py --workload C5all --steps 20 > $O/bench_C5all.json 2> $O/bench_C5all.err
for W in C2 C3 C4 C5; do
  python3 bench.py --workload $W --steps 60 --no-traceback --no-cpu-baseline > $O/bench_${W}_scores.json 2> $O/bench_${W}_scores.err
done
echo "scores done"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C2_driverstyle.json 2>/dev/null
python3 tools/collect_traffic.py --out $O/pmc C2 C3 C4 C5
