#!/bin/bash
# kstats.sh OUT "ENV=val,ENV=val" "WORKLOAD ..." [bench options] -- rocprofv3 kernel stats (one launch at a time: --streams 1) of the
# workloads under the given environment: gpurun_out/OUT/<workload>_kernel_stats.csv and the bench line next to it
set -e
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/$1; ENVS=$(echo "$2" | tr ',' ' '); WLS=$3; shift 3
mkdir -p $O
for e in $ENVS; do export $e; done
for W in $WLS; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/stats_$W -o k --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline --streams 1 "$@" > $R/$O/walk_${W}_bench.json 2> $R/$O/walk_${W}_rocprof.err) || true
  f=$(find $O/stats_$W -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && (head -1 $f; grep 'at::at_' $f) > $O/walk_${W}_kernel_stats.csv && python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-110s calls %6s avg_us %10.1f pct %5s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
  rm -rf $O/stats_$W
done
