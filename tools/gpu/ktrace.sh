#!/bin/bash
# ktrace.sh OUT "ENV=val,..." WORKLOAD [bench options] -- rocprofv3 kernel trace (start / end of every kernel launch) of one bench run with
# launches in flight: gpurun_out/OUT/<workload>_trace.txt lists the sweep / walk / render launches of the timed steps in start order
set -e
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/$1; ENVS=$(echo "$2" | tr ',' ' '); W=$3; shift 3
mkdir -p $O
for e in $ENVS; do export $e; done
(cd /tmp && rocprofv3 --kernel-trace -d $R/$O/trace_$W -o k --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline "$@" > $R/$O/${W}_trace_bench.json 2> $R/$O/${W}_trace.err) || true
f=$(find $O/trace_$W -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $O/${W}_trace.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("at_sweep16","at_walk16","at_render"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-60:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"]; n="SWEEP" if "at_sweep16" in n else "walk" if "at_walk16" in n else "render"
    print("%-7s q%-3s start %9.1f us  end %9.1f us  dur %8.1f" % (n, r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
rm -rf $O/trace_$W
tail -n 40 $O/${W}_trace.txt
