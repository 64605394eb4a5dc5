#!/bin/bash
# round 3, call C: what the pointer-cell diet and the concurrent sliver launch are worth, separately (same box)
set -e
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
run() {  # name, lib (new|old), tail mode, workload, extra args
  if [ $2 = old ]; then export AT_LIB_PATH=$PWD/aligntools/c_amd/exp/libaligntools_hip_old.so; else unset AT_LIB_PATH; fi
  export AT_TAIL_SPLIT=$3
  python3 bench.py --workload $4 --steps 30 --warmup 8 --no-cpu-baseline $5 > $O/$1.json 2> $O/$1.err
  python3 -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],1), 'GCUPS', round(d['ms_per_step'],3), 'ms/step; alone', round(d['roofline']['gcups_one_launch_at_a_time'],1))"
}
for w in C4 C5 C2 C3; do
  run ${w}_old_tail0 old 0 $w
  run ${w}_new_tail0 new 0 $w
  run ${w}_new_tail2 new 2 $w
  run ${w}_new_tail0_s1 new 0 $w "--streams 1"
  run ${w}_new_tail2_s1 new 2 $w "--streams 1"
done
unset AT_LIB_PATH AT_TAIL_SPLIT
# instruction counts and HBM bytes of the C4 sweep, old and new formats
for lib in old new; do
  if [ $lib = old ]; then export AT_LIB_PATH=$PWD/aligntools/c_amd/exp/libaligntools_hip_old.so; else unset AT_LIB_PATH; fi
  export AT_TAIL_SPLIT=0
  for c in SQ_INSTS_VALU WRITE_SIZE FETCH_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c -d $OLDPWD/$O/pmc_${lib}_$c -o out --output-format csv -- python3 $OLDPWD/bench.py --workload C4 --streams 1 --steps 3 --warmup 1 --no-render --no-cpu-baseline > /dev/null 2>&1)
    python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/pmc_${lib}_$c/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
v=[float(r["Counter_Value"]) for r in rows if "at_sweep16" in r["Kernel_Name"] and r["Counter_Name"]=="$c"]
print("C4 $lib $c per launch:", sum(v)/max(1,len(v)), "dispatches", len(v))
PY
  done
done
