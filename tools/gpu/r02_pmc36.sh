#!/bin/bash
# where does a launch of 36-base reads spend its time?  SQ counters of the sweep kernel (one --pmc pass, no trace flags)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02pmc36
mkdir -p $O
cd /tmp
for L in 36 150; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/l$L -o out --output-format csv -- python3 $R/bench.py --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --streams 1 --steps 3 --warmup 1 --no-render --no-cpu-baseline > $O/l$L.json 2> $O/l$L.err
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$O/l$L/**/*counter_collection.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
agg={}
for r in rows:
    if "at_sweep16" in r["Kernel_Name"]:
        agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("L=$L", {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
done
