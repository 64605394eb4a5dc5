#!/bin/bash
# What do the walks and the pointer stores cost?  Throw-away builds (AT_DIAG_NO_WALK, AT_DIAG_NO_STORE) against the product, same box.
set -e
mkdir -p gpurun_out/r02d
L=$PWD/aligntools/c_amd
for w in C4 C2 C3; do
for v in "" _nw _nws; do
  for st in 1 3; do
    AT_LIB_PATH=$L/libaligntools_hip$v.so timeout -k 10 300 python3 bench.py --workload $w --streams $st --steps 30 --warmup 5 --no-cpu-baseline --no-render > gpurun_out/r02d/${w}${v}_s$st.json 2> gpurun_out/r02d/${w}${v}_s$st.err || { tail -5 gpurun_out/r02d/${w}${v}_s$st.err; continue; }
    python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02d/${w}${v}_s$st.json") if l.startswith("{")][-1])
print("$w lib='$v' streams=$st", round(d["value"],1), round(d["ms_per_step"],3), d["roofline"].get("kernel_alone_ms"))
PY
  done
done
done
