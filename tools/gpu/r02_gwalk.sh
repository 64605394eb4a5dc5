#!/bin/bash
# global walks with the fit walks' look-ahead (product) against one pointer word per op (-DAT_GLOBAL_WALK_AHEAD=0), same box
set -e
mkdir -p gpurun_out/r02g
L=$PWD/aligntools/c_amd
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "global or golden or oracle_ops or uniform_batches or long_sequences or baseline_shapes or extremes or byte_alphabets or harsh" > gpurun_out/r02g/pytest.log 2>&1 || { tail -30 gpurun_out/r02g/pytest.log; exit 1; }
tail -1 gpurun_out/r02g/pytest.log
for rep in 1 2; do
for v in "" _g0; do
  for st in 1 3; do
    AT_LIB_PATH=$L/libaligntools_hip$v.so timeout -k 10 300 python3 bench.py --workload C3 --streams $st --steps 30 --warmup 5 --no-cpu-baseline --no-render > gpurun_out/r02g/C3${v}_s$st.json 2> gpurun_out/r02g/C3${v}_s$st.err || { tail -5 gpurun_out/r02g/C3${v}_s$st.err; continue; }
    python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02g/C3${v}_s$st.json") if l.startswith("{")][-1])
print("C3 lib='$v' streams=$st", round(d["value"],1), round(d["ms_per_step"],3), d["roofline"].get("kernel_alone_ms"))
PY
  done
done
done
for v in "" _g0; do
  for shape in "150 150 100000" "250 250 50000"; do
    set -- $shape
    AT_LIB_PATH=$L/libaligntools_hip$v.so timeout -k 10 300 python3 bench.py --workload C3 --l1 $1 --l2 $2 --pairs $3 --steps 30 --warmup 5 --no-cpu-baseline --no-render 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('global $1 x $2 lib=$v', round(d['value'],1), d['config']['kernel_config'][:60])"
  done
done
