#!/bin/bash
# local walks with look-ahead (-DAT_LOCAL_WALK_AHEAD=1) against one pointer word per op (product), same box; C4 on the SURVEY input
set -e
mkdir -p gpurun_out/r02l
L=$PWD/aligntools/c_amd
AT_LIB_PATH=$L/libaligntools_hip_l1.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "local or golden or oracle_ops or uniform_batches or lane_groups or extremes or byte_alphabets or large_scores" > gpurun_out/r02l/pytest.log 2>&1 || { tail -30 gpurun_out/r02l/pytest.log; exit 1; }
tail -1 gpurun_out/r02l/pytest.log
for rep in 1 2; do
for v in "" _l1; do
  for st in 1 3; do
    AT_LIB_PATH=$L/libaligntools_hip$v.so timeout -k 10 300 python3 bench.py --workload C2 --streams $st --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02l/C2${v}_s$st.json 2> gpurun_out/r02l/C2${v}_s$st.err || { tail -5 gpurun_out/r02l/C2${v}_s$st.err; continue; }
    python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02l/C2${v}_s$st.json") if l.startswith("{")][-1])
print("C2 lib='$v' streams=$st", round(d["value"],1), round(d["ms_per_step"],3), d["roofline"].get("kernel_alone_ms"))
PY
  done
done
done
for st in 1 3; do
timeout -k 10 300 python3 bench.py --workload C4 --streams $st --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 (windows) streams=$st', round(d['value'],1), d['ms_per_step'])"
done
