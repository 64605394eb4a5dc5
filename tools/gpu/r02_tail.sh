#!/bin/bash
# Tail split A/B (one box): sliver of a batch on the 64-lane kernel vs all on the narrow groups; lone launches and 3 streams.
set -e
mkdir -p gpurun_out/r02t
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sliver or packed_8_lane or packed16_uniform or full_size_c2 or baseline_shapes" > gpurun_out/r02t/pytest.log 2>&1 || { tail -30 gpurun_out/r02t/pytest.log; exit 1; }
tail -2 gpurun_out/r02t/pytest.log
for w in C2 C4; do
  for split in 0 1; do
    for st in 1 3; do
      AT_TAIL_SPLIT=$split timeout -k 10 300 python3 bench.py --workload $w --streams $st --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02t/${w}_split${split}_s${st}.json 2> gpurun_out/r02t/${w}_split${split}_s${st}.err
      python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02t/${w}_split${split}_s${st}.json") if l.startswith("{")][-1])
print("$w split=$split streams=$st", round(d["value"],1), d["ms_per_step"], d["config"]["kernel_config"][-60:])
PY
    done
  done
done
AT_TAIL_SPLIT=1 timeout -k 10 300 python3 bench.py --workload C2 --no-traceback --streams 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 scores split=1 s1', d['value'])" || true
AT_TAIL_SPLIT=0 timeout -k 10 300 python3 bench.py --workload C2 --no-traceback --streams 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 scores split=0 s1', d['value'])" || true
