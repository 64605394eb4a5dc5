#!/bin/bash
# C4 (fit -s 150 x 500) at the per-GPU share BASELINE.json's config names (10 M pairs on 8 GPUs = 1.25 M per GPU) against 100k-pair steps
set -e
mkdir -p gpurun_out/r02s
for p in 100000 400000 1250000; do
  steps=$((3000000 / p + 3))
  timeout -k 10 400 python3 bench.py --workload C4 --pairs $p --steps $steps --warmup 3 --no-cpu-baseline > gpurun_out/r02s/C4_$p.json 2> gpurun_out/r02s/C4_$p.err
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02s/C4_$p.json") if l.startswith("{")][-1])
print("C4 pairs=$p steps=$steps", round(d["value"],1), round(d["ms_per_step"],3), d["roofline"].get("frac"), d["roofline"].get("kernel_alone_ms"))
PY
done
