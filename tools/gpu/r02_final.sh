#!/bin/bash
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r02final
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02final/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r02final/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r02final/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02final/smoke.log 2>&1 && tail -1 gpurun_out/r02final/smoke.log
bash tools/gpu/r02_profiles.sh
