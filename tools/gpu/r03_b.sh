#!/bin/bash
# round 3, call B: pointer-cell diet (fit -s: 4-bit cells + jump bit plane; overlap: 2-bit cells) -- parity, then A/B against the round-2 formats
set -e
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
AT_FUZZ_MODES=fitj,overlap timeout -k 10 600 python3 tests/fuzz_parity.py 60000 301 > $O/fuzz_fitj_overlap.txt 2>&1 || { tail -30 $O/fuzz_fitj_overlap.txt; exit 1; }
tail -1 $O/fuzz_fitj_overlap.txt
for w in C4 C5; do
  for lib in new old; do
    if [ $lib = old ]; then export AT_LIB_PATH=$PWD/aligntools/c_amd/exp/libaligntools_hip_old.so; else unset AT_LIB_PATH; fi
    python3 bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_${w}_${lib}.json 2> $O/bench_${w}_${lib}.err
    python3 -c "import json,sys; d=json.load(open('$O/bench_${w}_${lib}.json')); print('$w $lib', round(d['value'],1), 'GCUPS', round(d['ms_per_step'],3), 'ms/step alone', round(d['roofline']['gcups_one_launch_at_a_time'],1), d['config']['kernel_config'])"
  done
done
unset AT_LIB_PATH
# where the batch CLI's wall clock goes
python3 - <<'PY' > $O/cli_trace.txt 2>&1
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from aligntools.c_amd.synth import synth_pairs_blob
EXE = os.path.join("aligntools", "c_amd", "bin", "alignTools")
n, l1, l2 = 100000, 150, 150
blob = synth_pairs_blob(0x5EED0002, n, l1, l2)
with tempfile.TemporaryDirectory() as d:
    plain = os.path.join(d, "pairs.fa")
    with open(plain, "wb") as fh:
        for k, row in enumerate(blob):
            fh.write(b">a%d\n" % k + row[:l1].tobytes() + b"\n>b%d\n" % k + row[l1:].tobytes() + b"\n")
    for it in range(3):
        t0 = time.perf_counter()
        p = subprocess.run([EXE, "batch", "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", plain], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                           env=dict(os.environ, AT_CLI_TRACE="1"))
        print("run %d: %.3f s" % (it, time.perf_counter() - t0))
        print(p.stderr.decode())
PY
tail -40 $O/cli_trace.txt
