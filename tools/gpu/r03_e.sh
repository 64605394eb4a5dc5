#!/bin/bash
# round 3, call E: walks with incremental addresses and run-length jump runs; host entry through page-locked staging: parity, A/B;
# where the CLI's wall clock goes outside main()
set -e
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_cli.py tests/test_c_consumer.py -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
AT_FUZZ_MODES=fitj,fit,local,global timeout -k 10 600 python3 tests/fuzz_parity.py 40000 303 > $O/fuzz.txt 2>&1 || { tail -30 $O/fuzz.txt; exit 1; }
tail -1 $O/fuzz.txt
run() {  # name, lib (new|old), tail mode, workload, extra args
  if [ $2 = old ]; then export AT_LIB_PATH=$PWD/aligntools/c_amd/exp/libaligntools_hip_old.so; else unset AT_LIB_PATH; fi
  export AT_TAIL_SPLIT=$3
  python3 bench.py --workload $4 --steps 30 --warmup 8 --no-cpu-baseline $5 > $O/$1.json 2> $O/$1.err
  python3 -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],1), 'GCUPS', round(d['ms_per_step'],3), 'ms/step; alone', round(d['roofline']['gcups_one_launch_at_a_time'],1))"
}
for w in C4 C2; do
  run ${w}_old old 0 $w
  run ${w}_new_tail0 new 0 $w
  run ${w}_new_tail1 new 1 $w
  run ${w}_new_tail1_s1 new 1 $w "--streams 1"
done
unset AT_LIB_PATH AT_TAIL_SPLIT
AT_HOST_TRACE=1 python3 tools/host_path_rate.py > $O/host_path_rate.txt 2> $O/host_trace.txt && cat $O/host_path_rate.txt
python3 tools/ragged_rate.py > $O/ragged_rate.txt 2>&1 && cat $O/ragged_rate.txt
python3 - <<'PY' > $O/cli_start.txt 2>&1
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from aligntools.c_amd.synth import synth_pairs_blob
EXE = os.path.join("aligntools", "c_amd", "bin", "alignTools")
def t(cmd, env=None, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env); best = min(best, time.perf_counter() - t0)
    return best, p
print("/bin/true: %.1f ms" % (t(["/bin/true"])[0] * 1e3))
print("alignTools (usage; libraries loaded, no HIP call): %.1f ms" % (t([EXE])[0] * 1e3))
n, l1, l2 = 100000, 150, 150
blob = synth_pairs_blob(0x5EED0002, n, l1, l2)
with tempfile.TemporaryDirectory() as d:
    plain = os.path.join(d, "pairs.fa")
    with open(plain, "wb") as fh:
        for k, row in enumerate(blob):
            fh.write(b">a%d\n" % k + row[:l1].tobytes() + b"\n>b%d\n" % k + row[l1:].tobytes() + b"\n")
    small = os.path.join(d, "one.fa")
    open(small, "wb").write(b">a\nACGTACGTAC\n>b\nACGTTACGTAC\n")
    for name, env in (("fast exit", dict(os.environ, AT_CLI_TRACE="1")), ("normal exit", dict(os.environ, AT_CLI_TRACE="1", AT_FAST_EXIT="0"))):
        dt, p = t([EXE, "batch", "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", plain], env)
        print("batch 100k pairs, %s: %.1f ms\n%s" % (name, dt * 1e3, p.stderr.decode()))
        dt, p = t([EXE, "local", small], env)
        print("single pair, %s: %.1f ms" % (name, dt * 1e3))
PY
cat $O/cli_start.txt
