#!/bin/bash
# profiles/r03/, part 2: the -m gpu suite on the final build, the BASELINE configs at their stated sizes as bench lines, the randomised campaign
set -e
export TMPDIR=/tmp
O=gpurun_out/r03q
rm -rf $O; mkdir -p $O
# the issue-rate microbenchmark once more (it prices the bit-parallel kernel with the cost of that kernel's own instruction mix), then the
# bench lines that use those classes
[ -x tools/bin/valu_issue ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/bin/valu_issue
tools/bin/valu_issue $O/valu_issue.json > $O/valu_issue.txt
cp $O/valu_issue.json profiles/r03/valu_issue.json
: > $O/workloads_bench_repriced.jsonl
for W in C5all E1k E150; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 60 >> $O/workloads_bench_repriced.jsonl 2>> $O/bench.err
done
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu_final.log 2>&1 || { tail -60 $O/pytest_gpu_final.log; exit 1; }
tail -12 $O/pytest_gpu_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log
: > $O/full_size_bench.jsonl
timeout -k 10 600 python3 bench.py --workload C4full --steps 5 --warmup 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
timeout -k 10 900 python3 bench.py --workload C5full --steps 1 --warmup 0 --streams 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
cut -c1-300 $O/full_size_bench.jsonl
# the C host's own collectives (at_comm_*: RCCL through dlopen) timed at a world of one -- init, the scoring broadcast, the gathers of
# 100k pairs' scores and strings -- next to the stages of the run (AT_CLI_TRACE)
python3 - <<'PY' > $O/cli_comm_trace.txt 2>&1
import os, subprocess, sys, tempfile
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
    for env in (dict(os.environ, AT_CLI_TRACE="1", AT_COMM_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), dict(os.environ, AT_CLI_TRACE="1")):
        p = subprocess.run([EXE, "batch", "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", plain], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        print("rc", p.returncode, "RCCL world of one" if "AT_COMM_FORCE_RCCL" in env else "no communicator (streaming pipeline)")
        print(p.stderr.decode())
PY
cat $O/cli_comm_trace.txt | cut -c1-120
: > $O/fuzz_parity.txt
for seed in 3301 3302 3303 3304; do
  timeout -k 10 900 python3 tests/fuzz_parity.py 60000 $seed >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
done
for seed in 3401; do
  AT_FUZZ_MODES=fitj,overlap timeout -k 10 900 python3 tests/fuzz_parity.py 60000 $seed >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
done
AT_FUZZ_MODES=overlap,edit AT_FUZZ_TB=0 timeout -k 10 600 python3 tests/fuzz_parity.py 60000 3501 >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
AT_FUZZ_MODES=edit AT_FUZZ_EDIT_UNIT=1 AT_MYERS_LANE_MIN_PAIRS=1 timeout -k 10 600 python3 tests/fuzz_parity.py 60000 3601 >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
grep -h "fuzz parity" $O/fuzz_parity.txt
