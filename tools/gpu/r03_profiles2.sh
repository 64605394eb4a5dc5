#!/bin/bash
# profiles/r03/, part 2: the -m gpu suite on the final build, the BASELINE configs at their stated sizes as bench lines, the randomised campaign
set -e
export TMPDIR=/tmp
O=gpurun_out/r03q
rm -rf $O; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu_final.log 2>&1 || { tail -60 $O/pytest_gpu_final.log; exit 1; }
tail -12 $O/pytest_gpu_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log
: > $O/full_size_bench.jsonl
timeout -k 10 600 python3 bench.py --workload C4full --steps 5 --warmup 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
timeout -k 10 900 python3 bench.py --workload C5full --steps 1 --warmup 0 --streams 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
cut -c1-300 $O/full_size_bench.jsonl
: > $O/fuzz_parity.txt
for seed in 3101 3102 3103 3104 3105 3106; do
  timeout -k 10 900 python3 tests/fuzz_parity.py 60000 $seed >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
done
for seed in 3201 3202; do
  AT_FUZZ_MODES=fitj,overlap timeout -k 10 900 python3 tests/fuzz_parity.py 60000 $seed >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
done
grep -h "fuzz parity" $O/fuzz_parity.txt
