#!/bin/bash
# suite.sh ROUND [fuzz cases per seed] [tests|fuzz|all] -- the -m gpu suite, smoke(), the BASELINE configs at their stated sizes as bench lines, and a
# randomised parity campaign (tests/fuzz_parity.py: the HIP path against the oracle), on the build as it lies in the tree.
# Results: gpurun_out/<ROUND>q (pytest_gpu_final.log, smoke.log, full_size_bench.jsonl, fuzz_parity.txt).
set -e
export TMPDIR=/tmp
RD=${1:-r04}
N=${2:-60000}
PART=${3:-all}
O=gpurun_out/${RD}q
mkdir -p $O
if [ $PART != fuzz ]; then
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu_final.log 2>&1 || { tail -60 $O/pytest_gpu_final.log; exit 1; }
tail -12 $O/pytest_gpu_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log
: > $O/full_size_bench.jsonl
timeout -k 10 600 python3 bench.py --workload C4full --steps 5 --warmup 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
timeout -k 10 900 python3 bench.py --workload C5full --steps 1 --warmup 0 --streams 1 >> $O/full_size_bench.jsonl 2>> $O/bench.err
timeout -k 10 400 python3 bench.py --workload C5full --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --min-score 30 >> $O/full_size_bench.jsonl 2>> $O/bench.err
cut -c1-300 $O/full_size_bench.jsonl
fi
[ $PART = tests ] && exit 0
: > $O/fuzz_parity.txt
for seed in ${AT_FUZZ_SEEDS:-4301 4302 4303 4304}; do
  timeout -k 10 900 python3 tests/fuzz_parity.py $N $seed >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
  AT_TWO_PASS=2 timeout -k 10 900 python3 tests/fuzz_parity.py $N $((seed + 50)) >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
done
AT_TWO_PASS=2 AT_TP_SPLIT=1 AT_FUZZ_TB=1 timeout -k 10 900 python3 tests/fuzz_parity.py $N 5101 >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }   # pass 2 as a kernel of its own
timeout -k 10 600 python3 tests/fuzz_walk_kernel.py 1500 6102 > $O/fuzz_walk_kernel.txt 2>&1 || { tail -30 $O/fuzz_walk_kernel.txt; exit 1; }   # ... aimed at its shape classes
AT_FUZZ_MODES=fitj,overlap timeout -k 10 900 python3 tests/fuzz_parity.py $N 4401 >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
AT_FUZZ_MODES=overlap,edit AT_FUZZ_TB=0 timeout -k 10 600 python3 tests/fuzz_parity.py $N 4501 >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
AT_FUZZ_MODES=edit AT_FUZZ_EDIT_UNIT=1 AT_MYERS_LANE_MIN_PAIRS=1 timeout -k 10 600 python3 tests/fuzz_parity.py $N 4601 >> $O/fuzz_parity.txt 2>&1 || { tail -30 $O/fuzz_parity.txt; exit 1; }
grep -h "fuzz parity" $O/fuzz_parity.txt
