#!/bin/bash
# last refresh of the round: GPU suite, smoke, read-length sweep, three fuzz seeds, the driver-style line
export TMPDIR=/tmp
O=gpurun_out/r02last
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log
: > $O/length_sweep.jsonl
for L in 30 36 48 50 64 75 76 100 125 150 152 200 250 300 350 416 512 608; do
  timeout -k 10 200 python3 bench.py --l1 $L --l2 $L --pairs $((2250000000 / L / L)) --steps 20 --warmup 5 --no-cpu-baseline >> $O/length_sweep.jsonl 2>> $O/bench.err || true
done
echo "sweep done"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/c2_driver_style_bench.json 2>> $O/bench.err
tail -c 400 $O/c2_driver_style_bench.json
: > $O/fuzz_parity.txt
for seed in 1101 1102; do
  timeout -k 10 150 python3 tests/fuzz_parity.py 100000 $seed 2>&1 | grep -v amdgpu >> $O/fuzz_parity.txt || echo "seed $seed: time limit or failure" >> $O/fuzz_parity.txt
  tail -1 $O/fuzz_parity.txt
done
