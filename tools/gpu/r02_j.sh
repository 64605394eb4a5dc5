#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02j
mkdir -p $O
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
for S in 2 3 4 5 6; do
  run C2_s${S}_20 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams $S
done
for S in 3 4 6; do
  run C2_s${S}_200 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --streams $S
done
run C2_s3_20_b python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 3
# the N > 1 pipeline of bench.py: RCCL at world size 1, gloo with two ranks on this one card
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/dist_nccl1.json 2> $O/dist_nccl1.err
python3 tools/bl.py nccl_world1 < $O/dist_nccl1.json
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --workload C4 --pairs 20000 > $O/dist_gloo2_C4.json 2> $O/dist_gloo2_C4.err
python3 tools/bl.py gloo_world2_C4 < $O/dist_gloo2_C4.json
# a larger randomised campaign
: > $O/fuzz_parity.txt
for seed in 301 302 303 304 305 306 307 308 309 310 311 312 313 314 315 316; do
  timeout -k 10 120 python3 tests/fuzz_parity.py 100000 $seed 2>&1 | grep -v amdgpu >> $O/fuzz_parity.txt || echo "seed $seed: time limit or failure" >> $O/fuzz_parity.txt
done
tail -4 $O/fuzz_parity.txt
