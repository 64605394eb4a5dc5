#!/bin/bash
O=gpurun_out/r02dist
mkdir -p $O
show() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', round(d['value'],1), d['n_gpus'], round(d['ms_per_step'],4))" || tail -5 ${1%.json}.err; }
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workload C4 > $O/n1_C4.json 2> $O/n1_C4.err; show $O/n1_C4.json "plain N=1 C4"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workload C4 > $O/nccl1_C4.json 2> $O/nccl1_C4.err; show $O/nccl1_C4.json "nccl world 1 C4"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29626 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/nccl1_g8.json 2> $O/nccl1_g8.err; show $O/nccl1_g8.json "nccl world 1 C2"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29623 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --workload C4 --pairs 20000 --gather-every 4 > $O/gloo2_C4.json 2> $O/gloo2_C4.err; show $O/gloo2_C4.json "gloo world 2 C4"
timeout -k 10 600 python3 -m pytest tests/test_bench_contract.py tests/test_distributed.py -x -q -m gpu 2>&1 | tail -2
