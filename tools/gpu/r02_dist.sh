#!/bin/bash
# the N > 1 pipeline rehearsed on the one card: plain N = 1, RCCL world of 1 (per-step and grouped gathers), gloo world of 2 and 3
O=gpurun_out/r02dist
mkdir -p $O
show() { python3 -c "import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', round(d['value'],1), d['n_gpus'], round(d['ms_per_step'],4), json.dumps(d['config'].get('gather'))[:200])" || tail -5 ${1%.json}.err; }
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/n1.json 2> $O/n1.err; show $O/n1.json "plain N=1"
for ge in 8 1 20; do
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$((ge % 10)) bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --gather-every $ge > $O/nccl1_g$ge.json 2> $O/nccl1_g$ge.err; show $O/nccl1_g$ge.json "nccl world 1, gather every $ge"
done
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workload C4 > $O/nccl1_C4.json 2> $O/nccl1_C4.err; show $O/nccl1_C4.json "nccl world 1 C4"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29622 bench.py --gpus 1 --steps 21 --warmup 3 --no-cpu-baseline --workload C3 --gather-every 4 > $O/nccl1_C3.json 2> $O/nccl1_C3.err; show $O/nccl1_C3.json "nccl world 1 C3 (21 steps, groups of 4)"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29623 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --workload C4 --pairs 20000 --gather-every 4 > $O/gloo2_C4.json 2> $O/gloo2_C4.err; show $O/gloo2_C4.json "gloo world 2 C4"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29624 bench.py --gpus 3 --steps 9 --warmup 2 --no-cpu-baseline --backend gloo --pairs 30000 > $O/gloo3_C2.json 2> $O/gloo3_C2.err; show $O/gloo3_C2.json "gloo world 3 C2"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29625 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --backend gloo --workload C5all --pairs 20000 > $O/gloo2_C5all.json 2> $O/gloo2_C5all.err; show $O/gloo2_C5all.json "gloo world 2 C5all"
