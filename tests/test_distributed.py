"""The N>1 path on CPU: world_size-2 gloo process group, pairs sharded by contiguous ranges,
scoring broadcast from rank 0, fixed-size + variable-length results gathered.  The per-shard
compute is injected (the oracle restatement -- allowed in tests) because there is no GPU here;
on the GPU box the same driver runs the HIP Aligner (tests/test_gpu_parity.py covers that path)."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys, json, random
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
import oracle as O
import aligntools.c_amd as A
from aligntools.c_amd.distributed import align_sharded, shard_range

def compute(mode, shard, opt, traceback):
    rs = [O.align(O.MODE_NAMES[mode], a, b, opt.m, opt.u, opt.o, opt.e, opt.j, opt.s, opt.sites) for a, b in shard]
    return dict(score=np.array([r["score"] for r in rs], dtype=np.int32), end_i=np.array([r["end_i"] for r in rs], dtype=np.int32),
                end_j=np.array([r["end_j"] for r in rs], dtype=np.int32), state=np.array([r["state"] for r in rs], dtype=np.int32),
                nops=np.array([len(r["ops"]) for r in rs], dtype=np.int32), ops=[r["ops"] for r in rs])

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
rng = random.Random(5)
pairs = [("".join(rng.choice("ACGT") for _ in range(rng.randint(5, 40))), "".join(rng.choice("ACGT") for _ in range(rng.randint(41, 80)))) for _ in range(%(n)d)]
# only rank 0 knows the real scoring; the others start from defaults and must receive it
SITES = [10, 20, 30] + list(range(100, 400))   # 303 sites: the list has no fixed-size slot in the broadcast
opt = A.opt_t(m=2, u=-2, o=-5, e=-2, j=-7, s=True, sites=SITES) if rank == 0 else A.opt_t()
out = {}
for mode in ("local", "fit", "edit"):
    res, got = align_sharded(mode, pairs, opt, compute=compute)
    assert (got.m, got.u, got.o, got.e, got.j, got.s, got.sites) == (2, -2, -5, -2, -7, True, SITES)
    out[mode] = dict(score=res["score"].tolist(), nops=res["nops"].tolist(), ops=[o.hex() for o in res.get("ops", [])])
print("RESULT" + json.dumps(dict(rank=rank, out=out, ranges=[shard_range(%(n)d, r, 2) for r in range(2)])))
dist.destroy_process_group()
'''


def test_two_rank_gloo_shard_broadcast_gather(tmp_path):
    import json
    import oracle as O
    n = 23
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, n=n))
    port = str(29500 + random.randint(0, 400))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se.decode()[-2000:]
        line = [x for x in so.decode().splitlines() if x.startswith("RESULT")][0]
        outs.append(json.loads(line[6:]))
    assert outs[0]["out"] == outs[1]["out"]                      # every rank holds the full batch
    assert outs[0]["ranges"] == [[0, 12], [12, 23]]             # contiguous, disjoint, covering
    rng = random.Random(5)
    pairs = [("".join(rng.choice("ACGT") for _ in range(rng.randint(5, 40))), "".join(rng.choice("ACGT") for _ in range(rng.randint(41, 80)))) for _ in range(n)]
    for mode in ("local", "fit", "edit"):
        ref = [O.align(O.MODE_NAMES[mode], a, b, 2, -2, -5, -2, -7, True, [10, 20, 30] + list(range(100, 400))) for a, b in pairs]
        assert outs[0]["out"][mode]["score"] == [r["score"] for r in ref]
        if mode != "edit":
            assert outs[0]["out"][mode]["ops"] == [r["ops"].hex() for r in ref]


def test_shard_ranges_partition():
    from aligntools.c_amd.distributed import shard_range
    for n in (0, 1, 7, 100000, 1250000000):
        for world in (1, 2, 4, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[k][1] == rs[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


HIP_WORKER = r'''
import os, sys, json, random
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
import aligntools.c_amd as A
from aligntools.c_amd.distributed import align_sharded
torch.cuda.set_device(0)                      # (two ranks share the one card of the test box; a node gives each its own)
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
rng = random.Random(11)
def dna(n): return "".join(rng.choice("ACGT") for _ in range(n))
pairs = []
for k in range(%(n)d):                        # C4's shape, half of the reads cut out of their contig
    c = dna(500)
    a = c[100:250] if k %% 2 else dna(150)
    pairs.append((a, c))
opt = A.opt_t(m=2, u=-2, o=-5, e=-1, j=-10, s=True, sites=[100, 200, 300, 400]) if rank == 0 else A.opt_t()
out = {}
for mode in ("fit", "local", "global", "overlap", "edit"):
    res, got = align_sharded(mode, pairs, opt)           # compute=None: the HIP Aligner of this rank
    out[mode] = dict(score=res["score"].tolist(), end_i=res["end_i"].tolist(), end_j=res["end_j"].tolist(),
                     ops=[o.hex() for o in res.get("ops", [])])
print("RESULT" + json.dumps(dict(rank=rank, out=out)))
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_two_ranks_hip_kernels_match_oracle(tmp_path):
    """The N > 1 path end to end with the HIP kernels in it: two ranks (gloo; both on the test box's one card), rank 0 owns
    the scoring block, each rank sweeps its contiguous shard on the GPU, results and CIGARs are gathered to every rank --
    and every score, end cell and ops string equals the oracle's."""
    import json
    import oracle as O
    n = 70
    script = tmp_path / "hip_worker.py"
    script.write_text(HIP_WORKER % dict(root=ROOT, n=n))
    port = str(29900 + random.randint(0, 90))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se.decode()[-3000:]
        outs.append(json.loads([x for x in so.decode().splitlines() if x.startswith("RESULT")][0][6:]))
    assert outs[0]["out"] == outs[1]["out"]
    rng = random.Random(11)

    def dna(k):
        return "".join(rng.choice("ACGT") for _ in range(k))
    pairs = []
    for k in range(n):
        c = dna(500)
        a = c[100:250] if k % 2 else dna(150)
        pairs.append((a, c))
    for mode in ("fit", "local", "global", "overlap", "edit"):
        ref = [O.align(O.MODE_NAMES[mode], a, b, 2, -2, -5, -1, -10, True, [100, 200, 300, 400]) for a, b in pairs]
        got = outs[0]["out"][mode]
        assert got["score"] == [r["score"] for r in ref], mode
        if mode != "edit":
            assert got["end_i"] == [r["end_i"] for r in ref] and got["end_j"] == [r["end_j"] for r in ref], mode
            assert got["ops"] == [r["ops"].hex() for r in ref], mode
