"""Multi-GPU driver: one process per GPU, pairs sharded by contiguous index ranges, no
data-path collective inside the DP.  Collectives (RCCL over xGMI with
backend "nccl"; "gloo" in the CPU tests):
  * broadcast of the scoring block (m,u,o,e,j, use_jump, sites) from rank 0,
  * all_gather of the fixed-size results (score, end_i, end_j, state, nops: 20 B/pair),
  * two-phase gather of the variable-length ops strings (sizes first, then one padded payload).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import opt_t, MODES, MODE_EDIT


def shard_range(n, rank, world):
    """Contiguous range [lo, hi) of pair indices owned by `rank` (pair k -> rank floor(k*world/n))."""
    lo = (n * rank + world - 1) // world
    hi = (n * (rank + 1) + world - 1) // world
    return lo, min(hi, n)


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_scoring(opt, src=0):
    """Rank `src` owns the scoring block; every rank returns an identical opt_t.  Two broadcasts: the fixed block
    (m,u,o,e,j, use_jump, number of sites), then a tensor of exactly that many junction sites -- the reference's site
    list has no upper bound (alignment.h:243-256), so neither has this."""
    dev = _dev()
    head = torch.zeros(8, dtype=torch.int32, device=dev)
    sites = [int(x) for x in opt.sites] if dist.get_rank() == src else []
    if dist.get_rank() == src:
        head[:7] = torch.tensor([opt.m, opt.u, opt.o, opt.e, opt.j, 1 if opt.s else 0, len(sites)], dtype=torch.int32, device=dev)
    dist.broadcast(head, src=src)
    v = head.cpu().tolist()
    ns = v[6]
    if ns > 0:
        sbuf = torch.zeros(ns, dtype=torch.int32, device=dev)
        if dist.get_rank() == src:
            sbuf[:] = torch.tensor(sites, dtype=torch.int32, device=dev)
        dist.broadcast(sbuf, src=src)
        sites = sbuf.cpu().tolist()
    return opt_t(m=v[0], u=v[1], o=v[2], e=v[3], j=v[4], s=bool(v[5]), sites=sites)


def gather_results(local, n_total):
    """local: dict(score,end_i,end_j,state,nops: int32 arrays of this rank's shard; ops: list of bytes).
    Every rank returns the full-batch dict in original pair order."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _dev()
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    width = max(counts) if counts else 0
    fixed = torch.zeros((5, width), dtype=torch.int32, device=dev)
    mine = np.stack([local[k] for k in ("score", "end_i", "end_j", "state", "nops")]).astype(np.int32)
    fixed[:, : mine.shape[1]] = torch.from_numpy(mine).to(dev)
    allfixed = torch.zeros((world * 5, width), dtype=torch.int32, device=dev)   # concatenation along dim 0
    dist.all_gather_into_tensor(allfixed, fixed)
    allfixed = allfixed.cpu().numpy().reshape(world, 5, width)
    out = {k: np.concatenate([allfixed[r, i, : counts[r]] for r in range(world)])
           for i, k in enumerate(("score", "end_i", "end_j", "state", "nops"))}
    if "ops" in local:
        # phase 1: payload sizes; phase 2: one padded byte tensor per rank
        mybytes = b"".join(local["ops"])
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        mysize = torch.tensor([len(mybytes)], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, mysize)
        sizes = sizes.cpu().tolist()
        pad = max(max(sizes), 1)
        payload = torch.zeros(pad, dtype=torch.uint8, device=dev)
        if mybytes:
            payload[: len(mybytes)] = torch.frombuffer(bytearray(mybytes), dtype=torch.uint8).to(dev)
        allpay = torch.zeros(world * pad, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(allpay, payload)
        allpay = allpay.cpu().numpy().reshape(world, pad)
        ops = []
        for r in range(world):
            lo, _ = shard_range(n_total, r, world)
            blob = allpay[r, : sizes[r]].tobytes()
            pos = 0
            for k in range(counts[r]):
                n = int(out["nops"][lo + k])
                ops.append(blob[pos:pos + n])
                pos += n
        out["ops"] = ops
    return out


def align_sharded(mode, pairs, opt=None, compute=None, traceback=True):
    """Align `pairs` (the same full list on every rank) across the process group.
    compute(mode, pairs_shard, opt, traceback) -> dict like Aligner.align_batch; defaults to the HIP Aligner
    bound to this rank's GPU.  Returns the full-batch dict on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    opt = broadcast_scoring(opt if opt is not None else opt_t())
    lo, hi = shard_range(len(pairs), rank, world)
    shard = pairs[lo:hi]
    if compute is None:
        from . import Aligner
        al = Aligner(torch.cuda.current_device())
        al.set_opt(opt)
        local = al.align_batch(mode, shard, traceback=traceback, render=False) if shard else None
        al.close()
    else:
        local = compute(mode, shard, opt, traceback) if shard else None
    if local is None:
        z = np.zeros(0, dtype=np.int32)
        local = dict(score=z, end_i=z, end_j=z, state=z, nops=z, ops=[])
    m = MODES[mode] if isinstance(mode, str) else mode
    if not traceback or m == MODE_EDIT:
        local = {k: v for k, v in local.items() if k != "ops"}
    return gather_results(local, len(pairs)), opt
