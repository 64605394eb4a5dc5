"""Deterministic synthetic DNA for benchmarks and parity runs:
bases iid uniform over ACGT from xorshift64* streams, 2 bits per base.  One
stream per pair, seeded by splitmix64(seed + pair index), so any shard of the
batch can be generated independently on any rank."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def synth_codes(seed, first_pair, npairs, nbases):
    """uint8 array [npairs, nbases] of base codes 0..3."""
    with np.errstate(over="ignore"):
        idx = np.arange(first_pair, first_pair + npairs, dtype=np.uint64)
        s = _splitmix64(idx + np.uint64(seed))
        s[s == 0] = np.uint64(0x9E3779B97F4A7C15)
        nwords = (nbases + 31) // 32
        out = np.empty((npairs, nwords * 32), dtype=np.uint8)
        shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
        for w in range(nwords):
            s ^= s >> np.uint64(12)
            s ^= (s << np.uint64(25)) & _M
            s ^= s >> np.uint64(27)
            r = (s * np.uint64(0x2545F4914F6CDD1D)) & _M
            out[:, w * 32:(w + 1) * 32] = ((r[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)
    return out[:, :nbases]


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def synth_pairs_blob(seed, npairs, l1, l2, first_pair=0):
    """uint8 array [npairs, l1+l2]: s1 then s2 of every pair, ASCII."""
    return _ACGT[synth_codes(seed, first_pair, npairs, l1 + l2)]


def mutate_pairs(blob, l1, l2, seed, sub=0.05):
    """Parity-run variant: s2 becomes a substituted copy of s1 (related pairs, long tracebacks)."""
    rng = np.random.default_rng(seed)
    out = blob.copy()
    n = min(l1, l2)
    out[:, l1:l1 + n] = blob[:, :n]
    mask = rng.random((blob.shape[0], n)) < sub
    out[:, l1:l1 + n][mask] = _ACGT[rng.integers(0, 4, size=int(mask.sum()))]
    return out


def _uniform01(seed, first_pair, npairs, ncols, stream):
    """[npairs, ncols] floats in [0, 1), a function of (seed, stream, pair index, column) only -- any shard, any rank."""
    with np.errstate(over="ignore"):
        idx = np.arange(first_pair, first_pair + npairs, dtype=np.uint64)
        base = _splitmix64(idx * np.uint64(0x100000001B3) + np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
        cols = np.arange(ncols, dtype=np.uint64)
        r = _splitmix64(base[:, None] + cols[None, :] * np.uint64(0x9E3779B97F4A7C15))
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def read_windows(blob, l1, l2, seed, first_pair=0, sub=0.05, ins=0.02, dele=0.02):
    """The C4 input of SURVEY.md 8(d): for every other pair (even pair index) the read (s1) is a mutated window of its contig
    (s2) -- a window start uniform over the contig, then 5 % substitutions, 2 % inserted and 2 % deleted bases, l1 bases in
    the end; the other pairs keep their unrelated uniform read.  Deterministic per pair index."""
    n = blob.shape[0]
    out = blob.copy()
    if l2 <= l1 + 16:
        return out
    u = _uniform01(seed, first_pair, n, 1, 1)[:, 0]
    start = (u * (l2 - l1 - 16)).astype(np.int64) + 4
    is_del = _uniform01(seed, first_pair, n, l1, 2) < dele      # the contig base before this read base is skipped
    is_ins = _uniform01(seed, first_pair, n, l1, 3) < ins       # this read base is new, the contig does not advance
    src = start[:, None] + np.arange(l1)[None, :] + np.cumsum(is_del, axis=1) - np.cumsum(is_ins, axis=1)
    src = np.clip(src, 0, l2 - 1)
    contig = blob[:, l1:l1 + l2]
    read = np.take_along_axis(contig, src, axis=1)
    rnd = _uniform01(seed, first_pair, n, l1, 4)
    newbase = _ACGT[(rnd * 4).astype(np.int64) & 3]
    change = is_ins | (_uniform01(seed, first_pair, n, l1, 5) < sub)
    read = np.where(change, newbase, read)
    even = ((np.arange(first_pair, first_pair + n) & 1) == 0)
    out[even, :l1] = read[even]
    return out


def workload_blob(mode, use_jump, seed, npairs, l1, l2, first_pair=0):
    """The synthetic batch of one bench workload: uniform ACGT pairs; for fit with the jump state (C4) every other read is a
    mutated window of its contig."""
    blob = synth_pairs_blob(seed, npairs, l1, l2, first_pair)
    if mode == "fit" and use_jump:
        blob = read_windows(blob, l1, l2, seed, first_pair)
    return blob


# ---- the same generators on a torch device: full-size workloads (C4: 10 M pairs = 6.5 GB of bases) are made where they are
# ---- used, in HBM; bit-identical to the numpy functions above (tests/test_synth.py) ----
def _s64(c):
    """a 64-bit constant as the signed value torch's int64 holds"""
    c &= 0xFFFFFFFFFFFFFFFF
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(x, k):
    """logical shift right of int64 lanes"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _splitmix64_t(x):
    x = x + _s64(0x9E3779B97F4A7C15)
    z = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def synth_codes_torch(seed, first_pair, npairs, nbases, device):
    """synth_codes on `device`: uint8 tensor [npairs, nbases] of base codes 0..3"""
    import torch
    idx = torch.arange(first_pair, first_pair + npairs, dtype=torch.int64, device=device)
    s = _splitmix64_t(idx + _s64(seed))
    s = torch.where(s == 0, torch.full_like(s, _s64(0x9E3779B97F4A7C15)), s)
    nwords = (nbases + 31) // 32
    out = torch.empty((npairs, nwords * 32), dtype=torch.uint8, device=device)
    shifts = (torch.arange(32, dtype=torch.int64, device=device) * 2)[None, :]
    for w in range(nwords):
        s = s ^ _lsr(s, 12)
        s = s ^ (s << 25)
        s = s ^ _lsr(s, 27)
        r = s * _s64(0x2545F4914F6CDD1D)
        out[:, w * 32:(w + 1) * 32] = ((r[:, None] >> shifts) & 3).to(torch.uint8)
    return out[:, :nbases]


def _uniform01_t(seed, first_pair, npairs, ncols, stream, device):
    import torch
    idx = torch.arange(first_pair, first_pair + npairs, dtype=torch.int64, device=device)
    base = _splitmix64_t(idx * _s64(0x100000001B3) + _s64(seed) + _s64(stream * 0xD1B54A32D192ED03))
    cols = torch.arange(ncols, dtype=torch.int64, device=device)
    r = _splitmix64_t(base[:, None] + cols[None, :] * _s64(0x9E3779B97F4A7C15))
    return _lsr(r, 11).to(torch.float64) * (1.0 / (1 << 53))


def workload_codes_torch(mode, use_jump, seed, npairs, l1, l2, first_pair, device, sub=0.05, ins=0.02, dele=0.02):
    """workload_blob as base CODES (0..3, uint8 [npairs, l1 + l2]) on `device`; ASCII = b"ACGT"[codes]"""
    import torch
    codes = synth_codes_torch(seed, first_pair, npairs, l1 + l2, device)
    if not (mode == "fit" and use_jump) or l2 <= l1 + 16:
        return codes
    n = npairs
    u = _uniform01_t(seed, first_pair, n, 1, 1, device)[:, 0]
    start = (u * (l2 - l1 - 16)).to(torch.int64) + 4
    is_del = _uniform01_t(seed, first_pair, n, l1, 2, device) < dele
    is_ins = _uniform01_t(seed, first_pair, n, l1, 3, device) < ins
    src = start[:, None] + torch.arange(l1, device=device)[None, :] + torch.cumsum(is_del, dim=1) - torch.cumsum(is_ins, dim=1)
    src = src.clamp(0, l2 - 1)
    contig = codes[:, l1:l1 + l2]
    read = torch.gather(contig, 1, src)
    rnd = _uniform01_t(seed, first_pair, n, l1, 4, device)
    newbase = ((rnd * 4).to(torch.int64) & 3).to(torch.uint8)
    change = is_ins | (_uniform01_t(seed, first_pair, n, l1, 5, device) < sub)
    read = torch.where(change, newbase, read)
    even = ((torch.arange(first_pair, first_pair + n, device=device) & 1) == 0)
    out = codes.clone()
    out[:, :l1] = torch.where(even[:, None], read, codes[:, :l1])
    return out


def pack2_torch(codes):
    """2-bit words of at_pack_batch for every row of a uint8 code tensor [n, L]: int32 [n, ceil(L / 16) + 1] (base k of a row in bits
    [2k % 32, 2k % 32 + 1] of word k / 16; one more word, which the kernels' windows read ahead into)"""
    import torch
    n, L = codes.shape
    nw = (L + 15) // 16
    pad = torch.zeros((n, (nw + 1) * 16), dtype=torch.int64, device=codes.device)
    pad[:, :L] = codes
    sh = (torch.arange(16, dtype=torch.int64, device=codes.device) * 2)[None, None, :]
    w = (pad.view(n, nw + 1, 16) << sh).sum(dim=2)
    return w.to(torch.int32)   # (values of 2^31 and more wrap into the sign bit: the same 32 bits)
