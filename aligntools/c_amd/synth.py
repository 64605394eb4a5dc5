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
