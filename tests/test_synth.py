"""CPU suite: the torch forms of the synthetic generators (made for full-size workloads generated in HBM) are bit-identical to the
numpy ones the bench and the parity tests have always used, and pack2_torch packs like at_pack_batch."""
import numpy as np
import torch

import aligntools.c_amd as A
from aligntools.c_amd import synth as S


def test_torch_generators_equal_numpy():
    dev = torch.device("cpu")
    for seed, first, n, l1, l2 in ((0x5EED0002, 0, 50, 150, 150), (0x5EED0004, 0, 64, 150, 500), (0x5EED0004, 12345, 33, 150, 500), (7, 3, 5, 40, 70)):
        a = S.synth_codes(seed, first, n, l1 + l2)
        b = S.synth_codes_torch(seed, first, n, l1 + l2, dev).numpy()
        assert (a == b).all()
        for mode, uj in (("fit", True), ("local", False)):
            blob = S.workload_blob(mode, uj, seed, n, l1, l2, first_pair=first)
            codes = S.workload_codes_torch(mode, uj, seed, n, l1, l2, first, dev).numpy()
            assert (np.frombuffer(b"ACGT", dtype=np.uint8)[codes] == blob).all(), (seed, mode)
    u = S._uniform01(99, 5, 7, 11, 3)
    assert (u == S._uniform01_t(99, 5, 7, 11, 3, dev).numpy()).all()


def test_pack2_torch_equals_at_pack_batch():
    rng = np.random.default_rng(3)
    for L1, L2 in ((150, 500), (16, 32), (1, 17), (33, 31)):
        codes = rng.integers(0, 4, size=(9, L1 + L2), dtype=np.uint8)
        asc = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
        pairs = [(row[:L1].tobytes(), row[L1:].tobytes()) for row in asc]
        words, woff1, woff2, len1, len2, bits = A.pack_pairs(pairs, bits=2)
        w1 = S.pack2_torch(torch.from_numpy(codes[:, :L1])).numpy().view(np.uint32)
        w2 = S.pack2_torch(torch.from_numpy(codes[:, L1:])).numpy().view(np.uint32)
        for k in range(9):
            assert (words[woff1[k]:woff1[k] + w1.shape[1]] == w1[k]).all()
            assert (words[woff2[k]:woff2[k] + w2.shape[1]] == w2[k]).all()
