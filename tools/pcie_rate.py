"""H2D / D2H rates from page-locked memory on the GPU box (what bounds the host entry): one copy of 1 / 5 / 30 MB, and
30 MB as six 5 MB copies on six streams side by side -- the shape of at_align_batch's chunks."""
import time

import torch

dev = torch.device("cuda", 0)
torch.cuda.init()


def rate(nbytes, nstreams, h2d=True, reps=20):
    per = nbytes // nstreams
    hs = [torch.empty(per, dtype=torch.uint8).pin_memory() for _ in range(nstreams)]
    ds = [torch.empty(per, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    ss = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for h, d, s in zip(hs, ds, ss):
            with torch.cuda.stream(s):
                if h2d:
                    d.copy_(h, non_blocking=True)
                else:
                    h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return nbytes / best / 1e9, best * 1e3


for mb in (1, 5, 30):
    for ns in (1, 6):
        if mb == 1 and ns == 6:
            continue
        g, ms = rate(mb << 20, ns)
        g2, ms2 = rate(mb << 20, ns, h2d=False)
        print("%2d MB on %d stream(s): H2D %5.1f GB/s (%.3f ms), D2H %5.1f GB/s (%.3f ms)" % (mb, ns, g, ms, g2, ms2))
