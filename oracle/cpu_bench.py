"""cpu_baseline leg of bench.py -- TEST/BENCH INFRASTRUCTURE (runs the checker,
never the product).  Times the REAL reference compiled in oracle/_ref
(kind "reference") or, if that binary is absent, the restatement (kind "port")
on a bounded sample of the bench workload, one pair per call, on T forked
worker processes.  Prints one JSON object.

    python -m oracle.cpu_bench --mode local --pairs 4000 --l1 150 --l2 150 --threads 8
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="local")
    ap.add_argument("--pairs", type=int, default=4000)
    ap.add_argument("--l1", type=int, default=150)
    ap.add_argument("--l2", type=int, default=150)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5EED0002)
    ap.add_argument("--scoring", default="2,-2,-5,-2,-10")
    ap.add_argument("--use-jump", type=int, default=0)
    ap.add_argument("--sites", default="")
    a = ap.parse_args()
    import oracle as O
    from aligntools.c_amd.synth import workload_blob
    m, u, o, e, j = [int(x) for x in a.scoring.split(",")]
    sites = [int(x) for x in a.sites.split("|")] if a.sites else []
    T = a.threads or min(os.cpu_count() or 1, 16)
    blob = workload_blob(a.mode, bool(a.use_jump), a.seed, a.pairs, a.l1, a.l2).tobytes()   # the same batch bench.py sweeps on the GPU
    mode = O.MODE_NAMES[a.mode]
    kind = "reference" if O.have_ref() else "port"
    per = a.l1 + a.l2
    # 1-core leg on a slice
    n1 = max(1, a.pairs // T)
    t1, _, _ = O.time_batch(mode, blob[: n1 * per], n1, a.l1, a.l2, m, u, o, e, j, a.use_jump, sites, kind)
    # T-process leg: fork workers, each times its own chunk, wall = slowest
    chunks = [(k * a.pairs // T, (k + 1) * a.pairs // T) for k in range(T)]
    pipes = []
    t0 = time.monotonic()
    for lo, hi in chunks:
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            dt, chk, _ = O.time_batch(mode, blob[lo * per: hi * per], hi - lo, a.l1, a.l2, m, u, o, e, j, a.use_jump, sites, kind)
            os.write(w, json.dumps([dt, chk]).encode())
            os._exit(0)
        os.close(w)
        pipes.append((pid, r))
    inner = []
    for pid, r in pipes:
        data = b""
        while True:
            chunk = os.read(r, 4096)
            if not chunk:
                break
            data += chunk
        os.close(r)
        os.waitpid(pid, 0)
        inner.append(json.loads(data.decode())[0])
    wall = time.monotonic() - t0
    cells = float(a.pairs) * a.l1 * a.l2
    print(json.dumps(dict(kind=kind, cores=T, pairs=a.pairs, l1=a.l1, l2=a.l2, wall_s=wall, slowest_worker_s=max(inner),
                          gcups=cells / wall / 1e9, gcups_1core=float(n1) * a.l1 * a.l2 / t1 / 1e9)))


if __name__ == "__main__":
    main()
