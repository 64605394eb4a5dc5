#!/usr/bin/env python3
"""bench.py -- GCUPS of the batched affine-gap DP hot path on MI355X.

Workload at N=1 = BASELINE.json configs[1] ("C2"): Smith-Waterman local, affine
gap, 100k synthetic 150x150 bp pairs, scores, tracebacks (ops strings) AND the
reference's two gapped strings per pair rendered on the GPU, m=2 u=-2 o=-5 e=-2
(SURVEY.md 8(d)).  A "step" is one pass of the hot path
over the whole batch with the packed inputs already resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: pairs are independent, so each rank aligns its own 100k-pair shard
(weak scaling); rank 0 broadcasts the scoring block over RCCL before the timed
region and every step's results are gathered to all ranks over RCCL, the
results of `--gather-every` (8) consecutive steps in one collective (fewer,
larger collectives, all of them on a communication stream of their own behind
events of the sweep streams): the fixed-size part (score, end cell, state,
CIGAR length: 20 B/pair) with an asynchronous all_gather; the CIGARs (ops
strings) in two phases -- compacted on the GPU step by step; sizes first (they
are the gathered CIGAR lengths: every rank sums them per peer and step), then
one payload per group, padded to the largest, that travels while the next
group computes.

Steps alternate over `--streams` HIP streams (default 3), each with its own
handle (workspace, work queue) and output buffers, so up to three launches are
in flight: the SIMDs that the draining tail of step k leaves idle are taken by
step k+1 (one launch alone spends 17 % of its time in its first and last round,
profiles/LOG.md C).  Every step still computes its whole batch; `--streams 1` runs
them strictly one after the other.

Prints ONE JSON line (rank 0).  `roofline` prices the sweep kernel against the
HBM roofline with ALGORITHMIC bytes (packed inputs + descriptors read, results +
ops written) over the HIP-event duration of the launch;
`cpu_baseline` is the real reference (oracle/_ref, kind "reference") or the
oracle restatement (kind "port") timed on the host cores, N=1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_ROUND = "r04"   # profiles/<round>/traffic_<workload>.json, valu_issue.json: the counters the roofline is priced with


def _traffic(workload, tb):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic_%s%s.json" % (workload, "" if tb else "_scores"))))
    except Exception:
        return None


def valu_roofline(workload, pairs, tb, kernel_config, steps, elapsed, cells_per_pair, src_sha):
    """The binding roofline of this path is VALU issue (integer add/max on packed int16: no HBM or MFMA bound comes near,
    DESIGN.md section 4).  achieved = SQ_INSTS_VALU of one launch (rocprofv3 --pmc on this workload and kernel
    configuration, profiles/<round>/traffic_<W>[_scores].json) x launches / timed seconds; peak = SIMDs x clock / cycles per
    wave64 instruction, both measured by tools/valu_issue.hip on the MI355X (profiles/<round>/valu_issue.json: the packed
    16-bit, bit-field and DPP instructions the step body is made of occupy a SIMD for 4 cycles, and so does everything
    else once it is mixed with them; the clock is what the chip held under that load, not the nominal 2.4 GHz).

    The counters belong to ONE state of the kernels: they are used only if the profile carries the fingerprint of today's
    kernel sources (aligntools.c_amd.kernel_source_sha16), the same kernel configuration string and the same batch size --
    otherwise achieved / peak / frac are null and `stale` says why.

    `frac` prices an instruction at the 4-cycle class this repository measured; MI355X_MICROARCH.md documents only the 2-cycle
    class of plain VOP2 integer adds, and `frac_at_documented_2_cycles` is the same quotient against that nominal rate.  What
    neither fraction shows -- a kernel with twice the instructions would score the same -- is in `insts_per_cell` and
    `insts_per_alignment`, with the scores-only kernel's figures beside them (the instructions a traceback costs)."""
    prof = _traffic(workload, tb)
    vi = vi_src = None
    for rnd in (PROFILE_ROUND, "r03", "r02"):   # the issue rates are a property of the chip: measured in round 2, measured again when a round re-runs tools/valu_issue.hip
        try:
            vi_src = "profiles/%s/valu_issue.json" % rnd
            vi = json.load(open(os.path.join(ROOT, vi_src)))
            break
        except Exception:
            vi = None
    if prof is None or vi is None:
        return None, None
    insts = (prof.get("sq_counters_per_launch") or {}).get("SQ_INSTS_VALU")
    traffic = prof.get("hbm_bytes_per_launch")
    stale = None
    if not insts:
        stale = "no SQ_INSTS_VALU in the profile"
    elif prof.get("kernel_source_sha16") != src_sha:
        stale = "kernel sources changed since the counters were collected (profile %s, sources %s)" % (prof.get("kernel_source_sha16"), src_sha)
    elif prof.get("kernel_config") != kernel_config:
        stale = "another kernel configuration (AT_GROUP, --l1 ...)"
    elif prof.get("pairs") != pairs:
        stale = "another batch size"
    if stale:
        return dict(bound="valu", achieved=None, peak=None, unit="G wave-instr/s", frac=None, stale=stale,
                    source="profiles/%s/traffic_%s%s.json" % (PROFILE_ROUND, workload, "" if tb else "_scores")), None
    # packed int16 kernels: every instruction of the step body is of the 4-cycle class; so are the bit-parallel edit kernel's
    # (v_bitop3_b32, v_alignbit_b32) and the two instructions per cell of the int32 ramp sweeps (overlap scores only, cell-by-cell
    # edit distance: v_add_u32_sdwa, v_max3_i32) -- classes the microbenchmark lists when it has measured them; the other int32
    # kernels mix 2-cycle adds in and are priced with the full-rate 2.0 (the only bound that holds for any mix)
    cpi = vi["roofline"]["cycles_per_inst"]
    mode = WORKLOADS[workload][0]
    cls = ("packed16" if "packed16" in kernel_config else "bitparallel" if kernel_config.startswith("myers")
           else "int32_ramp" if (mode == "edit" or (mode == "overlap" and not tb)) else "int32")
    cyc = cpi.get(cls, cpi["int32"])
    ghz = vi["roofline"]["clock_ghz"]
    simds = vi["roofline"]["simds"]
    peak = simds * ghz / cyc                      # G wave-instructions / s
    achieved = insts * steps / elapsed / 1e9
    out = dict(bound="valu", achieved=achieved, peak=peak, unit="G wave-instr/s", frac=achieved / peak,
               frac_at_documented_2_cycles=achieved / (simds * ghz / 2.0),
               valu_insts_per_launch=insts, cycles_per_inst=cyc, cycles_per_inst_class=cls, cycles_per_inst_source=vi_src + " (this repository's microbenchmark tools/valu_issue.hip; "
               "MI355X_MICROARCH.md documents the 2-cycle class only)", clock_ghz=ghz, simds=simds,
               insts_per_alignment=insts / pairs, insts_per_cell=insts / (pairs * cells_per_pair), kernel_source_sha16=src_sha,
               source="profiles/%s/traffic_%s%s.json + %s" % (PROFILE_ROUND, workload, "" if tb else "_scores", vi_src))
    # the same class from the production kernel's own counters: cycles the launch was resident on the chip (GRBM_GUI_ACTIVE, summed
    # over the 8 XCDs) per VALU instruction of one SIMD -- what the microbenchmark's 4-cycle class predicts if the SIMDs issue all the time
    sq = prof.get("sq_counters_per_launch") or {}
    if sq.get("GRBM_GUI_ACTIVE") and insts:
        out["cycles_per_inst_from_counters"] = (sq["GRBM_GUI_ACTIVE"] / 8.0) / (insts / float(simds))
        out["cycles_per_inst_from_counters_source"] = "GRBM_GUI_ACTIVE / 8 XCDs over SQ_INSTS_VALU / %d SIMDs of the profiled launch (one launch at a time, --streams 1)" % simds
    other = _traffic(workload, not tb)
    oi = ((other or {}).get("sq_counters_per_launch") or {}).get("SQ_INSTS_VALU")
    if oi and other.get("kernel_source_sha16") == src_sha and other.get("pairs") == pairs:
        key = "scores_only" if tb else "with_tracebacks"
        out["insts_per_alignment_" + key] = oi / pairs
        out["insts_per_cell_" + key] = oi / (pairs * cells_per_pair)
    return out, traffic

WORKLOADS = {
    # name: (mode, l1, l2, pairs per GPU, scoring m,u,o,e,j, use_jump, sites, seed)
    "C2": ("local", 150, 150, 100000, (2, -2, -5, -2, -10), False, [], 0x5EED0002),
    "C3": ("global", 1024, 1024, 10000, (1, -1, -4, -1, -10), False, [], 0x5EED0003),
    "C4": ("fit", 150, 500, 100000, (2, -2, -5, -1, -10), True, [100, 200, 300, 400], 0x5EED0004),
    "C5": ("overlap", 1000, 1000, 10000, (1, -2, -5, -1, -10), False, [], 0x5EED0005),
    # diagnostics: C2's shape under global (walks that cross the whole read), C4's without the jump state
    "G150": ("global", 150, 150, 100000, (2, -2, -5, -2, -10), False, [], 0x5EED0012),
    "F500": ("fit", 150, 500, 100000, (2, -2, -5, -1, -10), False, [], 0x5EED0014),
    # all-vs-all over 50k reads of 1 kbp (1.25e9 ordered pairs in full): each step scores a 100k-pair slice of the
    # triangle per GPU, pairs enumerated on the GPU (at_align_allpairs_device), scores + end cells only
    "C5all": ("overlap", 1000, 1000, 100000, (1, -2, -5, -1, -10), False, [], 0x5EED0005),
    # BASELINE configs[3] and [4] at the sizes they state, on ONE GPU (SURVEY.md 8(d) quotes them for 8): 10 M fit pairs in one launch per
    # step; the whole 50 000-read triangle (1 249 975 000 pairs) per step, in slices of 8 Mi pairs through at_align_allpairs_device.
    # Minutes per run: `--workload C4full --steps 5 --warmup 1`, `--workload C5full --steps 1 --warmup 0 --streams 1`
    "C4full": ("fit", 150, 500, 10000000, (2, -2, -5, -1, -10), True, [100, 200, 300, 400], 0x5EED0004),
    "C5full": ("overlap", 1000, 1000, 8 << 20, (1, -2, -5, -1, -10), False, [], 0x5EED0005),
    # edit distance with unit mismatch cost (`edit -u 1`): the bit-parallel kernel; "cells" are the DP cells it stands for
    "E1k": ("edit", 1000, 1000, 100000, (1, 1, -5, -1, -10), False, [], 0x5EED0006),
    "E150": ("edit", 150, 150, 400000, (1, 1, -5, -1, -10), False, [], 0x5EED0007),
}


def cpu_baseline(mode, l1, l2, scoring, use_jump, sites, seed, target_s=15.0):
    """Run oracle/cpu_bench.py in a child process (never touches the GPU)."""
    ns_per_cell = 25e-9
    threads = min(os.cpu_count() or 1, 16)
    pairs = int(max(threads * 4, min(200000, target_s * threads / (ns_per_cell * l1 * l2))))
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--mode", mode, "--pairs", str(pairs), "--l1", str(l1), "--l2", str(l2),
           "--threads", str(threads), "--seed", hex(seed), "--scoring", ",".join(str(x) for x in scoring),
           "--use-jump", "1" if use_jump else "0", "--sites", "|".join(str(x) for x in sites)]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as ex:   # the baseline is a reported number, not a gate
        return dict(value=None, unit="GCUPS", cores=0, kind="port", sample="failed: %r" % (ex,))
    multi, single = d["gcups"], d["gcups_1core"]
    best_multi = multi >= single
    return dict(value=multi if best_multi else single, unit="GCUPS", cores=d["cores"] if best_multi else 1, kind=d["kind"],
                sample="%d pairs of %dx%d %s, fill+traceback, one pair per call" % (pairs, l1, l2, mode),
                gcups_all_cores=multi, gcups_1core=single, host_cores_used=d["cores"])


def launch_ranks(n, backend):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>` as a child process (never
    an exec), pass its stderr through, relay the one JSON line rank 0 prints, return its exit code."""
    import socket
    try:   # the devices are counted by a short-lived child: the launcher itself never opens the GPU (a box may cap the processes per card)
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        have = int(r.stdout.strip().splitlines()[-1])
    except Exception:
        have = 0
    if backend == "nccl" and have < n and not os.environ.get("BENCH_LAUNCH_DRYRUN"):
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s): RCCL takes one rank per device "
                         "(--backend gloo rehearses the pipeline with ranks sharing a card)\n" % (n, have))
        return 2
    with socket.socket() as so:   # a free rendezvous port on the loopback
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("BENCH_LAUNCH_DRYRUN"):   # (tests without a GPU: what would be started)
        sys.stderr.write("bench.py: would start: %s\n" % " ".join(cmd))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:      # rank 0's JSON line goes to stdout; anything else a library printed there goes to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traceback", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to rehearse the N>1 pipeline with ranks sharing one GPU")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams (and handles) the steps alternate over: with 2, the head of step k+1 fills the SIMDs "
                         "that the draining tail of step k leaves idle")
    ap.add_argument("--bits", type=int, default=0, choices=[0, 2, 8],
                    help="sequence words: 0 = 2-bit when the batch is pure ACGT (it is), 8 = force byte words (the kernels for "
                         "reads with N / protein)")
    ap.add_argument("--l1", type=int, default=0, help="override the workload's first length (diagnostic)")
    ap.add_argument("--l2", type=int, default=0, help="override the workload's second length (diagnostic)")
    ap.add_argument("--no-jump", action="store_true", help="fit without the jump state (diagnostic: C4's shape on the plain fit kernels)")
    ap.add_argument("--no-uniform-promise", action="store_true",
                    help="call at_align_batch_device with uniform_shape = 0: the device checks the shapes itself (diagnostic)")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: steps whose results travel in one collective (fixed-size results and CIGAR payload each)")
    ap.add_argument("--no-cigar-gather", action="store_true", help="N > 1: gather only the fixed-size results (diagnostic)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: no per-step collective at all (diagnostic)")
    ap.add_argument("--min-score", type=int, default=None,
                    help="C5all / C5full: all-vs-all overlap scores with a threshold (at_set_min_score): pairs proven to score below it by the "
                         "bit-parallel bound are not swept, the others are exact")
    ap.add_argument("--no-render", action="store_true",
                    help="stop at the op codes (CIGARs): do not also turn them into the reference's two gapped strings on the GPU "
                         "inside every step (at_render_batch_device)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # launched by torch.distributed.run (any N)
    if args.gpus > 1 and not use_dist:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has made no GPU call and makes
        # none; the ranks are its grandchildren (torch.distributed.run, one rank per GPU), their one JSON line is relayed.
        sys.exit(launch_ranks(args.gpus, args.backend))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d (one rank per GPU: `python bench.py --gpus N` starts its own "
                 "torch.distributed.run, or start it under one with --nproc-per-node N)" % (args.gpus, world))
    mode, l1, l2, pairs, scoring, use_jump, sites, seed = WORKLOADS[args.workload]
    if args.pairs:
        pairs = args.pairs
    if args.no_jump:
        use_jump, sites = False, []
    l1, l2 = args.l1 or l1, args.l2 or l2

    # CPU baseline first (child process, before this process touches the GPU)
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(mode, l1, l2, scoring, use_jump, sites, seed)

    import numpy as np
    import torch
    import torch.distributed as dist
    import aligntools.c_amd as A
    from aligntools.c_amd.synth import synth_pairs_blob, workload_blob

    local_rank %= max(1, torch.cuda.device_count())   # (a rehearsal may put several ranks on one card)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its communicator comes up: keep stdout to the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- scoring block: rank 0 owns it, RCCL broadcast to the other ranks ----
    # (two broadcasts: the fixed block with the number of sites, then exactly that many sites -- no fixed-size slot)
    sc = torch.tensor(list(scoring) + [1 if use_jump else 0, len(sites)], dtype=torch.int32, device=dev)
    st = torch.tensor(list(sites), dtype=torch.int32, device=dev)
    if use_dist:
        if rank != 0:
            sc.zero_()
        dist.broadcast(sc, src=0)
        ns = int(sc[6].item())
        if ns:
            if rank != 0:
                st = torch.zeros(ns, dtype=torch.int32, device=dev)
            dist.broadcast(st, src=0)
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    ranks_info = None
    if use_dist:
        # who took part: every rank's device, as the process group sees it (config.gather.ranks of the JSON line)
        pr = torch.cuda.get_device_properties(local_rank)
        me = {"rank": rank, "pid": os.getpid(), "device": local_rank, "name": pr.name,
              "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    scl = sc.cpu().tolist()
    m, u, o, e, j, uj, ns = scl[:7]
    site_list = st.cpu().tolist()[:ns]
    allpairs_w = args.workload in ("C5all", "C5full")
    S = max(1, args.streams)
    LAG = S                                          # the CIGAR payload of step k is sent when its stream comes round again
    NB = S + 2                                       # buffer sets, used in turn
    als = [A.Aligner(local_rank) for _ in range(S)]  # one handle (workspace, work queue) per stream
    for x in als:
        x.set_scoring(m, u, o, e, j, bool(uj), site_list)
        if args.min_score is not None and allpairs_w:
            x.set_min_score(args.min_score)
    al = als[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]

    # ---- this rank's shard of the synthetic batch, packed, resident in HBM ----
    allpairs = args.workload in ("C5all", "C5full")
    full_triangle = args.workload == "C5full"
    ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
    from aligntools.c_amd.synth import workload_codes_torch, pack2_torch
    if allpairs:
        nreads = 50000
        blob = synth_pairs_blob(seed, nreads // 2, l1, l2)           # 25k rows of two 1 kbp reads
        plist = [(row[k * l1:(k + 1) * l1].tobytes(), b"") for row in blob for k in range(2)]
        words, woff1, woff2, len1, len2, bits = A.pack_pairs(plist, bits=args.bits)
        d_words = torch.from_numpy(words.view(np.int32)).to(dev)
        d_woff1, d_woff2, d_len1, d_len2 = (torch.from_numpy(x).to(dev) for x in (woff1, woff2, len1, len2))
        sample_rows = {}
    elif args.bits == 8:
        blob = workload_blob(mode, use_jump, seed, pairs, l1, l2, first_pair=rank * pairs)   # (C4: every other read is a mutated window of its contig)
        plist = [(row[:l1].tobytes(), row[l1:].tobytes()) for row in blob]
        words, woff1, woff2, len1, len2, bits = A.pack_pairs(plist, bits=args.bits)
        d_words = torch.from_numpy(words.view(np.int32)).to(dev)
        d_woff1, d_woff2, d_len1, d_len2 = (torch.from_numpy(x).to(dev) for x in (woff1, woff2, len1, len2))
        sample_rows = {k: plist[k] for k in range(0, pairs, max(1, pairs // 64))}
    else:
        # generated and packed where it is used, in HBM (aligntools/c_amd/synth.py: the torch forms are bit-identical to the numpy
        # generators, tests/test_synth.py): 10 M pairs take seconds here and a quarter of an hour in numpy
        bits = 2
        w1, w2 = (l1 + 15) // 16 + 1, (l2 + 15) // 16 + 1
        d_words = torch.zeros(pairs * (w1 + w2) + 4, dtype=torch.int32, device=dev)
        wv = d_words[:pairs * (w1 + w2)].view(pairs, w1 + w2)
        sample_rows, want = {}, set(range(0, pairs, max(1, pairs // 64)))
        for lo in range(0, pairs, 500000):
            nn = min(500000, pairs - lo)
            c = workload_codes_torch(mode, use_jump, seed, nn, l1, l2, rank * pairs + lo, dev)   # (C4: every other read is a mutated window of its contig)
            wv[lo:lo + nn, :w1] = pack2_torch(c[:, :l1])
            wv[lo:lo + nn, w1:] = pack2_torch(c[:, l1:])
            for k in sorted(x for x in want if lo <= x < lo + nn):
                row = ACGT[c[k - lo].cpu().numpy()]
                sample_rows[k] = (row[:l1].tobytes(), row[l1:].tobytes())
            del c
        d_woff1 = torch.arange(pairs, dtype=torch.int64, device=dev) * (w1 + w2)
        d_woff2 = d_woff1 + w1
        d_len1 = torch.full((pairs,), l1, dtype=torch.int32, device=dev)
        d_len2 = torch.full((pairs,), l2, dtype=torch.int32, device=dev)
    words_nbytes = d_words.numel() * 4
    tb = (not args.no_traceback) and mode != "edit" and not allpairs
    ops_off = np.arange(pairs, dtype=np.int64) * (l1 + l2)
    d_ops_off = torch.from_numpy(ops_off).to(dev)
    # score, end_i, end_j, state, nops (the CIGAR lengths) of a step: [5, pairs] int32.
    # N = 1: NB buffer sets used in turn.  N > 1: the results of G consecutive steps form one group buffer [G, 5, pairs] that is
    # gathered with ONE collective (fewer, larger collectives; the sweep streams carry no communication at all), NGB sets in turn.
    G = max(1, args.gather_every) if use_dist else 1
    NGB = 3
    grp_of, row_of = [], []      # step -> its group and its row in the group's buffers (plan_groups, per run)
    if use_dist:
        grp_res = [torch.zeros((G, 5, pairs), dtype=torch.int32, device=dev) for _ in range(NGB)]
        res_of = lambda k: grp_res[grp_of[k] % NGB][row_of[k]]
    else:
        d_res2 = [torch.zeros((5, pairs), dtype=torch.int32, device=dev) for _ in range(NB)]
        res_of = lambda k: d_res2[k % NB]
    slot_bytes = pairs * (l1 + l2)
    cig = use_dist and tb and not args.no_gather and not args.no_cigar_gather
    if cig:   # the ops slots of a whole group stay until the group's CIGARs have been compacted (one compaction per group)
        grp_ops = [torch.zeros(G * slot_bytes + 64, dtype=torch.uint8, device=dev) for _ in range(NGB)]
        ops_of = lambda k: grp_ops[grp_of[k] % NGB][row_of[k] * slot_bytes:]
    else:
        d_opss = [torch.zeros(slot_bytes + 64, dtype=torch.uint8, device=dev) if tb else None for _ in range(S)]
        ops_of = lambda k: d_opss[k % S]
    rend = tb and not args.no_render
    d_r1s = [torch.zeros(pairs * (l1 + l2) + 64, dtype=torch.uint8, device=dev) if rend else None for _ in range(S)]
    d_r2s = [torch.zeros(pairs * (l1 + l2) + 64, dtype=torch.uint8, device=dev) if rend else None for _ in range(S)]
    gath = use_dist and not args.no_gather
    # CIGAR gather: the ops slots of a whole group are compacted on the GPU by ONE compaction on the communication
    # stream (the sweep streams carry sweeps and rendering only); the per-rank, per-step totals come out of the fixed-size gather
    # (sums of the gathered CIGAR lengths); then ONE payload per group, padded to the largest group total.  The payload of group g
    # travels while group g + 1 computes.
    if gath:
        comm_stream = torch.cuda.Stream(device=dev)      # all communication is queued here, behind events of the sweep streams
        gathered = [torch.empty((world, G, 5, pairs), dtype=torch.int32, device=dev) for _ in range(NGB)]
        busy = [[] for _ in range(NGB)]                    # the collectives that still read or write each buffer set
        grp_rows = [0] * NGB                               # steps in the group that last used the set (the last group may be short)
        step_ev = [torch.cuda.Event() for _ in range(S)]   # "this stream has finished its latest step"
        dist.all_gather_into_tensor(gathered[0].view(world * G, 5, pairs), grp_res[0])   # set-up, not a step: the first collective of a shape builds RCCL's channels (~0.1 s)
        torch.cuda.synchronize()
    if cig:
        cap = G * slot_bytes
        al_comm = A.Aligner(local_rank)                    # its workspace holds the compaction's scan
        grp_packed = [torch.zeros(cap + 4096, dtype=torch.uint8, device=dev) for _ in range(NGB)]
        grp_nops = [torch.zeros((G, pairs), dtype=torch.int32, device=dev) for _ in range(NGB)]   # the group's nops rows, contiguous
        grp_ops_off = torch.arange(G * pairs, dtype=torch.int64, device=dev) * (l1 + l2)
        d_poff = [torch.zeros(G * pairs + 1, dtype=torch.int64, device=dev) for _ in range(NGB)]
        alltot = [torch.zeros((world, G), dtype=torch.int64, device=dev) for _ in range(NGB)]
        h_tot = [torch.zeros((world, G), dtype=torch.int64).pin_memory() for _ in range(NGB)]
        tot_ev = [torch.cuda.Event() for _ in range(NGB)]
        allpay = [None] * NGB                              # (sized on first use)
        pay_pad = [0] * NGB

    tri_sum = torch.zeros(NB, dtype=torch.int64, device=dev)   # C5full: sum of the scores of the triangle, per buffer set

    def step(k):
        d_res = res_of(k)
        if full_triangle:
            tri_sum[k % NB] = 0
        al, d_ops = als[k % S], ops_of(k)
        stream = torch.cuda.current_stream().cuda_stream
        if gath:   # the collectives of the group that last used this buffer set must be done before a step writes into it
            for wk in busy[grp_of[k] % NGB]:
                wk.wait()
        if full_triangle:   # the whole triangle (this rank's contiguous share of it), slice after slice into the same result buffers
            tri = nreads * (nreads - 1) // 2
            lo_r, hi_r = (tri * rank + world - 1) // world, min(tri, (tri * (rank + 1) + world - 1) // world)
            for first in range(lo_r, hi_r, pairs):
                nn = min(pairs, hi_r - first)
                al.align_allpairs_device(A.MODES[mode], nreads, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_len1.data_ptr(), l1,
                                         first, nn, False, d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(),
                                         d_res[3].data_ptr(), None, None, None, stream)
                tri_sum[k % NB] += d_res[0, :nn].sum(dtype=torch.int64)      # (every slice's scores are read before the next one overwrites them)
        elif allpairs:   # this rank's slice of the triangle, a different one every step
            first = ((k * world + rank) * pairs) % (nreads * (nreads - 1) // 2 - pairs)
            al.align_allpairs_device(A.MODES[mode], nreads, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_len1.data_ptr(), l1,
                                     first, pairs, False, d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(),
                                     d_res[3].data_ptr(), None, None, None, stream)
        else:
            al.align_batch_device(A.MODES[mode], pairs, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_len1.data_ptr(),
                                  d_woff2.data_ptr(), d_len2.data_ptr(), l1, l2, not args.no_uniform_promise, tb,
                                  d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(), d_res[3].data_ptr(),
                                  d_ops.data_ptr() if tb else None, d_ops_off.data_ptr() if tb else None,
                                  d_res[4].data_ptr() if tb else None, stream)
        return d_res

    def finish(k, d_res):
        al, d_ops, d_r1, d_r2 = als[k % S], ops_of(k), d_r1s[k % S], d_r2s[k % S]
        if rend:       # ops (END -> START) -> the reference's two strings, in HBM
            al.render_batch_device(pairs, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_woff2.data_ptr(), d_res[1].data_ptr(),
                                   d_res[2].data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), d_res[4].data_ptr(),
                                   d_r1.data_ptr(), d_r2.data_ptr(), None, False, torch.cuda.current_stream().cuda_stream)
        if not gath:
            return
        step_ev[k % S].record()

    def gather_group(g, rows):
        """Phase 1 for group g (`rows` steps): one all_gather of the fixed-size results (row 4 = nops: the CIGAR sizes), queued on the
        communication stream behind the sweep streams' latest steps; then every rank's payload size per step = the sum of its
        gathered nops, brought to the host."""
        gb = g % NGB
        grp_rows[gb] = rows
        for ev in step_ev:
            comm_stream.wait_event(ev)
        with torch.cuda.stream(comm_stream):
            w = dist.all_gather_into_tensor(gathered[gb].view(world * G, 5, pairs), grp_res[gb], async_op=True)
            if cig:
                # the group's CIGARs, back to back in (step, pair) order: one compaction over rows x pairs slots
                grp_nops[gb].copy_(grp_res[gb][:, 4, :])
                al_comm.compact_ops_device(rows * pairs, grp_ops[gb].data_ptr(), grp_ops_off.data_ptr(), grp_nops[gb].data_ptr(),
                                           grp_packed[gb].data_ptr(), cap, d_poff[gb].data_ptr(), comm_stream.cuda_stream)
                w.wait()
                torch.sum(gathered[gb][:, :, 4, :].clamp(min=0), dim=2, dtype=torch.int64, out=alltot[gb])
                h_tot[gb].copy_(alltot[gb], non_blocking=True)
                tot_ev[gb].record()
        busy[gb] = [w]

    def payload_group(g):
        """Phase 2 for group g, issued one group later: by then its sizes are on the host (the host never waits for the newest
        launches, so a whole group of them stays queued), and the transfer overlaps the sweeps of the group behind it."""
        gb = g % NGB
        rows = grp_rows[gb]
        tot_ev[gb].synchronize()
        pad = (max(int(h_tot[gb][:, :rows].sum(dim=1).max()), 1) + 4095) // 4096 * 4096   # the largest group total of any rank
        assert pad <= cap + 4096
        pay_pad[gb] = pad
        if allpay[gb] is None or allpay[gb].numel() < world * pad:   # (first use, or a group with longer CIGARs than any before)
            torch.cuda.synchronize()                    # (happens in the warm-up: every set is sized at once)
            room = min(cap + 4096, (pad + pad // 4) // rows * G)
            for x in range(NGB):
                if allpay[x] is None or allpay[x].numel() < world * room:
                    allpay[x] = torch.empty(world * room, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(comm_stream):
            busy[gb].append(dist.all_gather_into_tensor(allpay[gb][:world * pad], grp_packed[gb][:pad], async_op=True))

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def plan_groups(nsteps):
        """Groups of G consecutive steps; the last step is a group of its own, so that what remains to be done once the last sweep
        has finished (fixed-size gather, compaction, sizes to the host, payload) handles one step's results, not G."""
        del grp_of[:], row_of[:]
        k = 0
        while k < nsteps:
            n = min(G, nsteps - k)
            if k + n == nsteps and n > 1:
                n -= 1
            grp_of.extend([len(set(grp_of))] * n)
            row_of.extend(range(n))
            k += n

    def run(nsteps, timed_events=None):
        """Queue nsteps steps over the S streams; N > 1: gather every group of steps and wait for every collective before returning."""
        plan_groups(nsteps)
        for k in range(nsteps):
            with torch.cuda.stream(streams[k % S]):
                if timed_events is not None:
                    timed_events[k][0].record()
                r = step(k)
                if timed_events is not None:
                    timed_events[k][1].record()
                finish(k, r)
            if gath and (k == nsteps - 1 or grp_of[k + 1] != grp_of[k]):
                g = grp_of[k]
                gather_group(g, row_of[k] + 1)
                if cig and g >= 1:
                    payload_group(g - 1)
        t_issued = time.perf_counter()   # (the host has queued every step; what remains is the GPU draining them)
        if gath and nsteps > 0:
            if cig:
                payload_group(grp_of[nsteps - 1])
            for works in busy:
                for wk in works:
                    wk.wait()
        return t_issued

    run(args.warmup)
    sync_all()
    if gath:   # (the timed run starts with every buffer set free)
        for works in busy:
            del works[:]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    t_issued = run(args.steps, evs)
    sync_all()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # ---- checks + accounting (outside the timed region) ----
    kern_ms = sorted(a.elapsed_time(b) for a, b in evs)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)
    # the same launch with nothing else in flight (what a rocprofv3 trace of `--streams 1` shows)
    iso = []
    for k in range(0 if full_triangle else 4):   # (a step of C5full is the whole triangle, minutes long: its timed steps were alone already)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        step(args.steps - 1)   # the last step again, into the same buffers
        eb.record()
        torch.cuda.synchronize()
        iso.append(ea.elapsed_time(eb))
    kern_iso_ms = min(iso) if iso else kern_ms[0]
    # every step of the timed run holds the results of the same batch
    last = args.steps - 1
    if not allpairs and args.steps > 0:   # (all-vs-all sweeps a different slice of the triangle every step)
        for k in range(max(0, last - (G if use_dist else NB) + 1), last):
            assert torch.equal(res_of(k), res_of(last)), "buffer sets disagree"
    d_ops = ops_of(last)
    scores = res_of(last)[0].cpu().numpy()
    nops = res_of(last)[4].cpu().numpy() if tb else np.zeros(pairs, dtype=np.int32)
    gather_info = None
    if gath and args.steps > 0:
        # what arrived: every rank's block of the last group, and in it this rank's own results
        gb, row, rows = grp_of[last] % NGB, row_of[last], row_of[last] + 1
        gnp = gathered[gb].cpu().numpy()
        assert (gnp[rank, :rows] == grp_res[gb][:rows].cpu().numpy()).all(), "fixed-size gather: own block differs"
        try:
            bver = ".".join(str(x) for x in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        except Exception:
            bver = None
        gather_info = {"world": world, "backend": "nccl (RCCL %s)" % bver if args.backend == "nccl" else "gloo (rehearsal: ranks may share a card)",
                       "ranks": ranks_info, "steps_per_collective": G, "fixed_bytes_per_rank_and_step": 20 * pairs}
        if cig:
            # every rank's total per step is the sum of its gathered nops, and this rank's part of the payload is its own ops
            # slots back to back
            tots = h_tot[gb].numpy()
            assert (gnp[:, :rows, 4, :].clip(min=0).sum(axis=2) == tots[:, :rows]).all(), "CIGAR gather: sizes disagree with the gathered nops"
            pad = pay_pad[gb]
            at = int(tots[rank, :row].sum())       # the group's payload is in (step, pair) order: the last step's CIGARs start here
            mine = allpay[gb][:world * pad].view(world, pad)[rank, at:at + int(tots[rank, row])].cpu().numpy()
            slots = d_ops[:pairs * (l1 + l2)].cpu().numpy().reshape(pairs, l1 + l2)
            assert (mine == slots[np.arange(l1 + l2)[None, :] < nops[:, None]]).all(), "CIGAR gather: payload differs from the ops slots"
            gather_info.update({"cigar_bytes_per_rank_and_step": int(tots[rank, row]), "cigar_group_payload_padded_to": int(pad),
                                "phases": "per group of %d steps: one all_gather of the fixed-size results (sizes), then one padded "
                                          "payload; both on a communication stream, the payload of group g beside the sweeps of group g + 1" % G})
    assert (scores > -(1 << 30)).all() and (nops >= 0).all(), "kernel reported a domain error"
    if rend and args.steps > 0:
        # the GPU-rendered strings of a sample equal what at_render (host, one pair) makes of the same ops
        h_r1 = d_r1s[last % S].cpu().numpy().tobytes()
        h_r2 = d_r2s[last % S].cpu().numpy().tobytes()
        h_ops = d_ops[:slot_bytes].cpu().numpy()
        h_res = res_of(last).cpu().numpy()
        for k in sorted(sample_rows):
            oo, nk = int(ops_off[k]), int(nops[k])
            a, b = al.render(h_ops[oo:oo + nk].tobytes(), sample_rows[k][0], int(h_res[1][k]), sample_rows[k][1], int(h_res[2][k]))
            assert a.encode("latin1") == h_r1[oo:oo + nk] and b.encode("latin1") == h_r2[oo:oo + nk], "rendered strings differ"
    cells_per_step = float(pairs) * l1 * l2 * world
    if full_triangle:
        cells_per_step = float(nreads * (nreads - 1) // 2) * l1 * l2     # the whole triangle, whatever the number of ranks (strong scaling)
    gcups = cells_per_step * args.steps / elapsed / 1e9
    # algorithmic HBM bytes of one launch on one GPU
    bytes_in = words_nbytes + pairs * (8 + 8 + 4 + 4) + (pairs * 8 if tb else 0)
    bytes_out = pairs * 16 + (pairs * 4 + int(nops.sum()) if tb else 0)
    step_s = elapsed / max(1, args.steps)            # (launches overlap: a launch's own first-to-last-wave time is not a per-step figure)
    achieved = (bytes_in + bytes_out) / step_s / 1e9
    valu, traffic = valu_roofline(args.workload, pairs, tb, al.last_config, args.steps, elapsed, float(l1) * l2, A.kernel_source_sha16())

    if rank == 0:
        out = {
            "metric": "GCUPS (DP cell updates/s), SW affine-gap 150bp pairs" if args.workload == "C2"
                      else "GCUPS (DP cell updates/s), %s %dx%d" % (mode, l1, l2),
            "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if full_triangle else "weak",
            "vs_baseline": None, "dtype": "int16" if "packed16" in al.last_config else "u32" if al.last_config.startswith("myers") else "int32", "data": "synthetic",
            "config": {"workload": "%s: %s %s, %d x (%dx%d bp) pairs per GPU, uniform ACGT, m=%d u=%d o=%d e=%d%s, "
                                   "%s" % (args.workload, mode, "linear-gap" if mode == "overlap" else "unit-gap" if mode == "edit" else "affine-gap", pairs, l1, l2,
                                           m, u, o, e, " j=%d -s" % j if uj else "",
                                           ("scores+tracebacks+rendered strings" if rend else "scores+tracebacks") if tb else
                                           ("scores of the pairs that may reach %d (the others: an upper bound below it)" % args.min_score
                                            if (args.min_score is not None and allpairs_w) else "scores only")),
                       "pairs_per_gpu": pairs, "l1": l1, "l2": l2, "bits_per_base": bits, "kernel_config": al.last_config,
                       "streams": S,
                       "parallelism": "pairs sharded over %d GPU(s), one process per GPU" % world},
            "roofline": dict(valu or {"bound": "valu", "achieved": None, "peak": None, "unit": "G wave-instr/s", "frac": None,
                                      "note": "no SQ_INSTS_VALU profile for this kernel configuration under profiles/%s" % PROFILE_ROUND},
                             traffic=traffic,
                             kernel=("at_sweep16" if "packed16" in al.last_config else "at_myers" if "myers" in al.last_config else "at_sweep") + "<%s>" % mode +
                                    (" + at_walk16" if "walk kernel" in al.last_config else ""),
                             kernel_avg_ms=kern_avg_ms, kernel_min_ms=kern_ms[0], kernel_alone_ms=kern_iso_ms, launches_in_flight=S,
                             host_issue_ms_per_step=(t_issued - t0) * 1e3 / max(1, args.steps),
                             gcups_one_launch_at_a_time=cells_per_step / world / (kern_iso_ms * 1e-3) / 1e9,
                             # the HBM view BASELINE.json asks for, both ways: the ALGORITHMIC bytes of a launch over the time per step, and the
                             # bytes the chip really moved (PMC FETCH_SIZE x 2 + WRITE_SIZE of the same kernel configuration) over it
                             hbm={"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "achieved_counter": (traffic / step_s / 1e9) if traffic else None,
                                  "frac_counter": (traffic / step_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                                  "algorithmic_bytes_per_launch": bytes_in + bytes_out,
                                  "counter_bytes_per_launch": traffic,
                                  "traffic_over_algorithmic": (traffic / float(bytes_in + bytes_out)) if traffic else None}),
            "cpu_baseline": base,
        }
        if gather_info:
            out["config"]["gather"] = gather_info
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
