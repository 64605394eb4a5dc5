#!/usr/bin/env python3
"""collect_traffic.py -- per-launch PMC counters of the sweep kernel of each bench workload (runs on the GPU box).

    python3 tools/collect_traffic.py [--out gpurun_out/r02/pmc] [--round 2] C2 C3 C4 C5

For every workload: separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; the SQ instruction counters;
GRBM_GUI_ACTIVE rides with the SQ pass) over `python3 bench.py --workload W --streams 1 --steps 3 --warmup 1
--no-render --no-cpu-baseline`, no trace flags next to --pmc.  The rows of the sweep kernel (the dispatch with the largest
WRITE_SIZE / instruction count is the sweep; render/compact/pack kernels are listed separately) are averaged per
dispatch and written as traffic_<W>.json, in the shape bench.py reads from profiles/ (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950: the counter tallies 128-byte requests at 64 bytes).
This script itself never touches the GPU: rocprofv3 starts `python3 bench.py` directly.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
            "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"]),
    ("sq2", ["SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_SMEM", "GRBM_GUI_ACTIVE"]),
]


def run_pass(workload, tag, counters, outdir, extra):
    d = os.path.join(outdir, "%s_%s" % (workload, tag))
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["-d", d, "-o", "out", "--output-format", "csv", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--workload", workload, "--streams", "1", "--steps", "3", "--warmup", "1",
           "--no-render", "--no-cpu-baseline"] + extra
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    open(os.path.join(d, "log.txt"), "w").write(r.stdout[-4000:] + "\n----\n" + r.stderr[-4000:])
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    line = None   # the bench line of the profiled run: which kernel configuration and batch size the counters belong to
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            try:
                line = json.loads(ln)
            except ValueError:
                pass
    return r.returncode, rows, line


def per_kernel(rows):
    """{kernel name: {counter: [values per dispatch]}}"""
    out = {}
    for r in rows:
        k = r.get("Kernel_Name", "?")
        out.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="+")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04", "pmc"))
    ap.add_argument("--round", type=int, default=4)
    ap.add_argument("--no-traceback", action="store_true")
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)   # rocprofv3 runs with cwd=/tmp
    os.makedirs(args.out, exist_ok=True)
    extra = ["--no-traceback"] if args.no_traceback else []
    for w in args.workloads:
        res = {"workload": w, "round": args.round, "command": "rocprofv3 --pmc <counters> -- python3 bench.py --workload %s --streams 1 "
               "--steps 3 --warmup 1 --no-render --no-cpu-baseline%s" % (w, " --no-traceback" if extra else ""),
               "passes": {}, "kernels": {}}
        for tag, counters in PASSES:
            rc, rows, line = run_pass(w, tag, counters, args.out, extra)
            if line:
                res["kernel_config"] = line["config"]["kernel_config"]
                res["pairs"] = line["config"]["pairs_per_gpu"]
                res["profiled_run_ms_per_step"] = line["ms_per_step"]
            res["passes"][tag] = {"rc": rc, "rows": len(rows), "counters": counters}
            pk = per_kernel(rows)
            for k, cs in pk.items():
                e = res["kernels"].setdefault(k, {})
                for c, vals in cs.items():
                    e[c] = {"mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals), "dispatches": len(vals)}
            # keep the per-dispatch rows small: only the columns that matter
            with open(os.path.join(args.out, "%s_%s.csv" % (w, tag)), "w") as fh:
                fh.write("Dispatch_Id,Grid_Size,Workgroup_Size,VGPR_Count,Kernel_Name,Counter_Name,Counter_Value\n")
                for r in rows:
                    fh.write("%s,%s,%s,%s,\"%s\",%s,%s\n" % (r.get("Dispatch_Id"), r.get("Grid_Size"), r.get("Workgroup_Size"),
                                                           r.get("VGPR_Count"), r.get("Kernel_Name"), r["Counter_Name"], r["Counter_Value"]))
            print(w, tag, "rc", rc, "rows", len(rows), flush=True)
        # the sweep kernel = the at_sweep* / at_myers* kernel with the most VALU instructions
        sweeps = {k: v for k, v in res["kernels"].items() if "at_sweep" in k or "at_myers" in k}
        if sweeps:
            name = max(sweeps, key=lambda k: sweeps[k].get("SQ_INSTS_VALU", {}).get("mean", 0.0))
            s = sweeps[name]
            g = lambda c: s.get(c, {}).get("mean")
            res["kernel"] = name
            if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
                res["FETCH_SIZE_KB"] = g("FETCH_SIZE")
                res["WRITE_SIZE_KB"] = g("WRITE_SIZE")
                res["hbm_bytes_per_launch"] = int(round(2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024))
                res["note"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte read requests at 64 bytes); WRITE_SIZE as read"
            res["sq_counters_per_launch"] = {c: s[c]["mean"] for c in s if c.startswith("SQ_") or c.startswith("GRBM")}
            # two-pass tracebacks with pass 2 as a kernel of its own (at_walk16): a step is the sweep AND its walk kernel -- their counters added
            walks = {k: v for k, v in res["kernels"].items() if "at_walk16" in k}
            if walks:
                wname = max(walks, key=lambda k: walks[k].get("SQ_INSTS_VALU", {}).get("mean", 0.0))
                wk = walks[wname]
                res["walk_kernel"] = wname
                res["per_kernel"] = {"sweep": {"hbm_bytes": res.get("hbm_bytes_per_launch"), "SQ_INSTS_VALU": g("SQ_INSTS_VALU")},
                                     "walk": {"SQ_INSTS_VALU": wk.get("SQ_INSTS_VALU", {}).get("mean")}}
                if "FETCH_SIZE" in wk and "WRITE_SIZE" in wk and "hbm_bytes_per_launch" in res:
                    wb = int(round(2 * wk["FETCH_SIZE"]["mean"] * 1024 + wk["WRITE_SIZE"]["mean"] * 1024))
                    res["per_kernel"]["walk"]["hbm_bytes"] = wb
                    res["hbm_bytes_per_launch"] += wb
                for c in list(res["sq_counters_per_launch"]):
                    if c in wk:
                        res["sq_counters_per_launch"][c] += wk[c]["mean"]
        # the counters belong to this state of the kernels: bench.py checks the fingerprint before it prices a roofline with them
        sys.path.insert(0, ROOT)
        from aligntools.c_amd import kernel_source_sha16
        res["kernel_source_sha16"] = kernel_source_sha16()
        res["kernels"] = {k: v for k, v in res["kernels"].items() if "at_" in k}   # (ours; torch's fill / copy kernels of the bench are noise)
        path = os.path.join(os.path.dirname(args.out), "traffic_%s%s.json" % (w, "_scores" if extra else ""))
        json.dump(res, open(path, "w"), indent=1)
        print("wrote", path, flush=True)


if __name__ == "__main__":
    sys.exit(main())
