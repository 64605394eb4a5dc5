"""bench.py keeps the driver's contract: ONE JSON line on stdout with the agreed keys (a short run on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"]


@pytest.mark.gpu
def test_bench_prints_one_json_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--pairs", "30000", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    assert d["unit"] == "GCUPS" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] in ("int16", "int32")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the path is bound by VALU issue: achieved / peak in wave-instructions per second, priced with the
    # counters committed under profiles/ (none for a --pairs 30000 batch: frac is null then); the HBM view rides along
    assert r["bound"] == "valu" and r["unit"] == "G wave-instr/s"
    if r["frac"] is not None:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    hb = r["hbm"]
    assert hb["unit"] == "GB/s" and hb["peak"] == 8000.0 and abs(hb["frac"] - hb["achieved"] / hb["peak"]) < 1e-12
    assert d["value"] > 100 and abs(d["ms_per_step"] * d["value"] - 30000 * 150 * 150 / 1e6) < 1e-3 * d["ms_per_step"] * d["value"]


def test_bench_gpus_n_starts_its_own_launcher():
    """`python bench.py --gpus N` with no launcher in the environment (the way the driver runs N = 1) must not fail: it starts
    `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` as a child.  Without a GPU: the dry run names it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_LAUNCH_DRYRUN"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and p.stdout.strip() == "", p.stderr[-2000:]
    assert "-m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port" in p.stderr
    assert p.stderr.rstrip().endswith("bench.py --gpus 4 --steps 7 --warmup 2")


def test_bench_refuses_more_ranks_than_gpus_over_rccl():
    """RCCL takes one rank per device: asking for more ranks than the node has GPUs is refused before anything starts."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_LAUNCH_DRYRUN")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 2 and "one rank per device" in p.stderr and p.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_runs_two_ranks():
    """The driver's launch form for N > 1 may be the plain `python bench.py --gpus N`: bench.py is then its own launcher.  Two
    ranks share the test box's card over gloo (RCCL refuses that); the JSON line says who took part."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--pairs", "20000", "--steps", "5",
                        "--warmup", "2", "--gather-every", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    g = d["config"]["gather"]
    assert d["n_gpus"] == 2 and g["world"] == 2 and [r["rank"] for r in g["ranks"]] == [0, 1] and g["ranks"][0]["pid"] != g["ranks"][1]["pid"]
    assert g["backend"].startswith("gloo") and d["cpu_baseline"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend,extra", [(1, "nccl", ["--steps", "11", "--warmup", "3", "--gather-every", "4"]),
                                                 (2, "gloo", ["--steps", "7", "--warmup", "2", "--gather-every", "3", "--workload", "C4"])])
def test_bench_under_the_launcher_gathers_groups_of_steps(world, backend, extra):
    """The way the driver starts N > 1: `python -m torch.distributed.run … bench.py --gpus N`.  RCCL with a world of one (RCCL
    refuses two ranks on one device) and gloo with two ranks sharing the test box's card run the whole pipeline -- scoring
    broadcast, grouped fixed-size gathers, one CIGAR compaction and one padded payload per group, a last group of one step --
    and bench.py's own checks (own block arrived, sizes = gathered nops, payload = ops slots) must hold on every rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 29810 + world
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--pairs", "20000", "--no-cpu-baseline",
                        "--backend", backend] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["value"] > 0
    g = d["config"]["gather"]
    assert g["steps_per_collective"] == int(extra[extra.index("--gather-every") + 1]) and g["cigar_bytes_per_rank_and_step"] > 0
    assert g["world"] == world and len(g["ranks"]) == world and g["backend"].startswith("nccl (RCCL" if backend == "nccl" else "gloo")


def test_committed_counters_belong_to_todays_kernel_sources():
    """bench.py prices its roofline with the PMC counters under profiles/<round>/traffic_<W>[_scores].json only while they carry the
    fingerprint of the sweep kernels' sources (valu_roofline: otherwise achieved / peak / frac are null and `stale` says why).  A
    kernel edit without a new collection (tools/collect_traffic.py on the GPU box) is caught here, on the CPU, not by a driver bench
    line without a roofline."""
    import importlib.util
    import aligntools.c_amd as A
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = A.kernel_source_sha16()
    for w, tb in (("C2", True), ("C2", False), ("C3", True), ("C4", True), ("C5", True)):
        prof = bench._traffic(w, tb)
        assert prof is not None, (w, tb)
        assert prof.get("kernel_source_sha16") == sha, "profiles/%s/traffic_%s%s.json was collected on other kernel sources (%s, today %s)" % (
            bench.PROFILE_ROUND, w, "" if tb else "_scores", prof.get("kernel_source_sha16"), sha)
        assert (prof.get("sq_counters_per_launch") or {}).get("SQ_INSTS_VALU"), (w, tb)
