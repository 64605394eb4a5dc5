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


def test_bench_refuses_gpus_without_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "torch.distributed.run" in p.stderr and p.stdout.strip() == ""


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
