"""bench.py's own launcher: `python bench.py --gpus N` with no WORLD_SIZE starts N fresh ranks from a parent that
makes no GPU call (the driver starts the scaling runs exactly this way)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return env


def test_more_ranks_than_gpus_is_refused_cleanly():
    """One rank per GPU over RCCL: asking for more ranks than the machine has GPUs ends non-zero with a message
    (here: no GPU at all, or the GPU box's single one), and starts nothing."""
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2
    assert "one rank per GPU" in r.stderr and r.stdout.strip() == ""


def test_workloads_name_the_metric_cohort():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    kind, variants, samples, scaling, desc = bench.WORKLOADS[bench.parse_args([]).workload]
    assert (kind, variants, samples, scaling) == ("chisq", 10_000_000, 50_000, "strong")    # BASELINE.json metric
    assert bench.parse_args([]).gpus == 1


@pytest.mark.gpu
def test_self_launched_ranks_gloo_rehearsal():
    """Two self-launched ranks sharing the box's GPU (gloo: a rehearsal of the N > 1 code path, not a measurement):
    strong-scaled shards, two tiles per rank, result blocks gathered on rank 0 and checked against the oracle there."""
    for extra in ([], ["--resident", "no"]):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--variants", "200001", "--samples", "3001",
                            "--tile-gb", "0.2", "--steps", "2", "--warmup", "1"] + extra, env=_env(), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["variants"] == 200001
        assert d["config"]["tiles_per_gpu"] == 2 and d["config"]["resident"] == (not extra)
        assert d["parity"]["ok"] and d["parity"]["blocks"] == 3
