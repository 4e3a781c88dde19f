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


def test_parent_counts_gpus_without_hip():
    """The parent of the ranks finds the GPUs in sysfs / /dev, never through HIP or torch (a process that has touched the
    GPU must not be the one that starts the ranks)."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    src = open(BENCH).read()
    body = src[src.index("def count_gpus_no_hip"):src.index("# helpers of a rank")]
    assert "import torch" not in body and "hipGetDeviceCount" not in body
    n = bench.count_gpus_no_hip()
    assert n >= 0
    if not os.path.exists("/dev/kfd"):
        assert n == 0


@pytest.mark.gpu
def test_one_real_rccl_rank_through_the_bench_path():
    """`--gpus 1 --backend nccl --force-process-group`: init_process_group(nccl, world 1) + every result block through
    sharding.gather_blocks -> dist.gather on the GPU, double-buffered and overlapped as at N > 1; rank 0 checks its own
    blocks AND the block it received through RCCL against the oracle."""
    for extra in ([], ["--resident", "no"]):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--backend", "nccl", "--force-process-group", "--variants", "200001",
                            "--samples", "3001", "--tile-gb", "0.4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra,
                           env=_env(), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["tiles_per_gpu"] == 2
        assert d["parity"]["ok"] and d["parity"]["blocks"] == 3          # first tile, last tile, last tile as gathered
        assert "configs" not in d


@pytest.mark.gpu
def test_default_line_carries_every_baseline_config():
    """The one driver-timed line: the metric workload's fields at the top, c2 / c3 / c4 / stats under "configs", each with its
    own roofline, parity and cpu_baseline (here on a small main workload so the test stays short)."""
    r = subprocess.run([sys.executable, BENCH, "--workload", "smoke", "--configs", "all", "--steps", "3", "--warmup", "1",
                        "--config-steps", "3", "--config-warmup", "1", "--cpu-seconds", "0.5", "--config-cpu-seconds", "0.5"],
                       env=_env(), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(d["configs"]) == {"c2", "c3", "c4", "stats"}
    for name, c in d["configs"].items():
        assert c["parity"]["ok"], name
        assert c["roofline"]["frac"] is None or 0.0 < c["roofline"]["frac"] < 1.0, name
        assert c["cpu_baseline"]["value"] > 0 and c["value"] > 0
    assert d["configs"]["c3"]["roofline"]["bound"] == "valu" and d["configs"]["c2"]["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c2", "c3", "c4", "stats"])
def test_single_process_group_mode(workload):
    """`--single-process`: one process, the C ABI's group context ([0, 0] on the one-GPU box), results gathered on member 0
    through hpgv_group_*, checked against the oracle over both members' shards; rccl_ranks comes from the communicator."""
    r = subprocess.run([sys.executable, BENCH, "--single-process", "--devices", "0,0", "--workload", workload, "--variants", "60001",
                        "--samples", "1503", "--steps", "4", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 1 and d["config"]["variants"] == 2 * 60001
    assert d["parity"]["ok"] and d["parity"]["checked_variants"] > 100 and d["value"] > 0
