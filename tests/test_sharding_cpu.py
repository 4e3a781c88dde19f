"""CPU suite: the multi-GPU path's variant sharding and result gather, run with
world_size 2 and 3 over gloo (the GPU box uses the same code over nccl = RCCL).
The per-shard compute stand-in here is the oracle (test infrastructure)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, V, N, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    sh = import_module("hpg-variant_amd.sharding")
    from oracle import pyoracle as orc
    lo, hi = sh.variant_range(rank, world, V)
    n = hi - lo
    cond = (np.arange(N) % 2).astype(np.uint8)
    gt = orc.synth_matrix(lo, n, N, N)                    # this rank's global variant ids
    A1, A2, U1, U2 = orc.assoc_counts(gt, cond)
    odds, chisq, p = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
    lay = sh.result_block_layout(n)
    block = np.zeros(lay["bytes"], dtype=np.uint8)
    block[: 16 * n] = np.stack([A1, A2, U1, U2], 1).astype(np.int32).reshape(-1).view(np.uint8)
    for key, arr in (("odds", odds), ("chisq", chisq), ("p", p)):
        block[lay[key]: lay[key] + 8 * n] = arr.view(np.uint8)
    sizes = [40 * (sh.variant_range(r, world, V)[1] - sh.variant_range(r, world, V)[0]) for r in range(world)]
    blocks, work = sh.gather_blocks(torch.from_numpy(block), sizes, dst=0, async_op=True)
    if work is not None:
        work.wait()
    if rank == 0:
        counts, stats = [], []
        for r, b in enumerate(blocks):
            nr = sizes[r] // 40
            b = b.numpy()
            counts.append(b[: 16 * nr].view(np.int32).reshape(nr, 4))
            stats.append(b[16 * nr:].view(np.float64).reshape(3, nr))
        np.save(os.path.join(out_dir, "counts.npy"), np.concatenate(counts, 0))
        np.save(os.path.join(out_dir, "stats.npy"), np.concatenate(stats, 1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,V", [(2, 1000), (3, 1001), (2, 1)])
def test_sharded_scan_and_gather_equals_single_process(tmp_path, world, V):
    from oracle import pyoracle as orc
    N = 64
    mp.spawn(_worker, args=(world, _free_port(), V, N, str(tmp_path)), nprocs=world, join=True)
    counts = np.load(tmp_path / "counts.npy")
    stats = np.load(tmp_path / "stats.npy")
    cond = (np.arange(N) % 2).astype(np.uint8)
    gt = orc.synth_matrix(0, V, N, N)
    A1, A2, U1, U2 = orc.assoc_counts(gt, cond)
    odds, chisq, p = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
    assert np.array_equal(counts, np.stack([A1, A2, U1, U2], 1))
    for got, exp in ((stats[0], odds), (stats[1], chisq), (stats[2], p)):
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert np.allclose(got[~np.isnan(got)], exp[~np.isnan(exp)], rtol=0, atol=0)


def test_variant_ranges_partition_everything():
    from importlib import import_module
    sh = import_module("hpg-variant_amd.sharding")
    for V in (0, 1, 7, 1000, 10_000_000):
        for G in (1, 2, 4, 8):
            r = [sh.variant_range(g, G, V) for g in range(G)]
            assert r[0][0] == 0 and r[-1][1] == V
            assert all(r[i][1] == r[i + 1][0] for i in range(G - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _reduce_worker(rank, world, port, V, N, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    sh = import_module("hpg-variant_amd.sharding")
    from oracle import pyoracle as orc
    lo, hi = sh.variant_range(rank, world, V)
    gt = orc.synth_matrix(lo, hi - lo, N, N)
    miss = torch.from_numpy(orc.sample_missing(gt).astype(np.int32)) if hi > lo else torch.zeros(N, dtype=torch.int32)
    sh.reduce_sample_counters(miss, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "miss.npy"), miss.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,V", [(2, 500), (3, 301)])
def test_sample_counters_reduce_over_variant_shards(tmp_path, world, V):
    from oracle import pyoracle as orc
    N = 96
    mp.spawn(_reduce_worker, args=(world, _free_port(), V, N, str(tmp_path)), nprocs=world, join=True)
    assert np.array_equal(np.load(tmp_path / "miss.npy"), orc.sample_missing(orc.synth_matrix(0, V, N, N)))


def test_pair_row_bands_partition_the_triangle_evenly():
    from importlib import import_module
    sh = import_module("hpg-variant_amd.sharding")
    for V in (2, 63, 64, 65, 1000, 16384, 100_000):
        for G in (1, 2, 4, 8):
            bands = [sh.pair_row_range(g, G, V) for g in range(G)]
            assert bands[0][0] == 0 and bands[-1][1] == V
            assert all(bands[i][1] == bands[i + 1][0] for i in range(G - 1))
            assert all(lo % 64 == 0 or lo == V for lo, _ in bands)
            pairs = [sum(V - 1 - r for r in range(lo, hi)) if V <= 1000 else (hi * (2 * V - hi - 1) - lo * (2 * V - lo - 1)) // 2 for lo, hi in bands]
            assert sum(pairs) == V * (V - 1) // 2
            if V >= 16384:
                assert max(pairs) <= 1.05 * (sum(pairs) / G) + 64 * V


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("V,strong", [(10_000_000, True), (200_001, True), (7, True), (1_000_000, False)])
def test_tile_plan_covers_every_variant_once(world, V, strong):
    """bench.py's shard / tile plan: every variant of the cohort lies in exactly one (rank, tile); every rank walks the same
    number of tiles; no tile exceeds the buffer; the per-rank block sizes every rank computes agree with the ranks' own tiles."""
    from importlib import import_module
    sh = import_module("hpg-variant_amd.sharding")
    row, cap_bytes = 50_016, 126_000_000_000 if V > 1000 else 3 * 50_016
    plans = [sh.plan_tiles(r, world, V, row, cap_bytes, strong) for r in range(world)]
    assert len({p["n_tiles"] for p in plans}) == 1 and len({p["per_tile"] for p in plans}) == 1
    seen = []
    for r, p in enumerate(plans):
        assert p["per_tile"] * row <= cap_bytes or p["per_tile"] == 1
        for t, (lo, hi) in enumerate(p["tiles"]):
            assert 0 <= lo <= hi <= p["n"] and hi - lo <= p["per_tile"]
            assert hi - lo == p["counts"][t][r]
            assert p["counts"][t] == plans[0]["counts"][t]
            seen.append((p["v_lo"] + lo, p["v_lo"] + hi))
    seen = sorted(x for x in seen if x[1] > x[0])
    total = V if strong else V * world
    assert seen[0][0] == 0 and seen[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    if V == 10_000_000:                                   # the metric cohort: 4 / 2 / 1 / 1 tiles of <= 126 GB at 1 / 2 / 4 / 8 ranks
        assert plans[0]["n_tiles"] == {1: 4, 2: 2, 3: 2, 4: 1, 8: 1}[world]


def test_whole_genome_config_is_tiled_per_rank():
    """BASELINE configs[4] (40M SNP x 100k samples on 8 GPUs) as bench.py --workload c5full plans it: 5M variants per rank in four
    tiles of at most 126 GB, every rank the same number of tiles, all variants covered once."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    sharding = importlib.import_module("hpg-variant_amd.sharding")
    kind, V, N, scaling, _ = bench.WORKLOADS["c5full"]
    assert (kind, V, N, scaling) == ("chisq", 40_000_000, 100_000, "strong")
    pitch = 100_016
    seen = 0
    for rank in range(8):
        plan = sharding.plan_tiles(rank, 8, V, pitch, int(126e9), strong=True)
        assert plan["n"] == 5_000_000 and plan["n_tiles"] == 4 and plan["per_tile"] * pitch <= 126e9
        assert plan["v_lo"] == rank * 5_000_000
        assert sum(hi - lo for lo, hi in plan["tiles"]) == plan["n"]
        seen += plan["n"]
    assert seen == V
