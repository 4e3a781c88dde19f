"""The north star's variant shards + final gather BEHIND THE C ABI (hpgv_group_*): a group context owns the devices, the
streams and the RCCL communicator (ncclCommInitAll), member g scans [g*V/G, (g+1)*V/G), results land on member 0.
A one-GPU box can check: the communicator is really created (one rank), the shard split, the result placement, the
two-generation overlap, and -- with the test switch group_self_exchange -- real ncclSend / ncclRecv / ncclReduce traffic.
Everything is compared BIT FOR BIT with one ordinary context scanning all variants, and against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import TOL, hpgv
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    hpgv.build()
    exe = str(tmp_path_factory.mktemp("grp") / "group_scan")
    lib = os.path.join(ROOT, "hpg-variant_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "group_scan.c"), "-o", exe, "-L", lib, "-lhpgv",
                           "-Wl,-rpath," + lib, "-lm"])
    return exe


@pytest.fixture(scope="module")
def epi_driver(tmp_path_factory):
    hpgv.build()
    exe = str(tmp_path_factory.mktemp("grp") / "group_epi")
    lib = os.path.join(ROOT, "hpg-variant_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "group_epi.c"), "-o", exe, "-L", lib, "-lhpgv",
                           "-Wl,-rpath," + lib, "-lm"])
    return exe


@pytest.mark.parametrize("members,v,nA,nU,k,order,extra", [
    (1, 300, 400, 380, 5, 2, []),          # one member: the whole triangle, its list copied into place
    (1, 300, 400, 380, 5, 2, ["self"]),    # ... through ncclSend / ncclRecv to itself
    (2, 300, 400, 380, 5, 2, []),          # [0, 0]: two row bands, merged
    (3, 700, 150, 170, 10, 2, []),         # [0, 0, 0]
    (3, 100, 150, 170, 4, 2, []),          # fewer 64-row blocks than members want: an empty share
    (3, 60, 120, 140, 4, 3, []),           # triples dealt by first SNP
    (2, 60, 120, 140, 4, 3, ["self"]),
    (3, 16, 90, 110, 3, 4, []),            # order 4: the listed-combination kernel per member
    (2, 12, 90, 110, 3, 5, ["self"]),
])
def test_c_host_deals_the_epistasis_scan_to_a_group(epi_driver, members, v, nA, nU, k, order, extra):
    r = subprocess.run([epi_driver, str(members), str(v), str(nA), str(nU), str(k), str(order)] + extra, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "rccl_ranks=1" in r.stdout and "group epistasis ok" in r.stdout and r.stdout.count("bit-identical") == 2


def test_group_epistasis_against_the_dense_evaluation():
    """The group's ranking of a [0, 0, 0] group from Python, against the dense evaluation of every pair on one context."""
    from helpers import epi_random_dataset, epi_random_folds
    rng = np.random.default_rng(8)
    v, nA, nU, k, n = 200, 300, 260, 6, 15
    data = epi_random_dataset(rng, v, nA, nU)
    fold = epi_random_folds(rng, nA, nU, k)
    g = hpgv.Engine([0, 0, 0])
    g.epi_set_dataset(data, nA, nU)
    g.epi_set_folds(fold, k)
    shares = [g.group_epi_share(2, m) for m in range(3)]
    assert shares[0][0] == 0 and shares[-1][1] == v and all(a[1] == b[0] for a, b in zip(shares, shares[1:]))
    res = g.group_epi_rank(2, hpgv.EPI_TESTING, n)
    one = hpgv.Engine(0)
    one.epi_set_dataset(data, nA, nU)
    one.epi_set_folds(fold, k)
    acc, rm = one.epi_scan_pairs(hpgv.EPI_TESTING)
    pairs = [(i, j) for i in range(v) for j in range(i + 1, v)]
    for f in range(k):
        a = np.where(np.isnan(acc[f]), -np.inf, acc[f])
        order = sorted(range(len(pairs)), key=lambda p: (-a[p], pairs[p]))[:n]
        assert [tuple(c) for c in res["combs"][f].tolist()] == [pairs[p] for p in order]
        assert np.array_equal(res["accuracy"][f], acc[f][order]) and np.array_equal(res["risky"][f][:, 0], rm[f][order].astype(np.uint32))
    one.close()
    g.close()


@pytest.mark.parametrize("members,variants,samples,extra", [
    (1, 20011, 3001, []),            # one member, communicator created (ncclCommInitAll, n = 1), member 0 scans in place
    (1, 20011, 3001, ["self"]),      # the same through ncclSend / ncclRecv to itself: RCCL moves every result byte
    (2, 20011, 3001, []),            # [0, 0]: shard split and placement; the second context hands over by a device-local copy
    (3, 7, 30, []),                  # ragged: 7 variants over 3 members
    (4, 3, 30, []),                  # a member with an empty shard
    (2, 150001, 1201, ["self"]),
])
def test_c_host_drives_the_group_scan(driver, members, variants, samples, extra):
    r = subprocess.run([driver, str(members), str(variants), str(samples)] + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "rccl_ranks=1" in r.stdout and "group scan ok" in r.stdout
    assert r.stdout.count("bit-identical") == 4          # chi-square, Fisher, TDT, stats


def test_group_scan_against_the_oracle():
    """The gathered result arrays of a [0, 0, 0] group against the CPU oracle (not only against another HIP scan)."""
    N, V = 1501, 9001
    cond = (np.arange(N) % 2).astype(np.uint8)
    g = hpgv.Engine([0, 0, 0])
    _, _, pitch = g.set_cohort(cond)
    assert g.group_comm_init() == 1
    shards, mems = [], []
    for k in range(3):
        lo, hi = g.group_shard(V, k)
        assert (lo, hi) == ((V * k) // 3, (V * (k + 1)) // 3)
        m = g.member(k)
        p = m.alloc(max(hi - lo, 1) * pitch)
        m.synth(hpgv.LAYOUT_ASSOC, lo, hi - lo, p)
        m.sync()
        shards.append(p)
        mems.append(m)
    m0 = mems[0]
    d_counts, d_odds, d_chisq, d_p = m0.alloc(16 * V), m0.alloc(8 * V), m0.alloc(8 * V), m0.alloc(8 * V)
    g.group_assoc(hpgv.TASK_CHISQ, shards, V, d_counts, d_odds, d_chisq, d_p)
    g.group_sync()
    counts = m0.d2h(d_counts, (V, 4), np.int32)
    odds, chisq, p = (m0.d2h(d, (V,), np.float64) for d in (d_odds, d_chisq, d_p))
    rows = orc.synth_matrix(0, V, N, N)
    A1, A2, U1, U2 = orc.assoc_counts(rows, cond)
    assert np.array_equal(counts, np.stack([A1, A2, U1, U2], 1))
    for got, exp in zip((odds, chisq, p), orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)):
        with np.errstate(invalid="ignore"):
            assert np.all((np.abs(got - exp) <= TOL * np.maximum(1, np.abs(exp))) | (np.isnan(got) & np.isnan(exp)))
    for m in mems:
        m.close()
    g.close()


def test_group_calls_refuse_an_ordinary_context():
    e = hpgv.Engine(0)
    assert e.L.hpgv_group_comm_init(e.h) == 1                  # HPGV_ERR_INVALID
    assert e.L.hpgv_group_comm_ranks(e.h) == 0
    e.close()
