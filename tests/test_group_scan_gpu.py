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
