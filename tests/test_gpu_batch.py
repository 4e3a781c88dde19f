"""The per-batch host entry points as ONE fused kernel (hpgv_batch_kernels.h): same results as the oracle and as the
copy + layout + scan + statistics path (option batch_fused = 0), with the batch in pageable memory (copied up) and in
page-locked memory (read in place over the bus), for rows of every alignment."""
import numpy as np
import pytest

from helpers import assert_close, check_assoc, hpgv, make_families, oracle_assoc, random_codes
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def _same(a, b, exact=True):
    for k in a:
        if a[k] is None:
            assert b[k] is None
        elif a[k].dtype.kind == "f" and exact:
            assert np.array_equal(a[k], b[k], equal_nan=True), k        # same kernels' arithmetic: identical doubles
        elif a[k].dtype.kind == "f":
            assert_close(a[k], b[k], k)
        else:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("n_samples", [1, 15, 16, 17, 333, 4097, 10007])
def test_assoc_batch_fused_equals_unfused_and_oracle(n_samples):
    rng = np.random.default_rng(n_samples)
    e = hpgv.Engine(0)
    cond = rng.integers(0, 3, n_samples).astype(np.uint8)
    e.set_cohort(cond)
    lf = orc.logfact(max(n_samples * 10, 16))
    e.set_logfact(lf)
    nv = 203
    gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)       # half-missing calls included: the kernel drops them
    is_x = (rng.random(nv) < 0.3).astype(np.uint8)
    strict = np.where(((gt & 0xF) == 0xF) | ((gt >> 4) == 0xF), 0xFF, gt).astype(np.uint8)
    # page-locked copy with an odd pitch and an odd start: rows at every alignment
    pitch = n_samples + 5
    pin = e.host_array((nv * pitch + 3 + nv,))
    pin[:] = 0xEE
    view = pin[3: 3 + nv * pitch].reshape(nv, pitch)
    view[:, :n_samples] = gt
    pin_x = pin[3 + nv * pitch:]
    pin_x[:] = is_x
    for task in (hpgv.TASK_CHISQ, hpgv.TASK_FISHER):
        exp = oracle_assoc(task, strict, cond, is_x, lf)
        e.set_option("batch_fused", 1)
        a = e.assoc(task, gt, is_x)                                          # pageable: copied up, then the fused kernel
        b = e.assoc_view(task, view, n_samples, pin_x)                       # page-locked: read in place
        e.set_option("batch_fused", 0)
        c = e.assoc(task, gt, is_x)                                          # the kernel chain
        check_assoc(a, exp, task)
        check_assoc(c, exp, task)
        _same(a, b)
        # the chain's Fisher pass puts 16 lanes on a table, the per-batch kernel 64: other summation order, same 1e-10 bar
        _same(a, c, exact=(task == hpgv.TASK_CHISQ))
    e.close()


@pytest.mark.parametrize("n_samples,max_children", [(3, 1), (50, 1), (999, 3), (6001, 4)])
def test_tdt_batch_fused(n_samples, max_children):
    rng = np.random.default_rng(7 * n_samples)
    e = hpgv.Engine(0)
    fam = make_families(rng, n_samples, n_samples // 3, max_children=max_children, p_absent=0.05)
    e.set_families(n_samples, *fam)
    nv = 157
    gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)
    is_x = (rng.random(nv) < 0.4).astype(np.uint8)
    strict = np.where(((gt & 0xF) == 0xF) | ((gt >> 4) == 0xF), 0xFF, gt).astype(np.uint8)
    t1, t2 = orc.tdt_counts(strict, *fam, chrom_is_x=is_x)
    odds, chisq, p = orc.tdt_stats(t1, t2)
    pin = e.host_array((nv * n_samples + 1 + nv,))
    view = pin[1: 1 + nv * n_samples].reshape(nv, n_samples)
    view[:] = gt
    pin_x = pin[1 + nv * n_samples:]
    pin_x[:] = is_x
    e.set_option("batch_fused", 1)
    a = e.tdt(gt, is_x)
    b = e.tdt_view(view, n_samples, pin_x)
    e.set_option("batch_fused", 0)
    c = e.tdt(gt, is_x)
    assert np.array_equal(a["t1"], t1) and np.array_equal(a["t2"], t2)
    assert_close(a["odds"], odds, "odds"); assert_close(a["chisq"], chisq, "chisq"); assert_close(a["p"], p, "p")
    _same(a, b)
    _same(a, c)
    e.close()


@pytest.mark.parametrize("n_samples", [2, 31, 2500])
def test_stats_batch_fused(n_samples):
    rng = np.random.default_rng(n_samples)
    e = hpgv.Engine(0)
    e.set_stats_cohort(n_samples)
    nv = 120
    gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)
    e.set_option("batch_fused", 1)
    a = e.stats_ex(gt, multi_cap=nv)
    e.set_option("batch_fused", 0)
    c = e.stats_ex(gt, multi_cap=nv)
    for k in ("counts8", "hwe_chi2", "hwe_p", "multi_idx", "multi_table"):
        assert np.array_equal(a[k], c[k], equal_nan=True), k
    assert a["n_multi"] == c["n_multi"] and a["n_multi"] > 0
    for v in range(0, nv, 7):
        vs = orc.variant_stats(gt[v], 2)
        assert list(a["counts8"][v, :4]) == list(vs.genotypes_count)[:4]
        assert_close([a["hwe_chi2"][v]], [vs.hw_chi2], "hwe")
    e.close()


def test_long_rows_take_the_kernel_chain():
    """A row that does not fit the fused kernel's LDS window falls back to copy + layout + scan, same results."""
    n_samples = 200_000
    rng = np.random.default_rng(5)
    e = hpgv.Engine(0)
    cond = (np.arange(n_samples) % 2).astype(np.uint8)
    e.set_cohort(cond)
    gt = random_codes(rng, 9, n_samples, quirks=False)
    res = e.assoc(hpgv.TASK_CHISQ, gt)
    check_assoc(res, oracle_assoc(hpgv.TASK_CHISQ, gt, cond), hpgv.TASK_CHISQ)
    e.close()


@pytest.mark.parametrize("n_samples", [5, 333, 4100])
def test_sample_stats_groups_mendel_in_one_pass(n_samples):
    """hpgv_stats_ex with per-sample counters, hpgv_stats_groups and hpgv_mendel through k_stats_all (one pass over the batch)
    against the kernel chains and the oracle."""
    rng = np.random.default_rng(n_samples)
    nv = 211
    gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)
    is_x = (rng.random(nv) < 0.3).astype(np.uint8)
    n_trios = n_samples // 4
    cols = rng.permutation(n_samples)
    f, m, c = cols[:n_trios], cols[n_trios: 2 * n_trios], cols[2 * n_trios: 3 * n_trios]
    sex = rng.integers(0, 2, n_trios).astype(np.uint8)
    groups = rng.integers(-1, 3, n_samples).astype(np.int32)
    out = {}
    for fused in (1, 0):
        e = hpgv.Engine(0)
        e.set_option("batch_fused", fused)
        e.set_stats_cohort(n_samples)
        e.set_pedigree(n_samples, f, m, c, sex)
        e.set_stats_groups(groups, 3)
        miss = np.zeros(n_samples, np.int32)
        a = e.stats_ex(gt, sample_missing=miss, multi_cap=nv)
        g = e.stats_groups(gt, 3)
        cerr = np.zeros(n_trios, np.int32)
        merr = e.mendel(gt, is_x, cerr)
        out[fused] = (a, miss, g, merr, cerr)
        e.close()
    a1, miss1, g1, merr1, cerr1 = out[1]
    a0, miss0, g0, merr0, cerr0 = out[0]
    for k in ("counts8", "hwe_chi2", "hwe_p", "multi_idx", "multi_table"):
        assert np.array_equal(a1[k], a0[k], equal_nan=True), k
    assert np.array_equal(miss1, miss0) and np.array_equal(miss1, orc.sample_missing(gt))
    for k in ("counts8", "hwe_chi2", "hwe_p"):
        assert np.array_equal(g1[k], g0[k], equal_nan=True), k
    exp_err, exp_trio = orc.mendel_counts(gt, f, m, c, sex, is_x)
    assert np.array_equal(merr1, exp_err) and np.array_equal(merr0, exp_err)
    assert np.array_equal(cerr1, exp_trio) and np.array_equal(cerr0, exp_trio)
    for gi in range(3):
        sel = groups == gi
        for v in range(0, nv, 17):
            vs = orc.variant_stats(np.ascontiguousarray(gt[v][sel]), 2)
            assert list(g1["counts8"][gi, v, :4]) == list(vs.genotypes_count)[:4]
