"""GPU parity tests: the HIP path, called through the C ABI (include/hpgv.h),
against the CPU oracle on the same inputs.  Integer tallies bit-exact, FP64
statistics within 1e-10 (NaN == NaN).  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest

from helpers import (QUIRK_GTS, assert_close, check_assoc, hpgv, make_families, oracle_assoc,
                     random_codes)
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = hpgv.Engine(0)      # raises when libhpgv.so or the device is missing: no fallback
    yield e
    e.close()


def fresh():
    return hpgv.Engine(0)


# ---------------------------------------------------------------- assoc ----

@pytest.mark.parametrize("n_samples", [1, 2, 15, 16, 17, 31, 33, 100, 1000, 4099, 10007])
def test_assoc_chisq_ragged_shapes(n_samples):
    rng = np.random.default_rng(n_samples)
    e = fresh()
    cond = rng.choice([0, 1, 2], size=n_samples, p=[0.45, 0.45, 0.1]).astype(np.uint8)
    e.set_cohort(cond)
    for nv in (1, 3, 64, 257):
        gt = random_codes(rng, nv, n_samples)
        is_x = (rng.random(nv) < 0.3).astype(np.uint8)
        res = e.assoc(hpgv.TASK_CHISQ, gt, is_x)
        check_assoc(res, oracle_assoc(orc.TASK_CHISQ, gt, cond, is_x), hpgv.TASK_CHISQ)
        res = e.assoc(hpgv.TASK_CHISQ, gt, None)
        check_assoc(res, oracle_assoc(orc.TASK_CHISQ, gt, cond, None), hpgv.TASK_CHISQ)
    e.close()


def test_assoc_degenerate_cohorts():
    rng = np.random.default_rng(7)
    for cond in ([1] * 40, [0] * 40, [2] * 40, [1] + [0] * 39, []):
        e = fresh()
        cond = np.array(cond, dtype=np.uint8)
        e.set_cohort(cond)
        gt = random_codes(rng, 10, len(cond)) if len(cond) else np.zeros((10, 0), np.uint8)
        if len(cond) == 0:
            gt = np.zeros((10, 16), np.uint8)[:, :0]
            gt = np.ascontiguousarray(np.zeros((10, 1), np.uint8))   # pitch 1 >= n_samples 0
            exp = oracle_assoc(orc.TASK_CHISQ, np.zeros((10, 0), np.uint8), cond)
        else:
            exp = oracle_assoc(orc.TASK_CHISQ, gt, cond)
        res = e.assoc(hpgv.TASK_CHISQ, gt)
        check_assoc(res, exp, hpgv.TASK_CHISQ)   # zero margins: chi2 NaN, p NaN, OR NaN
        e.close()


def test_assoc_empty_batch(eng):
    eng.set_cohort(np.array([0, 1] * 8, dtype=np.uint8))
    res = eng.assoc(hpgv.TASK_CHISQ, np.zeros((0, 16), np.uint8))
    assert len(res["A1"]) == 0


def test_assoc_all_missing_and_all_one_class():
    e = fresh()
    cond = np.array([0, 1] * 500, dtype=np.uint8)
    e.set_cohort(cond)
    for code in (0xFF, 0x00, 0x11, 0x01, 0x10, 0xEE):
        gt = np.full((5, 1000), code, dtype=np.uint8)
        for is_x in (None, np.ones(5, np.uint8)):
            check_assoc(e.assoc(hpgv.TASK_CHISQ, gt, is_x), oracle_assoc(orc.TASK_CHISQ, gt, cond, is_x),
                        hpgv.TASK_CHISQ)
    e.close()


def test_assoc_half_missing_is_dropped():
    # staging must canonicalise "./1"-style bytes (0xF1, 0x1F) to missing: assoc.c:53
    e = fresh()
    cond = np.array([0, 1] * 50, dtype=np.uint8)
    e.set_cohort(cond)
    rng = np.random.default_rng(3)
    gt = random_codes(rng, 20, 100, strict=False)
    assert ((gt & 0xF) == 0xF).any()
    strict = gt.copy()
    strict[((gt & 0xF) == 0xF) | ((gt >> 4) == 0xF)] = 0xFF
    check_assoc(e.assoc(hpgv.TASK_CHISQ, gt), oracle_assoc(orc.TASK_CHISQ, strict, cond), hpgv.TASK_CHISQ)
    e.close()


@pytest.mark.parametrize("n_samples,nv", [(40, 50), (600, 200), (3001, 64)])
def test_assoc_fisher(n_samples, nv):
    rng = np.random.default_rng(100 + n_samples)
    e = fresh()
    cond = rng.choice([0, 1], size=n_samples).astype(np.uint8)
    e.set_cohort(cond)
    lf = orc.logfact(n_samples * 10)          # assoc_runner.c:164-166
    e.set_logfact(lf)
    gt = random_codes(rng, nv, n_samples, quirks=False)
    # push some variants towards strong association so tiny p-values are covered
    gt[: nv // 4, cond == 1] = 0x11
    res = e.assoc(hpgv.TASK_FISHER, gt)
    check_assoc(res, oracle_assoc(orc.TASK_FISHER, gt, cond, None, lf), hpgv.TASK_FISHER)
    e.close()


def test_fisher_needs_table():
    e = fresh()
    e.set_cohort(np.array([0, 1] * 8, dtype=np.uint8))
    with pytest.raises(hpgv.HpgvError):
        e.assoc(hpgv.TASK_FISHER, np.zeros((2, 16), np.uint8))
    e.set_logfact(orc.logfact(8))              # too short for 32 alleles
    with pytest.raises(hpgv.HpgvError):
        e.assoc(hpgv.TASK_FISHER, np.zeros((2, 16), np.uint8))
    e.close()


def test_concurrent_batches_like_the_runner():
    # assoc_runner.c:106-207: num_threads workers call assoc_test concurrently
    import threading
    e = fresh()
    n = 2000
    cond = (np.arange(n) % 2).astype(np.uint8)
    e.set_cohort(cond)
    out, errs = {}, []

    def work(t):
        try:
            rng = np.random.default_rng(t)
            for _ in range(5):
                gt = random_codes(rng, 200, n)
                check_assoc(e.assoc(hpgv.TASK_CHISQ, gt), oracle_assoc(orc.TASK_CHISQ, gt, cond), hpgv.TASK_CHISQ)
            out[t] = True
        except Exception as ex:       # pragma: no cover
            errs.append(ex)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    assert len(out) == 4
    e.close()


# ------------------------------------------------------------- synthetic ----

def test_synth_matches_oracle_bit_for_bit(eng):
    nv, ns, pitch = 300, 1234, 1248
    d = eng.alloc(nv * pitch)
    eng.synth_raw(5000, nv, ns, pitch, d)
    got = eng.d2h(d, (nv, pitch), np.uint8)
    exp = orc.synth_matrix(5000, nv, ns, pitch)
    assert np.array_equal(got, exp)
    eng.free(d)


def test_device_resident_assoc_on_synthetic_cohort():
    e = fresh()
    ns, nv = 10000, 4096
    cond = (np.arange(ns) % 2).astype(np.uint8)     # odd samples are cases
    nA, nU, pitch = e.set_cohort(cond)
    assert (nA, nU) == (5000, 5000) and pitch == 10016
    d_gt = e.alloc(nv * pitch)
    d_counts = e.alloc(nv * 16)
    d_out = e.alloc(nv * 24)
    e.synth(hpgv.LAYOUT_ASSOC, 77, nv, d_gt)
    e.assoc_scan(d_gt, nv, d_counts)
    base = d_out.value
    e.assoc_chisq(d_counts, nv, base, base + 8 * nv, base + 16 * nv)
    e.sync()
    counts = e.d2h(d_counts, (nv, 4), np.int32)
    stats = e.d2h(d_out, (3, nv), np.float64)
    gt = orc.synth_matrix(77, nv, ns, ns)
    exp = oracle_assoc(orc.TASK_CHISQ, gt, cond)
    res = dict(A1=counts[:, 0], A2=counts[:, 1], U1=counts[:, 2], U2=counts[:, 3],
               odds=stats[0], chisq=stats[1], p=stats[2])
    check_assoc(res, exp, hpgv.TASK_CHISQ)
    # the layout kernel from VCF order gives the same matrix as the direct generator
    d_raw = e.alloc(nv * 10000)
    e.synth_raw(77, nv, ns, 10000, d_raw)
    d_lay = e.alloc(nv * pitch)
    e.layout(hpgv.LAYOUT_ASSOC, d_raw, 10000, nv, d_lay)
    e.sync()
    assert np.array_equal(e.d2h(d_lay, (nv, pitch), np.uint8), e.d2h(d_gt, (nv, pitch), np.uint8))
    e.close()


@pytest.mark.parametrize("opt", [("nontemporal", 0), ("variants_per_wave", 1), ("variants_per_wave", 7),
                                 ("row_align", 128), ("row_align", 256), ("pipeline", 0), ("scan_unroll", 8),
                                 ("scan_unroll", 10), ("scan_unroll", 16), ("persistent", 1), ("pipe_waves", 6)])
def test_options_do_not_change_results(opt):
    from helpers import set_or_skip
    e = fresh()
    set_or_skip(e, *opt)
    if opt[0] in ("scan_unroll", "persistent"):
        set_or_skip(e, "pipeline", 0)      # these knobs belong to the non-pipelined kernel
    rng = np.random.default_rng(11)
    cond = rng.choice([0, 1, 2], size=3000).astype(np.uint8)
    e.set_cohort(cond)
    gt = random_codes(rng, 333, 3000)
    is_x = (rng.random(333) < 0.3).astype(np.uint8)
    check_assoc(e.assoc(hpgv.TASK_CHISQ, gt, is_x), oracle_assoc(orc.TASK_CHISQ, gt, cond, is_x), hpgv.TASK_CHISQ)
    if opt[0] == "pipeline":                 # also with the other tile shapes of the pipelined kernel
        for u in (8, 10):
            e.set_option("pipeline", 1); e.set_option("scan_unroll", u)
            check_assoc(e.assoc(hpgv.TASK_CHISQ, gt, is_x), oracle_assoc(orc.TASK_CHISQ, gt, cond, is_x), hpgv.TASK_CHISQ)
    e.close()


# ------------------------------------------------------------------ TDT ----

def _tdt_check(e, gt, fam, is_x=None):
    res = e.tdt(gt, is_x)
    t1, t2 = orc.tdt_counts(gt, *fam, chrom_is_x=is_x)
    assert np.array_equal(res["t1"], t1) and np.array_equal(res["t2"], t2), "TDT tallies differ"
    odds, chisq, p = orc.tdt_stats(t1, t2)
    assert_close(res["odds"], odds, "tdt odds")
    assert_close(res["chisq"], chisq, "tdt chisq")
    assert_close(res["p"], p, "tdt p")


def test_tdt_reference_kats(goldens):
    # test/test_tdt_runner.c:93-433 through the HIP path
    for case in goldens["kats"]["tdt"]:
        samples, fcol, mcol, coff, ccol, csex = [], [], [], [0], [], []
        for fam in case["families"]:
            b = len(samples)
            samples += [fam["father"], fam["mother"], fam["child"]]
            fcol.append(b); mcol.append(b + 1)
            if fam["child_affected"]:
                ccol.append(b + 2); csex.append(hpgv.SEX_MALE)
            coff.append(len(ccol))
        gt = orc.encode_matrix([samples])
        e = fresh()
        e.set_families(len(samples), fcol, mcol, coff, ccol, csex)
        res = e.tdt(gt)
        assert (res["t1"][0], res["t2"][0]) == (case["t1"], case["t2"]), case["name"]
        e.close()


def test_tdt_all_genotype_triples_exhaustive():
    # every (father, mother, child) combination of the quirk genotypes, as trios
    codes = sorted(set(orc.encode_sample(s) for s in QUIRK_GTS))
    trip = [(f, m, c) for f in codes for m in codes for c in codes]
    n = len(trip)
    gt = np.zeros((1, 3 * n), np.uint8)
    gt[0, 0::3] = [t[0] for t in trip]
    gt[0, 1::3] = [t[1] for t in trip]
    gt[0, 2::3] = [t[2] for t in trip]
    # one variant per trio so that every combination is checked on its own
    gts = np.full((n, 3 * n), 0xFF, np.uint8)
    for i in range(n):
        gts[i, 3 * i: 3 * i + 3] = gt[0, 3 * i: 3 * i + 3]
    fam = (np.arange(n) * 3, np.arange(n) * 3 + 1, np.arange(n + 1), np.arange(n) * 3 + 2,
           (np.arange(n) % 2).astype(np.uint8))
    fam = tuple(np.asarray(a) for a in fam)
    e = fresh()
    e.set_families(3 * n, *fam)
    for is_x in (None, np.ones(n, np.uint8)):
        _tdt_check(e, gts, fam, is_x)
    e.close()


@pytest.mark.parametrize("n_fam,max_children", [(1, 1), (17, 1), (700, 1), (300, 4), (5000, 1), (40, 30)])
def test_tdt_random_pedigrees(n_fam, max_children):
    rng = np.random.default_rng(n_fam * 31 + max_children)
    n_samples = n_fam * (2 + max_children) + 5
    fam = make_families(rng, n_samples, n_fam, max_children, p_absent=0.05 if n_fam > 10 else 0.0)
    e = fresh()
    n_fast, n_slow, pitch = e.set_families(n_samples, *fam)
    if max_children > 1 and n_fam >= 40:
        assert n_slow > 0
    nv = 97
    gt = random_codes(rng, nv, n_samples)
    is_x = (rng.random(nv) < 0.4).astype(np.uint8)
    _tdt_check(e, gt, fam, None)
    _tdt_check(e, gt, fam, is_x)
    gt2 = random_codes(rng, nv, n_samples, quirks=False)   # realistic mix: many counted trios
    _tdt_check(e, gt2, fam, is_x)
    e.close()


def test_tdt_synthetic_trio_cohort():
    # trio k = columns (3k, 3k+1, 3k+2), child sex alternating
    e = fresh()
    n_tr, nv = 5000, 512
    k = np.arange(n_tr)
    fam = (3 * k, 3 * k + 1, np.arange(n_tr + 1), 3 * k + 2, (k % 2).astype(np.uint8))
    n_fast, n_slow, pitch = e.set_families(3 * n_tr, *fam)
    assert (n_fast, n_slow) == (n_tr, 0)
    d_gt, d_tu = e.alloc(nv * pitch), e.alloc(nv * 8)
    e.synth(hpgv.LAYOUT_TDT, 1000, nv, d_gt)
    e.tdt_scan(d_gt, nv, d_tu)
    e.sync()
    tu = e.d2h(d_tu, (nv, 2), np.int32)
    gt = orc.synth_matrix(1000, nv, 3 * n_tr, 3 * n_tr)
    t1, t2 = orc.tdt_counts(gt, *fam)
    assert np.array_equal(tu[:, 0], t1) and np.array_equal(tu[:, 1], t2)
    assert t1.sum() > 0 and t2.sum() > 0
    e.close()


# ---------------------------------------------------------------- stats ----

@pytest.mark.parametrize("n_samples", [1, 16, 100, 2504, 10001])
def test_variant_stats_and_hwe(n_samples):
    rng = np.random.default_rng(n_samples + 5)
    e = fresh()
    e.set_stats_cohort(n_samples)
    nv = 130
    gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)
    gt[: nv // 2] = random_codes(rng, nv // 2, n_samples, quirks=False)
    res = e.stats(gt)
    for i in range(nv):
        vs = orc.variant_stats(gt[i], 2)
        g = list(vs.genotypes_count)[:4]
        c8 = res["counts8"][i]
        assert list(c8[:4]) == g, (i, list(c8), g)
        assert c8[4] == vs.missing_genotypes and c8[5] == vs.missing_alleles
        assert c8[6] == vs.alleles_count[0] and c8[7] == vs.alleles_count[1]
        assert_close([res["hwe_chi2"][i]], [vs.hw_chi2], "hwe chi2")
        assert_close([res["hwe_p"][i]], [vs.hw_p], "hwe p")
    e.close()


def test_sample_missing_and_multiallelic_tables():
    rng = np.random.default_rng(77)
    for n_samples, nv in ((1, 5), (1000, 300), (2049, 129), (5000, 400)):
        e = fresh()
        e.set_stats_cohort(n_samples)
        gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)
        gt[::3] = random_codes(rng, len(gt[::3]), n_samples, quirks=False)      # biallelic rows in between
        acc = np.full(n_samples, 7, np.int32)                                    # accumulates INTO the array
        res = e.stats_ex(gt, sample_missing=acc, multi_cap=nv)
        assert np.array_equal(acc - 7, orc.sample_missing(gt))
        multi = [i for i in range(nv) if ((gt[i] != 0x00) & (gt[i] != 0x01) & (gt[i] != 0x10) & (gt[i] != 0x11) &
                                          ((gt[i] >> 4) != 0xF) & ((gt[i] & 0xF) != 0xF)).any()]
        assert res["n_multi"] == len(multi) and list(res["multi_idx"]) == multi
        for k, i in enumerate(multi):
            assert np.array_equal(res["multi_table"][k], np.bincount(gt[i], minlength=256))
        # a too small capacity reports the true count and fills what fits
        small = e.stats_ex(gt, multi_cap=1)
        assert small["n_multi"] == len(multi) and len(small["multi_idx"]) == min(1, len(multi))
        e.close()


def test_count_derived_filters():
    rng = np.random.default_rng(31)
    n_samples, nv = 800, 500
    e = fresh()
    pitch = e.set_stats_cohort(n_samples)
    gt = random_codes(rng, nv, n_samples, quirks=False, strict=False, p_missing=0.05)
    gt[::5, : n_samples // 3] = 0xFF                         # some variants with a lot of missing calls
    gt[1::7] = 0x00                                          # monomorphic variants (maf 0)
    d_raw, d_lay, d_c8, d_keep = e.alloc(nv * n_samples), e.alloc(nv * pitch), e.alloc(nv * 32), e.alloc(nv)
    e.h2d(d_raw, gt)
    e.layout(hpgv.LAYOUT_STATS, d_raw, n_samples, nv, d_lay)
    e.stats_scan(d_lay, nv, d_c8)
    maf = np.zeros(nv); miss = np.zeros(nv)
    for i in range(nv):
        vs = orc.variant_stats(gt[i], 2)
        a0, a1 = vs.alleles_count[0], vs.alleles_count[1]
        maf[i] = min(a0, a1) / (a0 + a1) if a0 + a1 else 0.0
        miss[i] = vs.missing_genotypes / n_samples
    for min_maf, max_maf, max_missing in ((0.05, -1, -1), (-1, 0.2, -1), (-1, -1, 0.1), (0.01, 0.45, 0.3), (-1, -1, -1)):
        e.stats_filter(d_c8, nv, d_keep, min_maf, max_maf, max_missing)
        e.sync()
        keep = e.d2h(d_keep, (nv,), np.uint8)
        exp = np.ones(nv, bool)
        if min_maf >= 0: exp &= maf >= min_maf
        if max_maf >= 0: exp &= maf <= max_maf
        if max_missing >= 0: exp &= miss <= max_missing
        assert np.array_equal(keep.astype(bool), exp)
        assert 0 < exp.sum() <= nv
    e.close()


# ------------------------------------------------------- error behaviour ----

def test_state_and_argument_errors():
    e = fresh()
    with pytest.raises(hpgv.HpgvError):
        e.assoc(hpgv.TASK_CHISQ, np.zeros((1, 16), np.uint8))     # no cohort yet
    e.set_cohort(np.array([0, 1] * 50, dtype=np.uint8))
    with pytest.raises(hpgv.HpgvError):
        e.assoc(hpgv.TASK_CHISQ, np.zeros((1, 16), np.uint8))     # pitch < n_samples
    with pytest.raises(hpgv.HpgvError):
        e.assoc(3, np.zeros((1, 100), np.uint8))                   # bad task
    with pytest.raises(hpgv.HpgvError):
        e.tdt(np.zeros((1, 100), np.uint8))                        # no families
    with pytest.raises(hpgv.HpgvError):
        e.set_families(10, [11], [1], [0, 1], [2], [0])            # column out of range
    e.close()


def test_maximum_row_length():
    # rows up to ~2M samples are supported (16-bit per-lane partial sums); beyond that the engine refuses
    rng = np.random.default_rng(5)
    n_samples = 2_000_000
    cond = rng.choice([0, 1, 2], size=n_samples, p=[0.5, 0.45, 0.05]).astype(np.uint8)
    nv = 24
    gt = random_codes(rng, nv, n_samples, quirks=False)
    gt[0] = 0x00; gt[1] = 0x11; gt[2] = 0xFF            # extreme tallies: every sample the same
    is_x = np.zeros(nv, np.uint8); is_x[1::2] = 1
    from helpers import shipped
    for opts in ({}, {"pipeline": 0}, {"pipeline": 0, "scan_unroll": 16}):
        if opts and not shipped("pipeline", [0]):
            continue                                                  # the unpipelined scan: an ablation build's form
        e = fresh()
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_cohort(cond)
        check_assoc(e.assoc(hpgv.TASK_CHISQ, gt, is_x), oracle_assoc(orc.TASK_CHISQ, gt, cond, is_x), hpgv.TASK_CHISQ)
        e.close()
    e = fresh()
    e.set_stats_cohort(n_samples)
    res = e.stats(gt[:6])
    for i in range(6):
        vs = orc.variant_stats(gt[i], 2)
        assert list(res["counts8"][i][:4]) == list(vs.genotypes_count)[:4] and res["counts8"][i][4] == vs.missing_genotypes
    with pytest.raises(hpgv.HpgvError):
        e.set_cohort(np.zeros(2_200_000, np.uint8))      # too long a row
    with pytest.raises(hpgv.HpgvError):
        e.set_stats_cohort(2_200_000)
    e.close()
    # a pedigree as wide as a 600k-sample cohort (200k trios)
    n_tr = 200_000
    k = np.arange(n_tr)
    fam = (3 * k, 3 * k + 1, np.arange(n_tr + 1), 3 * k + 2, (k % 2).astype(np.uint8))
    e = fresh()
    e.set_families(3 * n_tr, *fam)
    gt = random_codes(rng, 8, 3 * n_tr, quirks=False)
    _tdt_check(e, gt, fam, np.array([0, 1] * 4, np.uint8))
    e.close()


def test_stats_per_phenotype_group():
    rng = np.random.default_rng(41)
    n_samples, nv, n_groups = 1234, 150, 3
    group = rng.integers(-1, n_groups, size=n_samples).astype(np.int32)     # -1: in no group
    e = fresh()
    pitch, sizes = e.set_stats_groups(group, n_groups)
    assert [int(x) for x in sizes] == [int((group == k).sum()) for k in range(n_groups)]
    gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)
    d_raw, d_lay, d_c8, d_hw = e.alloc(nv * n_samples), e.alloc(nv * pitch), e.alloc(nv * 32), e.alloc(nv * 16)
    e.h2d(d_raw, gt)
    e.layout(hpgv.LAYOUT_STATS_GROUPS, d_raw, n_samples, nv, d_lay)
    for k in range(n_groups):
        e.stats_scan_group(d_lay, nv, k, d_c8)
        e.stats_hwe(d_c8, nv, d_hw, d_hw.value + 8 * nv)
        e.sync()
        c8 = e.d2h(d_c8, (nv, 8), np.int32)
        hw = e.d2h(d_hw, (2, nv), np.float64)
        sub = gt[:, group == k]
        for i in range(nv):
            vs = orc.variant_stats(np.ascontiguousarray(sub[i]), 2)
            assert list(c8[i][:4]) == list(vs.genotypes_count)[:4]
            assert (c8[i][4], c8[i][5], c8[i][6], c8[i][7]) == (vs.missing_genotypes, vs.missing_alleles,
                                                                 vs.alleles_count[0], vs.alleles_count[1])
            assert_close([hw[0][i]], [vs.hw_chi2], "hwe chi2"); assert_close([hw[1][i]], [vs.hw_p], "hwe p")
    with pytest.raises(hpgv.HpgvError):
        e.stats_scan_group(d_lay, nv, n_groups, d_c8)
    # the host-batch entry point gives the same counters for all groups in one call
    res = e.stats_groups(gt, n_groups)
    for k in range(n_groups):
        sub = gt[:, group == k]
        for i in range(0, nv, 7):
            vs = orc.variant_stats(np.ascontiguousarray(sub[i]), 2)
            c = res["counts8"][k][i]
            assert list(c[:4]) == list(vs.genotypes_count)[:4] and (c[4], c[5]) == (vs.missing_genotypes, vs.missing_alleles)
            assert_close([res["hwe_chi2"][k][i]], [vs.hw_chi2], "hwe chi2"); assert_close([res["hwe_p"][k][i]], [vs.hw_p], "hwe p")
    e.close()


@pytest.mark.parametrize("n_trios", [1, 17, 1000, 2100])
def test_mendelian_errors_per_variant_and_per_child(n_trios):
    rng = np.random.default_rng(n_trios)
    n_samples = 3 * n_trios + 4
    cols = rng.permutation(n_samples)
    f, m, c = cols[:n_trios], cols[n_trios: 2 * n_trios], cols[2 * n_trios: 3 * n_trios]
    sex = rng.integers(0, 2, size=n_trios).astype(np.uint8)
    nv = 300
    gt = random_codes(rng, nv, n_samples, quirks=True, strict=False)
    gt[::2] = random_codes(rng, len(gt[::2]), n_samples, quirks=False)
    is_x = (rng.random(nv) < 0.4).astype(np.uint8)
    e = fresh()
    pitch = e.set_pedigree(n_samples, f, m, c, sex)
    d_raw, d_lay = e.alloc(nv * n_samples), e.alloc(nv * pitch)
    d_err, d_child, d_isx = e.alloc(nv * 4), e.alloc(n_trios * 4), e.alloc(nv)
    e.h2d(d_raw, gt); e.h2d(d_isx, is_x); e.h2d(d_child, np.zeros(n_trios, np.int32))
    e.layout(hpgv.LAYOUT_MENDEL, d_raw, n_samples, nv, d_lay)
    for x, dx in ((is_x, d_isx), (None, None)):
        e.h2d(d_child, np.zeros(n_trios, np.int32))
        e.mendel_scan(d_lay, nv, d_err, dx)
        e.mendel_children(d_lay, nv, d_child, dx)
        e.sync()
        exp_err, exp_trio = orc.mendel_counts(gt, f, m, c, sex, x)
        assert np.array_equal(e.d2h(d_err, (nv,), np.int32), exp_err)
        assert np.array_equal(e.d2h(d_child, (n_trios,), np.int32), exp_trio)
        assert exp_err.sum() > 0
    e.close()


def test_epistasis_dataset_reference_kat_and_random(goldens):
    # test/test_epistasis_dataset.c:87-154 through the HIP layout (cases first, codes 0/1/2/255)
    k = goldens["kats"]["epistasis_dataset"]
    n = k["num_samples"]
    rows = []
    for r in range(3):
        row = [k["possible_gts"][(i + r) % 3] for i in range(n)]
        if r == 0:
            row[-1] = "./."
        rows.append(row)
    gt = orc.encode_matrix(rows, strict=False)
    pheno = (np.arange(n) % 2).astype(np.uint8)                  # phenotypes[i] = i % 2, 1 = affected
    e = fresh()
    e.set_cohort(pheno)
    out = e.epi_dataset(gt)
    for r, exp in k["expected"].items():
        for dest, code in exp.items():
            assert out[int(r), int(dest)] == code, (r, dest)
    # destination rule of group_individuals_by_phenotype (test_destination): cases at (i+1)/2-1, controls at nA+(i+1)/2
    dest = [((i + 1) // 2 - 1) if i % 2 else (10 + (i + 1) // 2) for i in range(n)]
    full = np.zeros((3, n), np.uint8)
    for r in range(3):
        for i in range(n):
            a = orc.get_alleles(rows[r][i])
            full[r, dest[i]] = 255 if a[0] != 0 else (0 if (a[1] == 0 and a[2] == 0) else (1 if a[1] != a[2] else 2))
    assert np.array_equal(out, full)
    e.close()
    # random cohort with every genotype spelling
    rng = np.random.default_rng(2)
    n = 1037
    cond = rng.choice([0, 1, 2], size=n).astype(np.uint8)
    gt = random_codes(rng, 90, n, quirks=True, strict=False)
    e = fresh()
    e.set_cohort(cond)
    out = e.epi_dataset(gt)
    order = np.concatenate([np.flatnonzero(cond == 1), np.flatnonzero(cond == 0)])
    a1, a2 = gt[:, order] >> 4, gt[:, order] & 0xF
    exp = np.where((a1 == 15) | (a2 == 15), 255, np.where((a1 == 0) & (a2 == 0), 0, np.where(a1 != a2, 1, 2))).astype(np.uint8)
    assert np.array_equal(out, exp)
    e.close()


def test_fisher_table_sweep():
    # 2x2 tables straight into the Fisher p-pass (hpgv_assoc_fisher_dev): margins from balanced to very skewed, 2 k to
    # 400 k alleles, the observed table from the mode out to 30 sigma on either side and at the ends of the support --
    # every path of the kernel (whole-support scan, boundary windows that hit / miss, one or several tail rounds)
    rng = np.random.default_rng(2024)
    tabs = []
    for nn in (700, 2000, 20_000, 100_000, 400_000):
        for fr in (0.5, 0.3, 0.1, 0.02):
            for fc in (0.5, 0.3, 0.05, 0.9):
                r1, c1 = max(1, int(nn * fr)), max(1, int(nn * fc))
                r2 = nn - r1
                lo, hi = max(0, c1 - r2), min(r1, c1)
                mode = min(hi, max(lo, (r1 + 1) * (c1 + 1) // (nn + 2)))
                sigma = max(1.0, (r1 * (c1 / nn) * (1 - c1 / nn) * (nn - r1) / max(1, nn - 1)) ** 0.5)
                xs = {lo, hi, min(hi, lo + 1), max(lo, hi - 1), mode, min(hi, mode + 1), max(lo, mode - 1)}
                for t in (0.3, 1, 2, 3, 6, 12, 30):
                    for sgn in (-1, 1):
                        xs.add(int(min(hi, max(lo, round(mode + sgn * t * sigma + rng.uniform(-0.5, 0.5))))))
                for x in sorted(xs):
                    tabs.append((x, r1 - x, c1 - x, r2 - (c1 - x)))
    tabs = np.array(tabs, dtype=np.int32)
    assert (tabs >= 0).all()
    n = len(tabs)
    lf = orc.logfact(400_000 + 16)
    e = fresh()
    e.set_cohort(np.zeros(4, np.uint8))
    e.set_logfact(lf)
    d_counts, d_st = e.alloc(n * 16), e.alloc(n * 16)
    e.h2d(d_counts, tabs)
    e.assoc_fisher(d_counts, n, d_st.value, d_st.value + 8 * n)
    e.sync()
    got = e.d2h(d_st.value + 8 * n, (n,), np.float64)
    _, _, exp = orc.assoc_stats(orc.TASK_FISHER, tabs[:, 0], tabs[:, 1], tabs[:, 2], tabs[:, 3], lf)
    assert_close(got, exp, "fisher p over the table sweep")
    # relative agreement too, where the p-value is not denormal-small
    big = exp > 1e-280
    assert np.all(np.abs(got[big] - exp[big]) <= 1e-11 * exp[big]), np.max(np.abs(got[big] - exp[big]) / exp[big])
    e.close()


def test_fisher_random_tables():
    # 2x2 tables drawn at random (HPGV_SOAK_SHAPES thousand of them: 3 in the suite): totals from 1 to 400 k log-uniform, margins
    # anywhere, the observed table from the hypergeometric law or anywhere in its support; every sub-wave width of the p-pass
    rng = np.random.default_rng(int(os.environ.get("HPGV_FUZZ_SEED", "80")))
    n = 1000 * int(os.environ.get("HPGV_SOAK_SHAPES", "3"))
    nn = np.maximum(1, np.exp(rng.uniform(0, np.log(400_000), n)).astype(np.int64))
    r1 = (rng.random(n) * (nn + 1)).astype(np.int64)
    c1 = (rng.random(n) * (nn + 1)).astype(np.int64)
    lo, hi = np.maximum(0, c1 - (nn - r1)), np.minimum(r1, c1)
    x = np.where(rng.random(n) < 0.7, rng.hypergeometric(np.maximum(r1, 0), np.maximum(nn - r1, 0), np.maximum(c1, 0)) if True else 0,
                 lo + (rng.random(n) * (hi - lo + 1)).astype(np.int64))
    x = np.clip(x, lo, hi)
    tabs = np.stack([x, r1 - x, c1 - x, (nn - r1) - (c1 - x)], axis=1).astype(np.int32)
    assert (tabs >= 0).all()
    lf = orc.logfact(400_000 + 16)
    _, _, exp = orc.assoc_stats(orc.TASK_FISHER, tabs[:, 0], tabs[:, 1], tabs[:, 2], tabs[:, 3], lf)
    from helpers import shipped
    for width in shipped("fisher_width", (16, 64, 32, 8)):
        e = fresh()
        e.set_option("fisher_width", width)
        e.set_cohort(np.zeros(4, np.uint8))
        e.set_logfact(lf)
        d_counts, d_st = e.alloc(n * 16), e.alloc(n * 16)
        e.h2d(d_counts, tabs)
        e.assoc_fisher(d_counts, n, d_st.value, d_st.value + 8 * n)
        e.sync()
        got = e.d2h(d_st.value + 8 * n, (n,), np.float64)
        assert_close(got, exp, "fisher p over random tables, width %d" % width)
        big = exp > 1e-280
        assert np.all(np.abs(got[big] - exp[big]) <= 1e-11 * exp[big]), (width, np.max(np.abs(got[big] - exp[big]) / exp[big]))
        e.close()


def test_tdt_random_pedigrees_every_code():
    # pedigrees and matrices drawn at random (HPGV_SOAK_SHAPES of them: 4 in the suite): family counts, children per family,
    # absent parents, cohort widths; matrices of all 256 codes or of the common ones; chromosome X rows
    rng = np.random.default_rng(int(os.environ.get("HPGV_FUZZ_SEED", "81")))
    for _ in range(int(os.environ.get("HPGV_SOAK_SHAPES", "4"))):
        n_fam = int(rng.choice([1, int(rng.integers(2, 60)), int(rng.integers(60, 3000))]))
        max_children = int(rng.choice([1, 1, 2, int(rng.integers(3, 12))]))
        n_samples = n_fam * (2 + max_children) + int(rng.integers(0, 40))
        fam = make_families(rng, n_samples, n_fam, max_children, p_absent=float(rng.choice([0.0, 0.05, 0.3])))
        nv = int(rng.integers(1, 80))
        gt = rng.integers(0, 256, size=(nv, n_samples), dtype=np.uint8) if rng.random() < 0.5 else random_codes(rng, nv, n_samples, quirks=bool(rng.integers(0, 2)))
        is_x = (rng.random(nv) < 0.4).astype(np.uint8)
        e = fresh()
        e.set_families(n_samples, *fam)
        _tdt_check(e, gt, fam, is_x if rng.random() < 0.7 else None)
        e.close()


# ------------------------------------------------------ every code byte ----

def test_every_code_byte_through_every_scan():
    # HPGV8 is (first allele << 4 | second allele) with alleles 0 .. 14 and 15 = missing: every byte value is a code.  The matrices
    # of the other tests hold what a VCF commonly holds (alleles 0 .. 2, missing, half missing); here every one of the 256 values
    # occurs, so the nibble arithmetic of the scans (SWAR identities, class LUTs, v_perm tables) meets alleles 3 .. 14, bit 3 of a
    # nibble, and every mixture with a missing allele -- through the host batch calls, the device-resident scans and the text entry
    # points' row kernels, against the oracle.
    rng = np.random.default_rng(256)
    n_samples, nv = 1203, 211
    gt = rng.integers(0, 256, size=(nv, n_samples), dtype=np.uint8)
    gt[:16] = np.arange(256, dtype=np.uint8).reshape(16, 16).repeat(76, axis=1)[:, :n_samples]     # rows of one code each, sixteen codes per row
    is_x = (rng.random(nv) < 0.4).astype(np.uint8)
    # assoc
    cond = rng.choice([0, 1, 2], size=n_samples, p=[0.45, 0.45, 0.1]).astype(np.uint8)
    e = fresh()
    e.set_cohort(cond)
    for x in (is_x, None):
        check_assoc(e.assoc(hpgv.TASK_CHISQ, gt, x), oracle_assoc(orc.TASK_CHISQ, gt, cond, x), hpgv.TASK_CHISQ)
    e.close()
    # TDT: single-child families (the class planes) and families with several children (the scalar rule)
    for n_fam, max_children in ((390, 1), (150, 5)):
        fam = make_families(rng, n_samples, n_fam, max_children, p_absent=0.03)
        e = fresh()
        e.set_families(n_samples, *fam)
        _tdt_check(e, gt, fam, None)
        _tdt_check(e, gt, fam, is_x)
        e.close()
    # variant statistics, per-sample missing counters
    e = fresh()
    e.set_stats_cohort(n_samples)
    res = e.stats(gt)
    for i in range(nv):
        vs = orc.variant_stats(gt[i], 2)
        c8 = res["counts8"][i]
        assert list(c8[:4]) == list(vs.genotypes_count)[:4], i
        assert c8[4] == vs.missing_genotypes and c8[5] == vs.missing_alleles and c8[6] == vs.alleles_count[0] and c8[7] == vs.alleles_count[1], i
        assert_close([res["hwe_p"][i]], [vs.hw_p], "hwe p")
    acc = np.zeros(n_samples, np.int32)
    e.stats_ex(gt, sample_missing=acc, multi_cap=nv)
    assert np.array_equal(acc, orc.sample_missing(gt))
    e.close()
    # Mendelian errors
    n_trios = 400
    cols = rng.permutation(n_samples)
    f, m, c = cols[:n_trios], cols[n_trios: 2 * n_trios], cols[2 * n_trios: 3 * n_trios]
    sex = rng.integers(0, 2, size=n_trios).astype(np.uint8)
    e = fresh()
    pitch = e.set_pedigree(n_samples, f, m, c, sex)
    d_raw, d_lay = e.alloc(nv * n_samples), e.alloc(nv * pitch)
    d_err, d_child, d_isx = e.alloc(nv * 4), e.alloc(n_trios * 4), e.alloc(nv)
    e.h2d(d_raw, gt); e.h2d(d_isx, is_x)
    e.layout(hpgv.LAYOUT_MENDEL, d_raw, n_samples, nv, d_lay)
    for x, dx in ((is_x, d_isx), (None, None)):
        e.h2d(d_child, np.zeros(n_trios, np.int32))
        e.mendel_scan(d_lay, nv, d_err, dx)
        e.mendel_children(d_lay, nv, d_child, dx)
        e.sync()
        exp_err, exp_trio = orc.mendel_counts(gt, f, m, c, sex, x)
        assert np.array_equal(e.d2h(d_err, (nv,), np.int32), exp_err)
        assert np.array_equal(e.d2h(d_child, (n_trios,), np.int32), exp_trio)
    e.close()


def test_chisq_tdt_hwe_p_values_down_the_tail():
    # The statistics kernels straight from counts, the tables swept from no association to p ~ 1e-300: the absolute 1e-10
    # of north_star says nothing below 1e-10, so p-values are also held to 1e-9 RELATIVE against the oracle's
    # erfc(sqrt(x / 2)) (helpers.assert_p_close; the reference's own 1 - gsl_cdf_chisq_P quantises near 1e-16, which is
    # why the absolute bound is the contract against it and the relative one pins the engine to the oracle).
    from helpers import assert_p_close
    rng = np.random.default_rng(77)
    e = fresh()
    # chi-square association: cases carry allele 1 with frequency f + d, controls f - d
    n = 4000
    tot = rng.integers(200, 100_000, n)
    f = rng.uniform(0.05, 0.5, n)
    d = rng.uniform(0, 0.45, n) * rng.choice([0, 1, 1, 1], n)
    A2 = np.clip((tot * (f + d)).astype(np.int64), 0, tot); U2 = np.clip((tot * np.maximum(f - d, 0)).astype(np.int64), 0, tot)
    tabs = np.stack([tot - A2, A2, tot - U2, U2], axis=1).astype(np.int32)          # A1 A2 U1 U2
    d_counts, d_st = e.alloc(n * 16), e.alloc(n * 24)
    e.h2d(d_counts, tabs)
    e.assoc_chisq(d_counts, n, d_st.value, d_st.value + 8 * n, d_st.value + 16 * n)
    e.sync()
    got = e.d2h(d_st, (3, n), np.float64)
    odds, chisq, p = orc.assoc_stats(orc.TASK_CHISQ, tabs[:, 0], tabs[:, 1], tabs[:, 2], tabs[:, 3])
    assert_close(got[0], odds, "odds"); assert_close(got[1], chisq, "chisq"); assert_p_close(got[2], p, "chisq p")
    assert (p[np.isfinite(p)] < 1e-100).any() and (p[np.isfinite(p)] > 0.01).any()      # the sweep reaches both ends
    # TDT: transmitted / untransmitted tallies from balanced to one-sided
    t1 = rng.integers(0, 20_000, n); t2 = (t1 * rng.uniform(0, 1, n)).astype(np.int64)
    tu = np.stack([t1, t2], axis=1).astype(np.int32)
    tu[:5] = [[0, 0], [1, 0], [0, 1], [46340, 0], [30000, 30000]]
    d_tu = e.alloc(n * 8)
    e.h2d(d_tu, tu)
    e.tdt_stats(d_tu, n, d_st.value, d_st.value + 8 * n, d_st.value + 16 * n)
    e.sync()
    got = e.d2h(d_st, (3, n), np.float64)
    odds, chisq, p = orc.tdt_stats(tu[:, 0], tu[:, 1])
    assert_close(got[0], odds, "tdt odds"); assert_close(got[1], chisq, "tdt chisq"); assert_p_close(got[2], p, "tdt p")
    # Hardy-Weinberg: genotype counts from equilibrium to all-homozygous
    nn = rng.integers(50, 100_000, n); q = rng.uniform(0.02, 0.98, n); infl = rng.uniform(0, 1, n) * rng.choice([0, 1, 1], n)
    het = (2 * q * (1 - q) * nn * (1 - infl)).astype(np.int64)
    aa = ((nn - het) * (1 - q)).astype(np.int64); bb = nn - het - aa
    c8 = np.zeros((n, 8), np.int32)
    c8[:, 0], c8[:, 1], c8[:, 3] = aa, het, bb
    c8[:, 6], c8[:, 7] = 2 * aa + het, 2 * bb + het
    d_c8 = e.alloc(n * 32)
    e.h2d(d_c8, c8)
    e.stats_hwe(d_c8, n, d_st.value, d_st.value + 8 * n)
    e.sync()
    got = e.d2h(d_st, (2, n), np.float64)
    exp = np.array([orc.hwe(int(a), int(h), int(b)) for a, h, b in zip(aa, het, bb)])
    assert_close(got[0], exp[:, 0], "hwe chi2"); assert_p_close(got[1], exp[:, 1], "hwe p")
    e.close()
