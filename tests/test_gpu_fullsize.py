"""GPU parity at BASELINE.json's full single-GPU sizes.  The oracle cannot scan
1e10 genotypes in seconds, so every variant is covered by size-independent
properties (determinism, shard linearity, parity of allele totals, bounds) and
a sample (every 1000th variant plus the first / last 256 of the cohort) is
compared with the oracle bit for bit (counts) / within 1e-10 (statistics)."""
import numpy as np
import pytest

from helpers import assert_close, assert_p_close, hpgv
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def _sample_idx(n):
    return np.unique(np.concatenate([np.arange(min(256, n)), np.arange(0, n, 1000), np.arange(max(0, n - 256), n)]))


def _rows(v0, idx, n_samples):
    return np.stack([orc.synth_matrix(v0 + int(v), 1, n_samples, n_samples)[0] for v in idx])


# (1M, 10k) = BASELINE configs[1] / [2]; (1.25M, 50k) = the metric cohort's 1/8 shard; (1M, 100k) = one 100 GB tile of a GPU's
# shard of configs[4] (40M x 100k on 8 GPUs: 5 such tiles per GPU)
@pytest.mark.parametrize("V,N", [(1_000_000, 10_000), (1_250_000, 50_000), (1_000_000, 100_000)])
def test_assoc_chisq_and_fisher_full_size(V, N):
    e = hpgv.Engine(0)
    cond = (np.arange(N) % 2).astype(np.uint8)
    nA, nU, pitch = e.set_cohort(cond)
    try:
        d_gt = e.alloc(V * pitch)
    except hpgv.HpgvError as err:
        pytest.fail("cannot allocate the %.0f GB genotype tile of the %d x %d case on this GPU: %s" % (V * pitch / 1e9, V, N, err))
    d_counts, d_counts2 = e.alloc(V * 16), e.alloc(V * 16)
    d_st = e.alloc(V * 24)
    e.synth(hpgv.LAYOUT_ASSOC, 0, V, d_gt)
    e.assoc_scan(d_gt, V, d_counts)
    b = d_st.value
    e.assoc_chisq(d_counts, V, b, b + 8 * V, b + 16 * V)
    e.sync()
    counts = e.d2h(d_counts, (V, 4), np.int32)
    st = e.d2h(d_st, (3, V), np.float64)
    # properties over ALL variants
    assert ((counts[:, 0] + counts[:, 1]) % 2 == 0).all() and ((counts[:, 2] + counts[:, 3]) % 2 == 0).all()
    assert (counts >= 0).all()
    assert (counts[:, 0] + counts[:, 1] <= 2 * nA).all() and (counts[:, 2] + counts[:, 3] <= 2 * nU).all()
    miss = 1 - (counts.sum(1) / (2.0 * (nA + nU))).mean()
    assert 0.009 < miss < 0.011                                 # generator: 1 % missing
    p = st[2]
    assert ((p >= 0) & (p <= 1)).all() and np.isfinite(st[1]).all()
    # determinism + shard linearity: two halves scanned separately give the same table
    half = V // 2 + 7
    e.assoc_scan(d_gt, half, d_counts2)
    e.assoc_scan(d_gt.value + half * pitch, V - half, d_counts2.value + half * 16)
    e.sync()
    assert np.array_equal(e.d2h(d_counts2, (V, 4), np.int32), counts)
    # sampled oracle comparison
    idx = _sample_idx(V)
    for lo in range(0, len(idx), 256):
        sel = idx[lo: lo + 256]
        A1, A2, U1, U2 = orc.assoc_counts(_rows(0, sel, N), cond)
        assert np.array_equal(counts[sel], np.stack([A1, A2, U1, U2], 1))
        odds, chisq, pv = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
        assert_close(st[0][sel], odds, "odds"); assert_close(st[1][sel], chisq, "chisq"); assert_p_close(st[2][sel], pv, "p")
    # Fisher p-pass on the same counts (BASELINE configs[2])
    lf = orc.logfact(N * 10)
    e.set_logfact(lf)
    e.assoc_fisher(d_counts, V, b, b + 8 * V)
    e.sync()
    fp = e.d2h(d_st.value + 8 * V, (V,), np.float64)
    assert ((fp >= 0) & (fp <= 1)).all()
    sel = idx[:: max(1, len(idx) // 300)]
    _, _, pv = orc.assoc_stats(orc.TASK_FISHER, counts[sel, 0], counts[sel, 1], counts[sel, 2], counts[sel, 3], lf)
    assert_close(fp[sel], pv, "fisher p")
    e.close()


def test_tdt_full_size():
    # BASELINE configs[3]: 2M SNP x 5k trios
    V, n_tr = 2_000_000, 5000
    e = hpgv.Engine(0)
    k = np.arange(n_tr)
    fam = (3 * k, 3 * k + 1, np.arange(n_tr + 1), 3 * k + 2, (k % 2).astype(np.uint8))
    n_fast, n_slow, pitch = e.set_families(3 * n_tr, *fam)
    d_gt, d_tu, d_tu2, d_st = e.alloc(V * pitch), e.alloc(V * 8), e.alloc(V * 8), e.alloc(V * 24)
    e.synth(hpgv.LAYOUT_TDT, 0, V, d_gt)
    e.tdt_scan(d_gt, V, d_tu)
    b = d_st.value
    e.tdt_stats(d_tu, V, b, b + 8 * V, b + 16 * V)
    e.sync()
    tu = e.d2h(d_tu, (V, 2), np.int32)
    st = e.d2h(d_st, (3, V), np.float64)
    assert (tu >= 0).all() and (tu.sum(1) <= 2 * n_tr).all() and tu.sum() > 0
    assert ((st[2] >= 0) & (st[2] <= 1)).all()
    half = V // 3
    e.tdt_scan(d_gt, half, d_tu2)
    e.tdt_scan(d_gt.value + half * pitch, V - half, d_tu2.value + half * 8)
    e.sync()
    assert np.array_equal(e.d2h(d_tu2, (V, 2), np.int32), tu)
    idx = _sample_idx(V)
    for lo in range(0, len(idx), 256):
        sel = idx[lo: lo + 256]
        t1, t2 = orc.tdt_counts(_rows(0, sel, 3 * n_tr), *fam)
        assert np.array_equal(tu[sel, 0], t1) and np.array_equal(tu[sel, 1], t2)
        odds, chisq, p = orc.tdt_stats(t1, t2)
        assert_close(st[0][sel], odds, "odds"); assert_close(st[1][sel], chisq, "chisq"); assert_p_close(st[2][sel], p, "p")
    e.close()


def test_stats_full_size():
    V, N = 1_000_000, 10_000
    e = hpgv.Engine(0)
    pitch = e.set_stats_cohort(N)
    d_gt, d_c8, d_hw = e.alloc(V * pitch), e.alloc(V * 32), e.alloc(V * 16)
    e.synth(hpgv.LAYOUT_STATS, 0, V, d_gt)
    e.stats_scan(d_gt, V, d_c8)
    e.stats_hwe(d_c8, V, d_hw, d_hw.value + 8 * V)
    e.sync()
    c8 = e.d2h(d_c8, (V, 8), np.int32)
    hw = e.d2h(d_hw, (2, V), np.float64)
    assert (c8[:, :4].sum(1) + c8[:, 4] == N).all()             # biallelic cohort: cells + missing = samples
    assert (c8[:, 6] + c8[:, 7] + c8[:, 5] == 2 * N).all()      # allele counts + missing alleles = 2N
    assert ((hw[1] >= 0) & (hw[1] <= 1)).all()
    idx = _sample_idx(V)[::4]
    rows = _rows(0, idx, N)
    for j, v in enumerate(idx):
        vs = orc.variant_stats(rows[j], 2)
        assert list(c8[v, :4]) == list(vs.genotypes_count)[:4]
        assert (c8[v, 4], c8[v, 5], c8[v, 6], c8[v, 7]) == (vs.missing_genotypes, vs.missing_alleles,
                                                              vs.alleles_count[0], vs.alleles_count[1])
        assert_close([hw[0][v]], [vs.hw_chi2], "hwe chi2"); assert_p_close([hw[1][v]], [vs.hw_p], "hwe p")
    e.close()


def _layout_conditions(e, cond, n_samples):
    """condition vector per ROW POSITION of the assoc layout (pads are 'other')."""
    nA, nU, pitch = e.assoc_layout()
    segA = (nA + 15) // 16 * 16
    pos = np.full(pitch, orc.COND_OTHER, np.uint8)
    pos[:nA] = orc.AFFECTED
    pos[segA: segA + nU] = orc.UNAFFECTED
    return pos, pitch


def test_assoc_every_variant_against_the_oracle_c2():
    """SURVEY 8d parity protocol for C2: EVERY variant of the 1M x 10k cohort.  The device matrix is
    copied back in slabs and scanned by the (OpenMP) oracle, so the comparison also covers the
    on-device generator at full size."""
    V, N = 1_000_000, 10_000
    e = hpgv.Engine(0)
    cond = (np.arange(N) % 2).astype(np.uint8)
    e.set_cohort(cond)
    pos_cond, pitch = _layout_conditions(e, cond, N)
    d_gt, d_counts, d_st = e.alloc(V * pitch), e.alloc(V * 16), e.alloc(V * 24)
    e.synth(hpgv.LAYOUT_ASSOC, 0, V, d_gt)
    e.assoc_scan(d_gt, V, d_counts)
    b = d_st.value
    e.assoc_chisq(d_counts, V, b, b + 8 * V, b + 16 * V)
    e.sync()
    counts = e.d2h(d_counts, (V, 4), np.int32)
    st = e.d2h(d_st, (3, V), np.float64)
    slab = 50_000
    for lo in range(0, V, slab):
        n = min(slab, V - lo)
        rows = e.d2h(d_gt.value + lo * pitch, (n, pitch), np.uint8)
        A1, A2, U1, U2 = orc.assoc_counts(rows, pos_cond)
        assert np.array_equal(counts[lo: lo + n], np.stack([A1, A2, U1, U2], 1)), lo
        odds, chisq, p = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
        assert_close(st[0][lo: lo + n], odds, "odds")
        assert_close(st[1][lo: lo + n], chisq, "chisq")
        assert_p_close(st[2][lo: lo + n], p, "p")
    # the slab copied back is what the oracle generator produces (first and last slab, column order undone)
    nA, nU, _ = e.assoc_layout()
    segA = (nA + 15) // 16 * 16
    for lo in (0, V - 2000):
        rows = e.d2h(d_gt.value + lo * pitch, (2000, pitch), np.uint8)
        ref = orc.synth_matrix(lo, 2000, N, N)
        assert np.array_equal(rows[:, :nA], ref[:, cond == 1]) and np.array_equal(rows[:, segA: segA + nU], ref[:, cond == 0])
    e.close()


def test_tdt_every_variant_against_the_oracle_c4():
    """C4 (2M SNP x 5k trios), every variant: raw HPGV8 slabs are generated on the device in VCF column
    order, laid out (recoded) and scanned there, and the same raw slabs go through the oracle."""
    V, n_tr = 2_000_000, 5000
    N = 3 * n_tr
    e = hpgv.Engine(0)
    k = np.arange(n_tr)
    fam = (3 * k, 3 * k + 1, np.arange(n_tr + 1), 3 * k + 2, (k % 2).astype(np.uint8))
    _, _, pitch = e.set_families(N, *fam)
    slab = 100_000
    raw_pitch = (N + 15) // 16 * 16
    d_raw, d_lay, d_tu = e.alloc(slab * raw_pitch), e.alloc(slab * pitch), e.alloc(slab * 8)
    for lo in range(0, V, slab):
        e.synth_raw(lo, slab, N, raw_pitch, d_raw)
        e.layout(hpgv.LAYOUT_TDT, d_raw, raw_pitch, slab, d_lay)
        e.tdt_scan(d_lay, slab, d_tu)
        e.sync()
        tu = e.d2h(d_tu, (slab, 2), np.int32)
        raw = e.d2h(d_raw, (slab, raw_pitch), np.uint8)
        t1, t2 = orc.tdt_counts(raw, *fam)
        assert np.array_equal(tu[:, 0], t1) and np.array_equal(tu[:, 1], t2), lo
    e.close()


def test_epistasis_pair_scan_full_size():
    """4096 SNPs x 10 000 samples x 10 folds (8.4 M pairs): properties over every pair, the ranking against the dense
    scan, and an oracle sample of pairs bit for bit."""
    from helpers import epi_random_dataset, epi_random_folds
    rng = np.random.default_rng(2024)
    V, nA, nU, K, N = 4096, 5200, 4800, 10, 10
    data = epi_random_dataset(rng, V, nA, nU, p_missing=0.01)
    data[100, :nA] = rng.choice([1, 2], size=nA); data[3000, :nA] = rng.choice([1, 2], size=nA)     # a planted interaction
    fold = epi_random_folds(rng, nA, nU, K)
    e = hpgv.Engine(0)
    e.epi_set_dataset(data, nA, nU)
    e.epi_set_folds(fold, K)
    res = e.epi_rank_pairs(hpgv.EPI_TESTING, N)
    assert (int(res["i"][0][0]), int(res["j"][0][0])) == (100, 3000)
    # dense scan of two row bands: the ranking's entries inside the bands are the bands' best
    for lo, hi in ((64, 192), (2944, 3072)):
        acc, rm = e.epi_scan_pairs(hpgv.EPI_TESTING, lo, hi)
        with np.errstate(invalid="ignore"):
            assert ((acc >= 0) & (acc <= 1) | np.isnan(acc)).all()
        assert (rm < 512).all()
        pairs_i = np.concatenate([np.full(V - 1 - i, i) for i in range(lo, hi)])
        pairs_j = np.concatenate([np.arange(i + 1, V) for i in range(lo, hi)])
        for f in range(K):
            inside = [(int(a), int(b), float(c)) for a, b, c in zip(res["i"][f], res["j"][f], res["accuracy"][f]) if lo <= a < hi]
            best = np.nanmax(acc[f])
            assert all(c <= best for _, _, c in inside)
            for a, b, c in inside:
                p = np.flatnonzero((pairs_i == a) & (pairs_j == b))[0]
                assert acc[f][p] == c
            if res["n"][f] == N:                                     # nothing in the band beats the ranking's last entry unless listed
                thr = res["accuracy"][f][N - 1]
                above = np.flatnonzero(acc[f] > thr)
                assert set(zip(pairs_i[above].tolist(), pairs_j[above].tolist())) <= set(zip(res["i"][f].tolist(), res["j"][f].tolist()))
    # oracle sample
    masks = orc.fold_masks_from_assignment(fold, K)
    sel = [(int(a), int(b)) for a, b in zip(rng.integers(0, V - 1, 60), rng.integers(0, V, 60)) if a < b] + [(100, 3000), (0, 1), (V - 2, V - 1)]
    for (i, j) in sel:
        ea, em, _ = orc.epi_model([data[i], data[j]], nA, nU, masks, 0)
        ib = i // 64 * 64
        acc, rm = e.epi_scan_pairs(hpgv.EPI_TESTING, i, i + 1)
        p = j - i - 1
        assert np.array_equal(rm[:, p], em.astype(np.uint16)), (i, j)
        assert np.all((acc[:, p] == ea) | (np.isnan(acc[:, p]) & np.isnan(ea))), (i, j)
    # counts: every sample called at both SNPs falls in exactly one cell
    combs = np.array(sel[:20], np.int32)
    aff, unaff = e.epi_counts(combs)
    for n_, (i, j) in enumerate(sel[:20]):
        both = (data[i] < 3) & (data[j] < 3)
        assert aff[n_].sum() == both[:nA].sum() and unaff[n_].sum() == both[nA:].sum()
    e.close()
