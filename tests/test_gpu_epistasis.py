"""Epistasis / MDR counting on the GPU (hpgv_epi_*) against the oracle and the reference's unit-test vectors
(tests/golden/reference_kats.json "epistasis_model").  Integer counts and risky-cell masks bit-exact; the
balanced accuracy is a handful of IEEE double operations on those integers, so it is compared exactly too."""
import numpy as np
import pytest

from helpers import epi_random_dataset, epi_random_folds, epi_unpad, hpgv
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = hpgv.Engine(0)
    yield e
    e.close()


def _same(a, b):
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def test_counts_reference_kat(eng, goldens):
    k = goldens["kats"]["epistasis_model"]["counts"]
    nA, nU = k["num_affected"], k["num_unaffected"]
    rows = np.stack([epi_unpad(r, nA, nU) for r in k["padded_rows"]])
    eng.epi_set_dataset(rows, nA, nU)
    aff, unaff = eng.epi_counts([k["order2"]["rows"]])
    assert aff[0].tolist() == k["order2"]["aff"] and unaff[0].tolist() == k["order2"]["unaff"]
    aff, unaff = eng.epi_counts([k["order3"]["rows"]])
    assert aff[0].tolist() == k["order3"]["aff"] and unaff[0].tolist() == k["order3"]["unaff"]


def test_counts_all_folds_reference_kat(eng, goldens):
    k = goldens["kats"]["epistasis_model"]["counts_all_folds"]
    nA, nU, nf = k["num_affected"], k["num_unaffected"], k["num_folds"]
    rows = np.stack([epi_unpad(r, nA, nU) for r in k["padded_rows"]])
    eng.epi_set_dataset(rows, nA, nU)
    eng.epi_set_fold_masks(np.array(k["padded_fold_masks"], np.uint8), nf)     # the reference's own padded mask array
    aff, unaff = eng.epi_counts([k["order2"]["rows"]], all_folds=True)
    assert aff[:, 0, :].tolist() == k["order2"]["aff"] and unaff[:, 0, :].tolist() == k["order2"]["unaff"]
    aff, unaff = eng.epi_counts([k["order3"]["rows"]], all_folds=True)
    for f, cells in k["order3"]["some_cells"].items():
        for c, (ea, eu) in cells.items():
            assert (aff[int(f)][0][int(c)], unaff[int(f)][0][int(c)]) == (ea, eu), (f, c)
    bad = np.array(k["padded_fold_masks"], np.uint8)
    bad[1][0] = 0                                                     # sample 0 now left out of two folds
    with pytest.raises(hpgv.HpgvError):
        eng.epi_set_fold_masks(bad, nf)


def test_confusion_reference_kats_through_the_pair_scan(eng, goldens):
    # the order-2 confusion-matrix vectors: with the KAT's fold mask as one fold of a 2-fold partition the scan must
    # give the accuracy of the KAT's matrix whenever the MDR rule picks the KAT's risky cells; the matrices
    # themselves are pinned on the oracle (tests/test_epi_oracle.py), here the scan is compared with the oracle
    for case in goldens["kats"]["epistasis_model"]["confusion"]:
        if len(case["padded_rows"]) != 2:
            continue
        nA, nU = case["num_affected"], case["num_unaffected"]
        rows = np.stack([epi_unpad(r, nA, nU) for r in case["padded_rows"]])
        mask = epi_unpad(case["padded_fold_mask"], nA, nU)
        fold = np.where(mask == 1, 1, 0).astype(np.int32)            # training part of fold 0 = fold 1's testing samples
        if len(set(fold.tolist())) < 2:
            continue
        eng.epi_set_dataset(rows, nA, nU)
        eng.epi_set_folds(fold, 2)
        masks = orc.fold_masks_from_assignment(fold, 2)
        for subset in (hpgv.EPI_TESTING, hpgv.EPI_TRAINING):
            acc, rm = eng.epi_scan_pairs(subset)
            eacc, erm = orc.epi_scan_pairs(rows, nA, nU, masks, subset)
            assert np.array_equal(rm, erm.astype(np.uint16)) and _same(acc, eacc), case["line"]


@pytest.mark.parametrize("v,nA,nU,k", [(70, 37, 52, 5), (9, 16, 16, 2), (130, 300, 420, 10), (65, 5, 4, 3), (40, 1100, 1300, 4),
                                       (33, 64, 64, 1), (20, 700, 650, 16), (12, 2500, 2300, 8)])
def test_pair_scan_matches_the_oracle(eng, v, nA, nU, k):
    rng = np.random.default_rng(v * 7 + k)
    data = epi_random_dataset(rng, v, nA, nU, p_missing=0.03)
    fold = epi_random_folds(rng, nA, nU, k)
    eng.epi_set_dataset(data, nA, nU)
    eng.epi_set_folds(fold, k)
    masks = orc.fold_masks_from_assignment(fold, k)
    for subset in (hpgv.EPI_TESTING, hpgv.EPI_TRAINING):
        acc, rm = eng.epi_scan_pairs(subset)
        eacc, erm = orc.epi_scan_pairs(data, nA, nU, masks, subset)
        assert acc.shape == eacc.shape == (k, v * (v - 1) // 2)
        assert np.array_equal(rm, erm.astype(np.uint16)), "risky cells differ"
        assert _same(acc, eacc), "accuracy differs"
    # a band of rows gives the same numbers as the whole scan
    if v > 8:
        a2, m2 = eng.epi_scan_pairs(hpgv.EPI_TRAINING, 3, 8)
        lo = 3 * (2 * v - 3 - 1) // 2
        assert _same(a2, acc[:, lo: lo + a2.shape[1]]) and np.array_equal(m2, rm[:, lo: lo + a2.shape[1]])


def test_counts_against_the_oracle_order_2_and_3(eng):
    rng = np.random.default_rng(99)
    v, nA, nU, k = 25, 333, 401, 7
    data = epi_random_dataset(rng, v, nA, nU, p_missing=0.05)
    fold = epi_random_folds(rng, nA, nU, k)
    eng.epi_set_dataset(data, nA, nU)
    eng.epi_set_folds(fold, k)
    masks = orc.fold_masks_from_assignment(fold, k)
    for order in (2, 3):
        combs = np.array([sorted(rng.choice(v, size=order, replace=False)) for _ in range(40)], np.int32)
        aff, unaff = eng.epi_counts(combs)
        faff, funaff = eng.epi_counts(combs, all_folds=True)
        for n, comb in enumerate(combs):
            rows = [data[c] for c in comb]
            ea, eu = orc.epi_counts(rows, nA, nU)
            assert np.array_equal(aff[n], ea) and np.array_equal(unaff[n], eu)
            fa, fu = orc.epi_counts_all_folds(rows, nA, nU, masks)
            assert np.array_equal(faff[:, n, :], fa) and np.array_equal(funaff[:, n, :], fu)


def test_ranking_is_the_top_of_the_dense_scan(eng):
    rng = np.random.default_rng(17)
    v, nA, nU, k, n = 150, 180, 220, 5, 12
    data = epi_random_dataset(rng, v, nA, nU)
    # plant an interaction so the best models are not all ties
    data[7, :nA] = rng.choice([1, 2], size=nA); data[91, :nA] = rng.choice([1, 2], size=nA)
    fold = epi_random_folds(rng, nA, nU, k)
    eng.epi_set_dataset(data, nA, nU)
    eng.epi_set_folds(fold, k)
    acc, rm = eng.epi_scan_pairs(hpgv.EPI_TESTING)
    res = eng.epi_rank_pairs(hpgv.EPI_TESTING, n)
    pairs = [(i, j) for i in range(v) for j in range(i + 1, v)]
    for f in range(k):
        a = np.where(np.isnan(acc[f]), -np.inf, acc[f])
        order = sorted(range(len(pairs)), key=lambda p: (-a[p], pairs[p]))[:n]
        assert res["n"][f] == n
        assert [(int(x), int(y)) for x, y in zip(res["i"][f], res["j"][f])] == [pairs[p] for p in order]
        assert np.array_equal(res["accuracy"][f], acc[f][order]) and np.array_equal(res["risky"][f], rm[f][order])
    assert (7, 91) == (int(res["i"][0][0]), int(res["j"][0][0]))


def test_epistasis_error_paths(eng):
    e = hpgv.Engine(0)
    with pytest.raises(hpgv.HpgvError):
        e.L.hpgv_epi_set_folds.argtypes  # noqa: B018  (keeps the symbol referenced)
        e._chk(e.L.hpgv_epi_set_folds(e.h, None, 2))                  # no dataset yet
    data = np.zeros((4, 10), np.uint8)
    e.epi_set_dataset(data, 5, 5)
    with pytest.raises(hpgv.HpgvError):
        e.epi_set_folds(np.array([0, 1, 2, 3, 4, 0, 1, 2, 3, 9], np.int32), 5)   # fold id out of range
    with pytest.raises(hpgv.HpgvError):
        e.epi_set_folds(np.zeros(10, np.int32), 17)                  # more folds than supported
    with pytest.raises(hpgv.HpgvError):
        e.epi_counts(np.array([[0, 4]], np.int32))                   # SNP index outside the dataset
    e.close()
