"""The epistasis / MDR oracle (oracle/hpgv_epi_oracle.c) against the reference's own unit tests
(tests/golden/reference_kats.json "epistasis_model": test/test_epistasis_model.c, test/test_mdr.c) and
against an independent numpy statement of the same definitions.  CPU only."""
import numpy as np
import pytest

from helpers import epi_counts_from_reference_masks, epi_random_dataset, epi_random_folds, epi_unpad
from oracle import pyoracle as orc


@pytest.fixture(scope="module")
def kat(goldens):
    return goldens["kats"]["epistasis_model"]


def test_counts_reference_kat(kat):
    k = kat["counts"]
    nA, nU = k["num_affected"], k["num_unaffected"]
    rows = [epi_unpad(r, nA, nU) for r in k["padded_rows"]]
    for name in ("order2", "order3"):
        aff, unaff = orc.epi_counts([rows[i] for i in k[name]["rows"]], nA, nU)
        assert list(aff) == k[name]["aff"] and list(unaff) == k[name]["unaff"], name


def test_counts_order_4_from_the_reference_masks(kat):
    # test_get_masks (test/test_epistasis_model.c:34-100) holds the masks of FOUR SNPs: the order-4 counts are
    # combination_counts' arithmetic on them, and sub-combinations of two and three of the SNPs give the same answer as
    # the reference's own order-2 / order-3 count vectors
    k = kat["masks_order4"]
    nA, nU = k["num_affected"], k["num_unaffected"]
    rows = [epi_unpad(r, nA, nU) for r in k["padded_rows"]]
    ea, eu = epi_counts_from_reference_masks(k["masks"], nA, nU, 4)
    aff, unaff = orc.epi_counts(rows, nA, nU)
    assert len(ea) == 81 and list(aff) == ea and list(unaff) == eu
    assert sum(ea) == nA and sum(eu) == nU                            # no missing call in the vectors: every sample is in one cell
    c = kat["counts"]
    a2, u2 = epi_counts_from_reference_masks(k["masks"][:2], nA, nU, 2)
    a3, u3 = epi_counts_from_reference_masks(k["masks"][:3], nA, nU, 3)
    assert a2 == c["order2"]["aff"] and u2 == c["order2"]["unaff"] and a3 == c["order3"]["aff"] and u3 == c["order3"]["unaff"]


@pytest.mark.parametrize("order", [4, 5])
def test_model_of_higher_orders_is_consistent_with_its_counts(order):
    # orc_epi_model_wide for orders beyond the reference's tests: its risky cells are the MDR rule on the training counts, its
    # confusion matrix the sums of the evaluated part's counts over those cells (two independent routes through the oracle)
    rng = np.random.default_rng(order)
    nA, nU, k = 90, 70, 4
    data = epi_random_dataset(rng, order, nA, nU, p_missing=0.03)
    fold = epi_random_folds(rng, nA, nU, k)
    masks = orc.fold_masks_from_assignment(fold, k)
    rows = [data[s] for s in range(order)]
    tr_a, tr_u = orc.epi_counts_all_folds(rows, nA, nU, masks)
    all_a, all_u = orc.epi_counts(rows, nA, nU)
    for subset in (0, 1):
        acc, rm, mat = orc.epi_model_wide(rows, nA, nU, masks, subset)
        for f in range(k):
            risky = [c for c in range(3 ** order) if orc.mdr_high_risk2(int(tr_a[f][c]), int(tr_u[f][c]), nA, nU)]
            bits = np.zeros(8, np.uint32)
            for c in risky:
                bits[c // 32] |= np.uint32(1) << np.uint32(c % 32)
            assert np.array_equal(rm[f], bits)
            ev_a = tr_a[f] if subset == 1 else all_a - tr_a[f]
            ev_u = tr_u[f] if subset == 1 else all_u - tr_u[f]
            tp, fp = int(ev_a[risky].sum()), int(ev_u[risky].sum())
            assert mat[f][0] == tp and mat[f][2] == fp


def test_counts_all_folds_reference_kat(kat):
    k = kat["counts_all_folds"]
    nA, nU = k["num_affected"], k["num_unaffected"]
    rows = [epi_unpad(r, nA, nU) for r in k["padded_rows"]]
    masks = np.stack([epi_unpad(m, nA, nU) for m in k["padded_fold_masks"]])
    assert ((masks == 0).sum(axis=0) == 1).all()                     # the KAT's masks are a partition into testing folds
    aff, unaff = orc.epi_counts_all_folds([rows[i] for i in k["order2"]["rows"]], nA, nU, masks)
    assert aff.tolist() == k["order2"]["aff"] and unaff.tolist() == k["order2"]["unaff"]
    aff, unaff = orc.epi_counts_all_folds(rows, nA, nU, masks)
    for f, cells in k["order3"]["some_cells"].items():
        for c, (ea, eu) in cells.items():
            assert (aff[int(f)][int(c)], unaff[int(f)][int(c)]) == (ea, eu), (f, c)


def test_confusion_matrix_reference_kats(kat):
    for case in kat["confusion"]:
        nA, nU = case["num_affected"], case["num_unaffected"]
        rows = [epi_unpad(r, nA, nU) for r in case["padded_rows"]]
        mask = epi_unpad(case["padded_fold_mask"], nA, nU)
        training = case["subset"] == "TRAINING"
        m = orc.epi_confusion(case["risky"], rows, nA, nU, mask, 1 if training else 0,
                              case["training_size"] if training else case["testing_size"])
        assert m == case["matrix"], case["line"]


def test_evaluation_formulas_reference_kat(kat):
    for case in kat["evaluate"]:
        for fn, name in ((0, "CA"), (1, "BA"), (3, "GAMMA"), (4, "TAU_B")):
            assert abs(orc.epi_evaluate(case["matrix"], fn) - case[name]) <= 1e-6, (case["matrix"], name)
    assert orc.epi_evaluate([40, 2, 4, 10], 1) == ((40 / 42) + (10 / 14)) / 2
    assert np.isnan(orc.epi_evaluate([0, 0, 3, 4], 1))                # no affected sample in the subset: 0/0 as in C


def test_mdr_high_risk_reference_kats(kat):
    k = kat["mdr_high_risk"]
    nA, nU = k["num_affected"], k["num_unaffected"]
    risky = [i for i, (a, u) in enumerate(k["scalar"]["counts"]) if orc.mdr_high_risk(a, u, nA, nU)]
    assert risky == k["scalar"]["risky_indices"]
    got = [orc.mdr_high_risk2(a, u, nA, nU) for a, u in zip(k["vector"]["aff"], k["vector"]["unaff"])]
    assert got == k["vector"]["risky"]
    assert not orc.mdr_high_risk2(0, 0, nA, nU) and not orc.mdr_high_risk(0, 0, nA, nU)      # empty cell: 0/0, never risky
    assert orc.mdr_high_risk2(1, 8, 10, 80)                          # exactly the cohort's ratio: >= holds


def _numpy_model(ri, rj, nA, nU, fold, k, subset):
    """Independent statement: counts by bincount, the MDR rule in float32, confusion from the cell counts."""
    n = nA + nU
    valid = (ri < 3) & (rj < 3)
    cell = np.where(valid, ri.astype(int) * 3 + rj, 9)
    is_aff = np.arange(n) < nA
    acc, masks = [], []
    ratio = np.float32(nA) / np.float32(nU)
    for f in range(k):
        train = fold != f
        ca = np.bincount(cell[train & is_aff], minlength=10)[:9].astype(np.float32)
        cu = np.bincount(cell[train & ~is_aff], minlength=10)[:9].astype(np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            total = ca + cu
            prop = cu * ratio
            nu_ = prop * (total / (prop + ca))
            risky = (total - nu_) >= nu_
        sel = train if subset == 1 else ~train
        pred = valid & risky[np.minimum(cell, 8)] & sel
        tp, fp = int((pred & is_aff).sum()), int((pred & ~is_aff).sum())
        sa, su = int((sel & is_aff).sum()), int((sel & ~is_aff).sum())
        with np.errstate(invalid="ignore", divide="ignore"):
            acc.append((np.float64(tp) / np.float64(sa) + np.float64(su - fp) / np.float64(su)) / 2)
        masks.append(int(sum(1 << c for c in range(9) if risky[c])))
    return np.array(acc), np.array(masks, np.uint32)


@pytest.mark.parametrize("nA,nU,k", [(37, 52, 5), (16, 16, 2), (100, 33, 10), (5, 4, 3)])
def test_pair_scan_against_numpy(nA, nU, k):
    rng = np.random.default_rng(nA * 1000 + nU)
    v = 12
    data = epi_random_dataset(rng, v, nA, nU, p_missing=0.05)
    fold = epi_random_folds(rng, nA, nU, k)
    masks = orc.fold_masks_from_assignment(fold, k)
    for subset in (0, 1):
        acc, rm = orc.epi_scan_pairs(data, nA, nU, masks, subset)
        p = 0
        for i in range(v):
            for j in range(i + 1, v):
                ea, em = _numpy_model(data[i], data[j], nA, nU, fold, k, subset)
                got = acc[:, p]
                assert np.array_equal(rm[:, p], em), (i, j)
                assert np.all((got == ea) | (np.isnan(got) & np.isnan(ea))), (i, j, got, ea)
                p += 1
        assert p == acc.shape[1]


def test_model_matrices_consistent_with_counts():
    rng = np.random.default_rng(5)
    nA, nU, k = 40, 60, 4
    data = epi_random_dataset(rng, 3, nA, nU)
    fold = epi_random_folds(rng, nA, nU, k)
    masks = orc.fold_masks_from_assignment(fold, k)
    for rows in ([data[0], data[1]], [data[0], data[1], data[2]]):
        acc, rm, mat = orc.epi_model(rows, nA, nU, masks, 0)
        aff, unaff = orc.epi_counts_all_folds(rows, nA, nU, masks)
        tot_a, tot_u = orc.epi_counts(rows, nA, nU)
        for f in range(k):
            cells = [c for c in range(3 ** len(rows)) if rm[f] >> c & 1]
            tp = sum(int(tot_a[c] - aff[f][c]) for c in cells)             # testing part = everybody minus the training part
            fp = sum(int(tot_u[c] - unaff[f][c]) for c in cells)
            assert (int(mat[f][0]), int(mat[f][2])) == (tp, fp)
            assert int(mat[f][0] + mat[f][1]) == int(((fold == f) & (np.arange(nA + nU) < nA)).sum())


def test_k_folds_reference_kat_and_mask_layout(kat):
    """get_k_folds / get_k_folds_masks of the host mirror (cross_validation.c:16-100,247-281): the fold sizes the
    reference's test pins, cases and controls never mixed, the padded mask layout (test_cross_validation.c:36-283,
    664-716).  Pure host code: no engine call."""
    import ctypes as C
    from importlib import import_module
    b = import_module("hpg-variant_amd._build")
    L = C.CDLL(b.HOSTLIB)
    L.get_k_folds.restype = C.POINTER(C.POINTER(C.c_int))
    L.get_k_folds.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.POINTER(C.c_uint))]
    L.get_k_folds_masks.restype = C.POINTER(C.c_uint8)
    L.get_k_folds_masks.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_uint)]
    libc = C.CDLL(None)
    libc.srand(12345)
    for case in kat["k_folds"]:
        nA, nU, k = case["num_affected"], case["num_unaffected"], case["k"]
        sizes = C.POINTER(C.c_uint)()
        folds = L.get_k_folds(nA, nU, k, C.byref(sizes))
        seen = []
        for f, exp in enumerate(case["sizes"]):
            got = [sizes[3 * f], sizes[3 * f + 1], sizes[3 * f + 2]]
            assert exp is None or got == exp, (nA, nU, k, f, got)
            members = [folds[f][j] for j in range(got[0])]
            assert members == sorted(members)
            assert all(m < nA for m in members[: got[1]]) and all(m >= nA for m in members[got[1]:])
            seen += members
        assert sorted(seen) == list(range(nA + nU))                  # a partition of the cohort
        pa, pu = -(-nA // 16) * 16, -(-nU // 16) * 16
        m = L.get_k_folds_masks(nA, nU, k, folds, sizes)
        masks = np.ctypeslib.as_array(m, shape=(k, pa + pu)).copy()
        assert (masks[:, nA:pa] == 0).all() and (masks[:, pa + nU:] == 0).all()          # both paddings are 0
        real = np.concatenate([masks[:, :nA], masks[:, pa: pa + nU]], axis=1)
        assert ((real == 0).sum(axis=0) == 1).all()                  # every sample is left out of exactly one fold
        for f in range(k):
            out = set(np.flatnonzero(real[f] == 0).tolist())
            assert out == set(folds[f][j] for j in range(sizes[3 * f]))


def test_count_invariants_of_the_oracle():
    """Size-independent properties the GPU tests rely on at full size: cells partition the fully called samples;
    training counts of the folds of a partition add up to (k - 1) x the totals; a pair and its transpose."""
    rng = np.random.default_rng(77)
    nA, nU, k = 123, 211, 6
    data = epi_random_dataset(rng, 5, nA, nU, p_missing=0.1)
    fold = epi_random_folds(rng, nA, nU, k)
    masks = orc.fold_masks_from_assignment(fold, k)
    for rows in ([data[0], data[1]], [data[2], data[3], data[4]]):
        a, u = orc.epi_counts(rows, nA, nU)
        called = np.all(np.stack(rows) < 3, axis=0)
        assert a.sum() == called[:nA].sum() and u.sum() == called[nA:].sum()
        fa, fu = orc.epi_counts_all_folds(rows, nA, nU, masks)
        assert np.array_equal(fa.sum(axis=0), (k - 1) * a) and np.array_equal(fu.sum(axis=0), (k - 1) * u)
    a01, _ = orc.epi_counts([data[0], data[1]], nA, nU)
    a10, _ = orc.epi_counts([data[1], data[0]], nA, nU)
    assert np.array_equal(a01.reshape(3, 3), a10.reshape(3, 3).T)
