"""Shared helpers for the parity tests (HIP path through the C ABI vs the oracle)."""
import importlib

import numpy as np

from oracle import pyoracle as orc

hpgv = importlib.import_module("hpg-variant_amd")

def shipped(key, values):
    """The values of option `key` this build of libhpgv.so takes.  The shipped library holds ONE form of each kernel; the forms
    that lost their A/B live in an ablation build (tools/build_ablation.py, HPGV_LIB=...), where these tests cover them too."""
    e = hpgv.Engine(0)
    ok = []
    for v in values:
        try:
            e.set_option(key, v)
            ok.append(v)
        except hpgv.HpgvError:
            pass
    e.close()
    return ok


def set_or_skip(engine, key, value):
    """set_option, or skip the test when the value names a form only an ablation build holds"""
    import pytest
    try:
        engine.set_option(key, value)
    except hpgv.HpgvError as err:
        if "ablation" in str(err):
            pytest.skip("%s = %s is an ablation build's form" % (key, value))
        raise


TOL = 1e-10   # north_star: chi2 / Fisher / HWE p-values within 1e-10 of the reference path


def close(got, exp, tol=TOL):
    """|d| <= tol * max(1, |exp|), NaN == NaN, inf == inf."""
    got = np.asarray(got, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    both_nan = np.isnan(got) & np.isnan(exp)
    same_inf = np.isinf(got) & np.isinf(exp) & (np.sign(got) == np.sign(exp))
    with np.errstate(invalid="ignore"):
        ok = np.abs(got - exp) <= tol * np.maximum(1.0, np.abs(exp))
    return bool(np.all(ok | both_nan | same_inf))


def assert_close(got, exp, what="", tol=TOL):
    if not close(got, exp, tol):
        got = np.asarray(got); exp = np.asarray(exp)
        with np.errstate(invalid="ignore"):
            bad = ~((np.abs(got - exp) <= tol * np.maximum(1.0, np.abs(exp))) |
                    (np.isnan(got) & np.isnan(exp)) | (np.isinf(got) & np.isinf(exp)))
        idx = np.flatnonzero(bad)[:5]
        raise AssertionError("%s: %d mismatches, first at %s: got %s expected %s" %
                             (what, bad.sum(), idx, got[idx], exp[idx]))


P_REL = 1e-9   # relative agreement of p-values wherever a relative statement means something (p >= 1e-280)


def assert_p_close(got, exp, what="p"):
    """p-values: north_star's absolute 1e-10 AND, above 1e-280, a relative 1e-9.  The absolute bound alone says nothing about
    a p-value below 1e-10 (every genome-wide hit).  The reference itself computes 1 - gsl_cdf_chisq_P(x, 1)
    (assoc_basic_test.c:61, tdt.c:292), which cannot tell p = 1e-17 from 0: below ~1e-16 its digits are rounding noise, so
    there the contract against the REFERENCE is the absolute one; the relative check pins the engine to the oracle's
    erfc(sqrt(x / 2)), which keeps its digits all the way down."""
    assert_close(got, exp, what)
    got = np.asarray(got, dtype=np.float64); exp = np.asarray(exp, dtype=np.float64)
    sel = np.isfinite(exp) & (exp >= 1e-280)
    with np.errstate(invalid="ignore"):
        bad = sel & ~(np.abs(got - exp) <= P_REL * exp)
    if bad.any():
        idx = np.flatnonzero(bad)[:5]
        raise AssertionError("%s: %d p-values off by more than %g relative, first at %s: got %s expected %s" %
                             (what, bad.sum(), P_REL, idx, got[idx], exp[idx]))


# genotype strings covering every branch of assoc.c:94-125 and tdt.c:103-213
QUIRK_GTS = ["0/0", "0/1", "1/0", "1/1", "1/2", "2/1", "0/2", "2/0", "./.", "./1", "0/.", "1|0", "0|1",
             "1", ".", "3/3", "14/0", "20/20", "15/16", "0/17"]


def random_codes(rng, n_variants, n_samples, quirks=True, strict=True, p_missing=0.05):
    """Random HPGV8 matrix in VCF column order."""
    if quirks:
        table = np.array([orc.encode_sample(s, 0, strict) for s in QUIRK_GTS], dtype=np.uint8)
        w = np.ones(len(table)); w[:4] = 6.0
        idx = rng.choice(len(table), size=(n_variants, n_samples), p=w / w.sum())
        return table[idx]
    base = np.array([0x00, 0x01, 0x11, 0xFF], dtype=np.uint8)
    p = np.array([0.45, 0.35, 0.2 - p_missing, p_missing])
    return base[rng.choice(4, size=(n_variants, n_samples), p=p)]


def oracle_assoc(task, gt, condition, is_x=None, lf=None):
    A1, A2, U1, U2 = orc.assoc_counts(gt, condition, is_x)
    odds, chisq, p = orc.assoc_stats(task, A1, A2, U1, U2, lf)
    return dict(A1=A1, A2=A2, U1=U1, U2=U2, odds=odds, chisq=chisq, p=p)


def check_assoc(res, exp, task):
    for k in ("A1", "A2", "U1", "U2"):
        assert np.array_equal(res[k], exp[k]), "%s differs (bit-exact required)" % k
    assert_close(res["odds"], exp["odds"], "odds")
    if task == hpgv.TASK_CHISQ:
        assert_close(res["chisq"], exp["chisq"], "chisq")
    assert_close(res["p"], exp["p"], "p")


def make_families(rng, n_samples, n_families, max_children=1, p_absent=0.0):
    """Random pedigree over distinct VCF columns -> CSR arrays of hpgv_set_families."""
    cols = rng.permutation(n_samples)
    pos = 0
    fcol, mcol, coff, ccol, csex = [], [], [0], [], []
    for _ in range(n_families):
        nc = int(rng.integers(1, max_children + 1)) if max_children > 1 else 1
        if rng.random() < 0.1 and max_children > 1:
            nc = 0
        need = 2 + nc
        if pos + need > n_samples:
            break
        f, m = int(cols[pos]), int(cols[pos + 1])
        if rng.random() < p_absent:
            f = -1
        fcol.append(f); mcol.append(m)
        for k in range(nc):
            ccol.append(int(cols[pos + 2 + k]))
            csex.append(int(rng.integers(0, 2)))
        coff.append(len(ccol))
        pos += need
    return (np.array(fcol, np.int32), np.array(mcol, np.int32), np.array(coff, np.int32),
            np.array(ccol, np.int32), np.array(csex, np.uint8))


# ---- epistasis / MDR ---------------------------------------------------------------------------

def epi_unpad(padded, n_affected, n_unaffected):
    """The reference's SSE layout [affected | pad to 16 | unaffected | pad to 16] -> [affected | unaffected]."""
    a = np.asarray(padded, dtype=np.uint8)
    pa = -(-n_affected // 16) * 16
    return np.concatenate([a[:n_affected], a[pa: pa + n_unaffected]])


def epi_random_dataset(rng, n_variants, n_affected, n_unaffected, p_missing=0.02):
    """vcf2epi rows: codes 0/1/2, 255 = missing, cases first."""
    codes = np.array([0, 1, 2, 255], np.uint8)
    p = np.array([0.5, 0.35, 0.15 - p_missing, p_missing])
    return codes[rng.choice(4, size=(n_variants, n_affected + n_unaffected), p=p)]


def epi_random_folds(rng, n_affected, n_unaffected, num_folds):
    """Fold of every sample as get_k_folds deals them out (cross_validation.c:16-100): shuffled cases and
    shuffled controls handed round-robin to the folds."""
    f = np.empty(n_affected + n_unaffected, np.int32)
    f[rng.permutation(n_affected)] = np.arange(n_affected) % num_folds
    f[n_affected + rng.permutation(n_unaffected)] = np.arange(n_unaffected) % num_folds
    return f


def epi_counts_from_reference_masks(kat_masks, num_affected, num_unaffected, order):
    """combination_counts (model.c:76-124) on the reference's own expected mask arrays (test_get_masks): per SNP
    [genotype 0 | 1 | 2] x 32 padded samples of 0 / 255 bytes.  For every cell (the last SNP's genotype varying fastest)
    AND the SNPs' masks and count the set bytes (the reference counts bits and divides by 8), cases and controls apart."""
    m = np.array(kat_masks, np.uint8).reshape(order, 3, 32)
    pad_a = (num_affected + 15) // 16 * 16
    aff, unaff = [], []
    for c in range(3 ** order):
        digits = [(c // 3 ** (order - 1 - s)) % 3 for s in range(order)]
        x = np.full(32, 255, np.uint8)
        for s, g in enumerate(digits):
            x &= m[s][g]
        aff.append(int(np.unpackbits(x[:num_affected]).sum() // 8))
        unaff.append(int(np.unpackbits(x[pad_a: pad_a + num_unaffected]).sum() // 8))
    return aff, unaff
