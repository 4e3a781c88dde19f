"""CPU suite: pins the oracle (oracle/hpgv_oracle.c) against the reference's own
known-answer tests and against independent goldens (tests/golden/)."""
import math

import numpy as np
import pytest

from oracle import pyoracle as orc

SEX = {"M": orc.MALE, "F": orc.FEMALE}


# ---- reference KATs -------------------------------------------------------

def test_check_mendel_reference_kats(goldens):
    # test/test_checks_family.c:16-111
    for chrom, f1, f2, m1, m2, c1, c2, sex, op, exp in goldens["kats"]["check_mendel"]:
        got = orc.check_mendel(chrom, f1, f2, m1, m2, c1, c2, SEX[sex])
        ok = {"==": got == exp, "<=": got <= exp, ">=": got >= exp}[op]
        assert ok, (chrom, f1, f2, m1, m2, c1, c2, sex, op, exp, got)


def _tdt_kat_inputs(case):
    """Lays a KAT out as the reference test does: samples in VCF order
    father, mother, child per family (test_tdt_runner.c:103-116)."""
    samples, fcol, mcol, coff, ccol, csex = [], [], [], [0], [], []
    for fam in case["families"]:
        base = len(samples)
        samples += [fam["father"], fam["mother"], fam["child"]]
        fcol.append(base); mcol.append(base + 1)
        if fam["child_affected"]:          # tdt.c:144 only affected children count
            ccol.append(base + 2); csex.append(orc.MALE)
        coff.append(len(ccol))
    gt = orc.encode_matrix([samples])
    return gt, fcol, mcol, coff, ccol, csex


def test_tdt_reference_kats(goldens):
    # test/test_tdt_runner.c:93-433
    for case in goldens["kats"]["tdt"]:
        gt, fcol, mcol, coff, ccol, csex = _tdt_kat_inputs(case)
        t1, t2 = orc.tdt_counts(gt, fcol, mcol, coff, ccol, csex)
        assert (t1[0], t2[0]) == (case["t1"], case["t2"]), case["name"]


def test_vcf2epi_genotype_classes(goldens):
    # dataset_creator.c:255-266: the classes our code byte must preserve
    for s, exp in goldens["kats"]["genotype_codes_vcf2epi"]["cases"]:
        st, a1, a2 = orc.get_alleles(s)
        if st != 0:
            got = 255
        elif a1 == 0 and a2 == 0:
            got = 0
        elif a1 != a2:
            got = 1
        else:
            got = 2
        assert got == exp, s
        code = orc.encode_sample(s, 0, True)
        assert (code == 0xFF) == (exp == 255)


# ---- get_alleles / code ----------------------------------------------------

@pytest.mark.parametrize("s,pos,exp", [
    ("0/1", 0, (0, 0, 1)), ("1|0", 0, (0, 1, 0)), ("./.", 0, (3, -1, -1)),
    ("./1", 0, (1, -1, 1)), ("0/.", 0, (2, 0, -1)), ("1", 0, (4, 1, -1)),
    (".", 0, (3, -1, -1)), ("0/1:35:99", 0, (0, 0, 1)), ("35:1/2:9", 1, (0, 1, 2)),
    ("12/3", 0, (0, 12, 3)), ("0/1", 2, (3, -1, -1)),
])
def test_get_alleles(s, pos, exp):
    assert orc.get_alleles(s, pos) == exp


def test_code_roundtrip():
    assert orc.encode_sample("0/0") == 0x00
    assert orc.encode_sample("0/1") == 0x01
    assert orc.encode_sample("1/0") == 0x10
    assert orc.encode_sample("1|1") == 0x11
    assert orc.encode_sample("2/1") == 0x21
    assert orc.encode_sample("./.") == 0xFF
    assert orc.encode_sample("./1") == 0xFF            # strict: not ALLELES_OK -> dropped
    assert orc.encode_sample("./1", 0, False) == 0xF1   # stats keeps the called allele
    assert orc.encode_sample("1", 0, False) == 0x1F
    assert orc.encode_sample("20/3") == 0xE3            # clamp to 14
    assert orc.lib().orc_get_field_position_in_format(b"GT", b"DP:GT:GQ") == 1
    assert orc.lib().orc_get_field_position_in_format(b"GT", b"DP:GQ") == -1


# ---- assoc counting (assoc.c:87-134) ---------------------------------------

def test_assoc_count_rules_autosome_and_x():
    # hand-derived from assoc.c:94-125
    gts = ["0/0", "0/1", "1/0", "1/1", "1/2", "./.", "./1", "0/2"]
    gt = orc.encode_matrix([gts * 3])
    cond = [orc.AFFECTED] * 8 + [orc.UNAFFECTED] * 8 + [orc.COND_OTHER] * 8
    A1, A2, U1, U2 = orc.assoc_counts(gt, cond)
    # per group: 0/0 ->ref2; 0/1,1/0,0/2 -> 1+1 each; 1/1,1/2 -> alt2 each
    assert (A1[0], A2[0], U1[0], U2[0]) == (5, 7, 5, 7)
    A1, A2, U1, U2 = orc.assoc_counts(gt, cond, chrom_is_x=[1])
    # X: 0/0 -> ref+1; both non-zero -> alt+1; hets ignored
    assert (A1[0], A2[0], U1[0], U2[0]) == (1, 2, 1, 2)
    assert orc.lib().orc_chrom_is_x(b"X", 1) == 1
    assert orc.lib().orc_chrom_is_x(b"1", 1) == 0
    assert orc.lib().orc_chrom_is_x(b"X1", 2) == 0


# ---- statistics vs independent goldens --------------------------------------

def _close(got, exp, tol=1e-10):
    if exp is None:
        return math.isnan(got)
    return abs(got - exp) <= tol * max(1.0, abs(exp))


def test_chisq_and_odds_goldens(goldens):
    L = orc.lib()
    for c in goldens["stat"]["chi2"]:
        x = L.orc_assoc_basic_test(c["a"], c["b"], c["c"], c["d"])
        assert _close(x, c["chisq"], 1e-12), c
        assert _close(L.orc_chisq_p_value(x), c["p"]), c
        # (A1,A2,U1,U2) = (a,c,b,d)
        assert _close(L.orc_assoc_odds_ratio(c["a"], c["c"], c["b"], c["d"]), c["odds"], 1e-13), c
    # SURVEY 8c probe of the reference file itself
    assert L.orc_assoc_basic_test(30, 20, 10, 40) == 16.666666666666668


def test_pvalue_goldens(goldens):
    L = orc.lib()
    for c in goldens["stat"]["pvalue"]:
        assert abs(L.orc_chisq_p_value(c["x"]) - c["p"]) <= 1e-10
    assert math.isnan(L.orc_chisq_p_value(float("nan")))


def test_fisher_goldens(goldens):
    lf = orc.logfact(4000)
    assert abs(lf[10] - math.lgamma(11)) < 1e-12
    L = orc.lib()
    import ctypes as C
    p_lf = lf.ctypes.data_as(C.POINTER(C.c_double))
    for c in goldens["stat"]["fisher"]:
        got = L.orc_fisher_two_sided(c["a"], c["b"], c["c"], c["d"], p_lf)
        assert abs(got - c["p"]) <= 1e-10, c


def test_hwe_goldens(goldens):
    for c in goldens["stat"]["hwe"]:
        chi2, p = orc.hwe(c["n_AA"], c["n_Aa"], c["n_aa"])
        assert _close(chi2, c["chi2"], 1e-12), c
        assert _close(p, c["p"]), c


def test_tdt_stats_rules():
    # tdt.c:255-260,290-292
    odds, chisq, p = orc.tdt_stats([0, 3, 5, 0], [0, 0, 5, 4])
    assert chisq[0] == -1 and p[0] == 1.0 and math.isnan(odds[0])
    assert chisq[1] == 3.0 and math.isnan(odds[1])
    assert chisq[2] == 0.0 and p[2] == 1.0 and odds[2] == 1.0
    assert chisq[3] == 4.0 and odds[3] == 0.0


def test_tdt_sibling_carry_over():
    # tdt.c:128-132: trA/trB live at family scope and are not reset per child.
    # parents 0/1 x 0/1; child1 0/0 sets trA=1,trB=1 (t1+=2); child2 1/1 ... sets both.
    # parents 0/1 x 0/0: child1 0/0 -> trA=1; no trB ever.
    # Mixed: parents 0/1 x 0/1, child1 = 0/1 (trA=1,trB=2), then with mother hom the
    # second branch cannot occur in one family, so exercise stickiness through a
    # Mendel-skipped child between two counted ones.
    gt = orc.encode_matrix([["0/1", "0/1", "0/1", "./.", "0/0"]])
    t1, t2 = orc.tdt_counts(gt, [0], [1], [0, 3], [2, 3, 4], [0, 0, 0])
    # child 0/1: t1+1 (trA=1), t2+1 (trB=2); child ./. skipped; child 0/0: trA=1,trB=1 -> t1+2
    assert (t1[0], t2[0]) == (3, 1)
    # parents 0/1 x 1/1: child 0/1 -> trA=1 only (trB stays 0);
    gt = orc.encode_matrix([["0/1", "1/1", "0/1", "1/1"]])
    t1, t2 = orc.tdt_counts(gt, [0], [1], [0, 2], [2, 3], [0, 0])
    # child 0/1: het dad, hom mum non-zero-first -> trA=1 ; child 1/1: trA=2
    assert (t1[0], t2[0]) == (1, 1)


def test_variant_stats_counts():
    row = orc.encode_matrix([["0/0", "0/1", "1/0", "1/1", "./.", "./1", "1/2", "2/2"]], strict=False)[0]
    vs = orc.variant_stats(row, 3)
    assert list(vs.alleles_count)[:3] == [4, 6, 3]
    assert vs.missing_alleles == 3 and vs.missing_genotypes == 2
    g = list(vs.genotypes_count)[:9]
    assert g == [1, 1, 0, 1, 1, 1, 0, 0, 1]
    assert (vs.hw_n_AA, vs.hw_n_Aa, vs.hw_n_aa) == (1, 2, 1)


# ---- synthetic generator ----------------------------------------------------

def test_synth_generator_properties():
    assert orc.lib().orc_splitmix64(0) == 0
    thr = orc.synth_thresholds(12345)
    assert thr[0] == 167772 and thr[0] < thr[1] < thr[2] <= (1 << 24)
    gt = orc.synth_matrix(0, 64, 4000, 4000)
    vals, counts = np.unique(gt, return_counts=True)
    assert set(vals.tolist()) <= {0x00, 0x01, 0x11, 0xFF}
    miss = counts[vals.tolist().index(0xFF)] / gt.size
    assert 0.007 < miss < 0.013
    # determinism and row independence of the starting variant
    gt2 = orc.synth_matrix(10, 8, 4000, 4000)
    assert np.array_equal(gt[10:18], gt2)


def test_text_and_packed_paths_agree():
    import ctypes as C
    gt = orc.synth_matrix(0, 20, 101, 101)
    cond = (np.arange(101) % 2).astype(np.uint8)
    A = orc.assoc_counts(gt, cond)
    strs = []
    for row in gt:
        for g in row:
            strs.append(b"./." if g == 0xFF else b"%d/%d" % (g >> 4, g & 0xF))
    arr = (C.c_char_p * len(strs))(*strs)
    fm = (C.c_char_p * 20)(*([b"GT"] * 20))
    outs = [np.zeros(20, dtype=np.int32) for _ in range(4)]
    L = orc.lib()
    L.orc_assoc_text.restype = None
    L.orc_assoc_text(arr, 20, 101, fm, cond.ctypes.data_as(C.POINTER(C.c_uint8)), None,
                     *[o.ctypes.data_as(C.POINTER(C.c_int32)) for o in outs])
    for a, b in zip(A, outs):
        assert np.array_equal(a, b)


def test_alleles_above_14_keep_their_inequality():
    """the packed code stores allele indices above 14 as 14; "15/16" must stay heterozygous, because the reference
    compares the raw ints (tdt.c:113 skips a family whose parents are both homozygous, :185-187 tests a1 != a2)."""
    assert orc.encode_sample("15/16") == 0xDE and orc.encode_sample("16/15") == 0xDE
    assert orc.encode_sample("15/15") == 0xEE and orc.encode_sample("14/20") == 0xDE and orc.encode_sample("3/20") == 0x3E
    # father 15/16 (het), mother 0/0, affected child 0/15: counted on the sample strings and on the packed codes alike
    rows = [["15/16", "0/0", "0/15"], ["15/15", "0/0", "0/15"], ["0/16", "16/15", "15/16"]]
    fam = (np.array([0]), np.array([1]), np.array([0, 1]), np.array([2]), np.array([0], np.uint8))
    t_text = orc.tdt_text(rows, *fam)
    t_code = orc.tdt_counts(orc.encode_matrix(rows), *fam)
    assert np.array_equal(t_text[0], t_code[0]) and np.array_equal(t_text[1], t_code[1])
    assert t_text[0][0] + t_text[1][0] == 1 and t_text[0][1] + t_text[1][1] == 0      # het father counted, hom father skipped
