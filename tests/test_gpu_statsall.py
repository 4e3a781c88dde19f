"""k_stats_all2 (hpgv_statsall_kernels.h): every statistic of the stats tool from one read of the tokenizer's matrix, columns
owned by threads across a band of rows.  Through hpgv_stats_text_groups on shapes that take every form of the kernel (one to
four 16-byte chunks per thread, workgroups of 64 to 512 threads, no / some / all samples in phenotype groups, trios on and
off, chromosome X rows), against the row-staging kernel k_stats_all (HPGV_STATS_ALL2=0: byte-identical) and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import hpgv
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def _text(codes, chroms):
    a1, a2 = codes >> 4, codes & 15
    names = np.array([str(i) for i in range(15)] + ["."])
    cells = np.char.add(np.char.add(names[a1], "/"), names[a2])
    lines = []
    for v in range(codes.shape[0]):
        lines.append("%s\t%d\trs%d\tA\tC,G,T\t.\tPASS\t.\tGT\t%s\n" % (chroms[v], 100 + v, v, "\t".join(cells[v])))
    return "".join(lines).encode()


def _call(e, text, m, n_samples, n_trios, n_groups, want):
    L = e.L
    nl, nm = C.c_int(0), C.c_int(m)
    out = dict(line_off=np.zeros(m + 2, np.uint64), field_off=np.zeros(m * 10, np.uint32), status=np.zeros(m, np.int32),
               c8=np.zeros(m * 8, np.int32), hw=np.zeros(2 * m), smiss=np.zeros(n_samples, np.int32), cerr=np.zeros(max(n_trios, 1), np.int32),
               merr=np.zeros(m, np.int32), midx=np.zeros(m, np.int32), mtab=np.zeros(256 * m, np.int32),
               gc8=np.zeros(max(n_groups, 1) * m * 8, np.int32), ghw=np.zeros(2 * max(n_groups, 1) * m))
    p = lambda a: C.c_void_p(a.ctypes.data)
    g = n_groups > 0 and "groups" in want
    rc = L.hpgv_stats_text_groups(e.h, text, len(text), m, C.byref(nl), p(out["line_off"]), p(out["field_off"]), p(out["status"]),
                                  p(out["c8"]), p(out["hw"]), p(out["hw"][m:]), p(out["smiss"]) if "sm" in want else None,
                                  p(out["midx"]), p(out["mtab"]), C.byref(nm), p(out["merr"]) if "me" in want else None,
                                  p(out["cerr"]) if "ce" in want else None, p(out["gc8"]) if g else None, p(out["ghw"]) if g else None,
                                  p(out["ghw"][max(n_groups, 1) * m:]) if g else None)
    assert rc == 0, L.hpgv_last_error(e.h)
    assert nl.value == m
    return out


@pytest.mark.parametrize("n_samples,n_groups,all_grouped,want", [
    (70, 0, False, ("sm",)),                               # one chunk per thread, one wave
    (1000, 2, True, ("sm", "me", "ce", "groups")),         # 63 chunks; the last group derived
    (4100, 3, False, ("sm", "me", "ce", "groups")),        # 257 chunks: 1 x 320 threads; three masked groups
    (5000, 4, True, ("groups", "me")),                     # three masked + one derived
    (8200, 1, False, ("sm", "groups")),                    # 513 chunks: 2 x 320
    (10000, 3, True, ("sm", "me", "ce", "groups")),        # 625 chunks: 2 x 320 (the bench shape)
    (12300, 2, False, ("sm", "ce", "groups")),             # 769 chunks: 2 x 448
    (16384, 2, True, ("sm", "me", "ce", "groups")),        # the widest it takes: 1024 chunks, 2 x 512
    (16400, 2, True, ("sm", "me", "groups")),              # wider: the row-staging kernel
    (3000, 6, True, ("sm", "me", "groups")),               # more groups than it masks: the row-staging kernel
])
def test_stats_all2_equals_row_staging_kernel_and_oracle(n_samples, n_groups, all_grouped, want):
    rng = np.random.default_rng(n_samples + n_groups)
    m = 61
    codes = rng.choice(np.array([0x00, 0x01, 0x10, 0x11, 0xFF, 0x0F, 0xF1, 0x12, 0x22, 0x2F, 0x5E], np.uint8), size=(m, n_samples),
                       p=[0.4, 0.2, 0.1, 0.2, 0.03, 0.01, 0.01, 0.02, 0.01, 0.01, 0.01])
    chroms = np.where(rng.random(m) < 0.3, "X", "7")
    text = _text(codes, chroms)
    n_trios = min(n_samples // 3, 4000)
    cols = rng.permutation(n_samples)
    f, mo, c = cols[:n_trios], cols[n_trios: 2 * n_trios], cols[2 * n_trios: 3 * n_trios]
    sex = rng.integers(0, 2, n_trios).astype(np.uint8)
    groups = rng.integers(0 if all_grouped else -1, max(n_groups, 1), n_samples).astype(np.int32)
    res = {}
    for all2 in ("1", "0"):
        os.environ["HPGV_STATS_ALL2"] = all2
        try:
            e = hpgv.Engine(0)
            e.set_stats_cohort(n_samples)
            e.set_pedigree(n_samples, f, mo, c, sex)
            if n_groups:
                e.set_stats_groups(groups, n_groups)
            first = _call(e, text, m, n_samples, n_trios, n_groups, want)
            again = _call(e, text, m, n_samples, n_trios, n_groups, want)          # accumulating outputs: a second call adds
            res[all2] = (first, again)
            e.close()
        finally:
            del os.environ["HPGV_STATS_ALL2"]
    a, a2 = res["1"]
    b, _ = res["0"]
    for k in ("c8", "hw", "smiss", "cerr", "merr", "gc8", "ghw", "status"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
        assert np.array_equal(a[k], a2[k], equal_nan=True), k                      # fresh output arrays per call: same again
    # the oracle
    is_x = (chroms == "X").astype(np.uint8)
    if "sm" in want:
        assert np.array_equal(a["smiss"], orc.sample_missing(codes))
    if "me" in want or "ce" in want:
        exp_err, exp_trio = orc.mendel_counts(codes, f, mo, c, sex, is_x)
        if "me" in want:
            assert np.array_equal(a["merr"], exp_err)
        if "ce" in want:
            assert np.array_equal(a["cerr"][:n_trios], exp_trio)
    c8 = a["c8"].reshape(m, 8)
    for v in range(0, m, 5):
        vs = orc.variant_stats(np.ascontiguousarray(codes[v]), 4)
        assert c8[v, 4] == vs.missing_genotypes and c8[v, 5] == vs.missing_alleles
        assert c8[v, 0] == vs.genotypes_count[0] and c8[v, 6] == vs.alleles_count[0] and c8[v, 7] == vs.alleles_count[1]
        if n_groups and "groups" in want:
            gc8 = a["gc8"].reshape(n_groups, m, 8)
            for gi in range(n_groups):
                gs = orc.variant_stats(np.ascontiguousarray(codes[v][groups == gi]), 4)
                assert gc8[gi, v, 4] == gs.missing_genotypes and gc8[gi, v, 0] == gs.genotypes_count[0]
                assert gc8[gi, v, 6] == gs.alleles_count[0] and gc8[gi, v, 7] == gs.alleles_count[1]


def test_every_code_byte_through_the_text_entry_points():
    # every HPGV8 code (alleles 0 .. 14 and missing, in both positions) as VCF text -- "13/.", "./7", "14/14" ... -- through the
    # tokenizer and the kernels that read its raw rows: k_stats_all2 (stats), k_assoc_rows (assoc), k_batch (tdt), against the oracle
    from helpers import check_assoc, make_families, oracle_assoc
    rng = np.random.default_rng(4096)
    n_samples, m = 1000, 48
    codes = rng.integers(0, 256, size=(m, n_samples), dtype=np.uint8)
    codes[:16] = np.arange(256, dtype=np.uint8).reshape(16, 16).repeat(63, axis=1)[:, :n_samples]
    chroms = np.where(rng.random(m) < 0.3, "X", "7")
    is_x = (chroms == "X").astype(np.uint8)
    text = _text(codes, chroms)
    # stats: everything on, two groups (the second derived)
    n_trios = 300
    cols = rng.permutation(n_samples)
    f, mo, c = cols[:n_trios], cols[n_trios: 2 * n_trios], cols[2 * n_trios: 3 * n_trios]
    sex = rng.integers(0, 2, n_trios).astype(np.uint8)
    groups = rng.integers(0, 2, n_samples).astype(np.int32)
    e = hpgv.Engine(0)
    e.set_stats_cohort(n_samples)
    e.set_pedigree(n_samples, f, mo, c, sex)
    e.set_stats_groups(groups, 2)
    a = _call(e, text, m, n_samples, n_trios, 2, ("sm", "me", "ce", "groups"))
    e.close()
    assert np.array_equal(a["smiss"], orc.sample_missing(codes))
    exp_err, exp_trio = orc.mendel_counts(codes, f, mo, c, sex, is_x)
    assert np.array_equal(a["merr"], exp_err) and np.array_equal(a["cerr"][:n_trios], exp_trio)
    c8, gc8 = a["c8"].reshape(m, 8), a["gc8"].reshape(2, m, 8)
    for v in range(m):
        vs = orc.variant_stats(np.ascontiguousarray(codes[v]), 15)
        assert c8[v, 4] == vs.missing_genotypes and c8[v, 5] == vs.missing_alleles, v
        assert c8[v, 0] == vs.genotypes_count[0] and c8[v, 6] == vs.alleles_count[0] and c8[v, 7] == vs.alleles_count[1], v
        for gi in range(2):
            gs = orc.variant_stats(np.ascontiguousarray(codes[v][groups == gi]), 15)
            assert gc8[gi, v, 4] == gs.missing_genotypes and gc8[gi, v, 0] == gs.genotypes_count[0] and gc8[gi, v, 7] == gs.alleles_count[1], (v, gi)
    # assoc and tdt on the same text (strict: a half-called genotype is a missing one)
    strict = np.where(((codes >> 4) == 15) | ((codes & 15) == 15), 0xFF, codes).astype(np.uint8)
    cond = rng.choice([0, 1, 2], size=n_samples, p=[0.45, 0.45, 0.1]).astype(np.uint8)
    e = hpgv.Engine(0)
    e.set_cohort(cond)
    check_assoc(e.assoc_text(hpgv.TASK_CHISQ, text), oracle_assoc(orc.TASK_CHISQ, strict, cond, is_x, None), hpgv.TASK_CHISQ)
    e.close()
    fam = make_families(rng, n_samples, 250, 3, p_absent=0.03)
    e = hpgv.Engine(0)
    e.set_families(n_samples, *fam)
    res = e.tdt_text(text)
    t1, t2 = orc.tdt_counts(strict, *fam, chrom_is_x=is_x)
    assert np.array_equal(res["t1"], t1) and np.array_equal(res["t2"], t2)
    e.close()


def test_random_shapes():
    # cohort widths, group counts and output selections drawn at random (HPGV_SOAK_SHAPES of them: 6 in the suite, hundreds in a
    # soak run): the column-owning kernels pick chunks per thread and workgroup sizes from the width, and a width next to any of
    # their boundaries must do
    rng = np.random.default_rng(int(os.environ.get("HPGV_FUZZ_SEED", "77")))
    outs = ("sm", "me", "ce", "groups")
    for _ in range(int(os.environ.get("HPGV_SOAK_SHAPES", "6"))):
        n_samples = int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(200, 4200)), int(rng.integers(4200, 16500)), 16 * int(rng.integers(1, 1030))]))
        n_groups = int(rng.integers(0, 5))
        want = tuple(o for o in outs if rng.random() < 0.7) or ("sm",)
        if n_samples < 3:
            want = tuple(o for o in want if o not in ("me", "ce")) or ("sm",)
        test_stats_all2_equals_row_staging_kernel_and_oracle(n_samples, n_groups, bool(rng.integers(0, 2)), want)
