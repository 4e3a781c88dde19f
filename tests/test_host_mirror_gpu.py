"""GPU tests of the host mirror (include/hpgv_host.h): the reference's per-batch
functions assoc_test / tdt_test / get_variants_stats, called from a C driver
shaped like the reference's runners and unit tests, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from helpers import QUIRK_GTS, assert_close, hpgv
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    hpgv.build()
    from importlib import import_module
    b = import_module("hpg-variant_amd._build")
    exe = str(tmp_path_factory.mktemp("drv") / "host_driver")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-fopenmp", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "host_driver.c"), "-o", exe,
                           "-L", b.LIBDIR, "-lhpgv_host", "-lhpgv", "-Wl,-rpath," + b.LIBDIR, "-lm"])
    return exe


def test_reference_tdt_unit_cases_through_tdt_test(driver):
    # test/test_tdt_runner.c:93-433, same construction, HIP engine underneath
    r = subprocess.run([driver, "kat"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "KAT OK" in r.stdout and "FAIL" not in r.stdout


def _write_inputs(tmp, rng, n_fam, n_extra, n_variants, chroms=("1", "X", "22")):
    """Random pedigree + cohort as text files; returns the structures for the oracle."""
    people = []          # (fid, iid, pat, mat, sex, pheno)
    for f in range(n_fam):
        nc = int(rng.integers(1, 4))
        fa, mo = "F%dp" % f, "F%dm" % f
        people.append(("fam%d" % f, fa, "0", "0", 1, int(rng.integers(1, 3))))
        people.append(("fam%d" % f, mo, "0", "0", 2, int(rng.integers(1, 3))))
        for k in range(nc):
            people.append(("fam%d" % f, "F%dc%d" % (f, k), fa, mo, int(rng.integers(1, 3)),
                           int(rng.choice([1, 2, 2, 2, 0]))))
    for k in range(n_extra):
        people.append(("solo%d" % k, "S%d" % k, "0", "0", int(rng.integers(1, 3)), int(rng.choice([1, 2, 0]))))
    order = rng.permutation(len(people))                   # VCF column order != PED order
    names = [people[i][1] for i in order]
    with open(tmp / "ped.txt", "w") as f:
        for p in people:
            f.write("%s %s %s %s %d %d\n" % p)
    rows = []
    with open(tmp / "batch.txt", "w") as f:
        f.write("%d %d\n%s\n" % (len(names), n_variants, " ".join(names)))
        for v in range(n_variants):
            fmt = ["GT", "GT:DP", "DP:GT"][v % 3]
            gts = [QUIRK_GTS[int(i)] if rng.random() < 0.2 else ["0/0", "0/1", "1/1", "0/1"][int(rng.integers(0, 4))]
                   for i in rng.integers(0, len(QUIRK_GTS), size=len(names))]
            if fmt == "GT":
                samples = gts
            elif fmt == "GT:DP":
                samples = [g + ":12" for g in gts]
            else:
                samples = ["7:" + g for g in gts]
            chrom = chroms[v % len(chroms)]
            f.write("%s %d rs%d A C %s %s\n" % (chrom, 1000 + v, v, fmt, " ".join(samples)))
            rows.append((chrom, fmt, samples))
    return people, names, rows


def _codes(rows, strict):
    return np.array([[orc.encode_sample(s, fmt.split(":").index("GT"), strict) for s in samples]
                     for _, fmt, samples in rows], dtype=np.uint8)


def _parse_table(path):
    out = []
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        for line in f:
            out.append(line.rstrip("\n").split("\t"))
    return header, out


def _fl(x):
    return float("nan") if "nan" in x.lower() else float(x)


def test_assoc_test_runner_shape(driver, tmp_path):
    rng = np.random.default_rng(5)
    people, names, rows = _write_inputs(tmp_path, rng, 30, 40, 450)
    r = subprocess.run([driver, "assoc", str(tmp_path / "batch.txt"), str(tmp_path / "ped.txt"),
                        str(tmp_path / "out"), "4", "200"], capture_output=True, text=True)
    assert r.returncode == 0 and "ASSOC OK" in r.stdout, r.stdout + r.stderr
    pheno = {p[1]: p[5] for p in people}
    cond = np.array([{2: orc.AFFECTED, 1: orc.UNAFFECTED}.get(pheno[n], orc.COND_OTHER) for n in names], np.uint8)
    gt = _codes(rows, True)
    is_x = np.array([1 if c == "X" else 0 for c, _, _ in rows], np.uint8)
    A1, A2, U1, U2 = orc.assoc_counts(gt, cond, is_x)
    lf = orc.logfact(len(names) * 10)
    for task, ext in ((orc.TASK_CHISQ, "chisq"), (orc.TASK_FISHER, "fisher")):
        odds, chisq, p = orc.assoc_stats(task, A1, A2, U1, U2, lf)
        path = str(tmp_path / "out") + "." + ext
        gnu = subprocess.run(["sort", "-k1,1h", "-k2,2n", path], capture_output=True, text=True,
                             env=dict(os.environ, LC_ALL="C"), check=True).stdout
        assert open(path).read() == gnu                    # already in the order the reference's `sort` call gives
        header, table = _parse_table(path)
        exp_header = "#CHR POS ID A1 C_A1 C_U1 F_A1 F_U1 A2 C_A2 C_U2 F_A2 F_U2 OR".split() + \
            (["CHISQ", "P-VALUE"] if task == orc.TASK_CHISQ else ["P-VALUE"])
        assert header == exp_header                        # assoc_runner.c:295,297
        assert len(table) == len(rows)
        by_pos = {int(t[1]): t for t in table}             # results arrive unordered (assoc_runner.c:255-258)
        for v in range(len(rows)):
            t = by_pos[1000 + v]
            assert t[0] == rows[v][0] and t[2] == "rs%d" % v and t[3] == "A" and t[8] == "C"
            assert (int(t[4]), int(t[9]), int(t[5]), int(t[10])) == (A1[v], A2[v], U1[v], U2[v])
            na, nu = A1[v] + A2[v], U1[v] + U2[v]
            assert t[6] == "%6f" % (A1[v] / na if na else 0.0) and t[7] == "%6f" % (U1[v] / nu if nu else 0.0)
            # statistics are printed with %6f (6 decimals): compare at that resolution
            got = [_fl(x) for x in t[13:]]
            exp = [odds[v]] + ([chisq[v]] if task == orc.TASK_CHISQ else []) + [p[v]]
            for g, e in zip(got, exp):
                assert (np.isnan(g) and np.isnan(e)) or abs(g - e) <= 6e-7 * max(1.0, abs(e)) or (np.isinf(g) and np.isinf(e))


def _families_csr(people, names):
    col = {n: i for i, n in enumerate(names)}
    fams = {}
    for p in people:
        fams.setdefault(p[0], []).append(p)
    fcol, mcol, coff, ccol, csex = [], [], [0], [], []
    for fid in dict.fromkeys(p[0] for p in people):
        members = fams[fid]
        founders = [p for p in members if p[2] == "0" and p[3] == "0"]
        father = mother = None
        for p in founders:                                 # tdt.c:62-73
            if father and mother:
                break
            if p[4] == 1:
                father = p
            elif p[4] == 2:
                mother = p
        f, m = (col[father[1]], col[mother[1]]) if father and mother else (-1, -1)
        fcol.append(f); mcol.append(m)
        if f >= 0:
            for p in members:
                if p[2] == "0" and p[3] == "0":
                    continue
                if p[5] != 2:                              # tdt.c:144
                    continue
                ccol.append(col[p[1]])
                csex.append(orc.MALE if p[4] == 1 else orc.FEMALE)
        coff.append(len(ccol))
    return fcol, mcol, coff, ccol, csex


def test_tdt_test_runner_shape(driver, tmp_path):
    rng = np.random.default_rng(9)
    people, names, rows = _write_inputs(tmp_path, rng, 60, 10, 420)
    r = subprocess.run([driver, "tdt", str(tmp_path / "batch.txt"), str(tmp_path / "ped.txt"),
                        str(tmp_path / "out"), "4", "100"], capture_output=True, text=True)
    assert r.returncode == 0 and "TDT OK" in r.stdout, r.stdout + r.stderr
    gt = _codes(rows, True)
    is_x = np.array([1 if c == "X" else 0 for c, _, _ in rows], np.uint8)
    t1, t2 = orc.tdt_counts(gt, *_families_csr(people, names), chrom_is_x=is_x)
    odds, chisq, p = orc.tdt_stats(t1, t2)
    header, table = _parse_table(str(tmp_path / "out") + ".tdt")
    assert header == "#CHR POS ID A1 A2 T U OR CHISQ P-VALUE".split()     # tdt_runner.c:288
    assert len(table) == len(rows)
    by_pos = {int(t[1]): t for t in table}
    assert t1.sum() + t2.sum() > 0
    for v in range(len(rows)):
        t = by_pos[1000 + v]
        assert (int(t[5]), int(t[6])) == (t1[v], t2[v]), v
        for g, e in zip([_fl(x) for x in t[7:]], (odds[v], chisq[v], p[v])):
            assert (np.isnan(g) and np.isnan(e)) or (np.isinf(g) and np.isinf(e)) or abs(g - e) <= 6e-7 * max(1.0, abs(e))


@pytest.mark.parametrize("n_fam,n_extra,n_variants", [(20, 30, 200), (0, 100, 1000)])   # second: BASELINE configs[0], 1k x 100
def test_get_variants_stats_and_sample_stats(driver, tmp_path, n_fam, n_extra, n_variants):
    rng = np.random.default_rng(13)
    people, names, rows = _write_inputs(tmp_path, rng, n_fam, n_extra, n_variants)
    r = subprocess.run([driver, "stats", str(tmp_path / "batch.txt"), str(tmp_path / "stats.tsv"), str(tmp_path / "ped.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and ("STATS OK variants=%d" % n_variants) in r.stdout, r.stdout + r.stderr
    gt = _codes(rows, False)
    lines = [l.rstrip("\n").split("\t") for l in open(tmp_path / "stats.tsv")]
    vlines = [t for t in lines if t[0] == "V"]
    slines = [t for t in lines if t[0] == "S"]
    assert len(vlines) == len(rows) and len(slines) == len(names)
    n_multi = 0
    for v, t in enumerate(vlines):                         # one worker: order preserved
        na = int(t[3])
        used = [n for code in gt[v] for n in (code >> 4, code & 0xF) if n != 0xF]
        assert na == max(2, max(used) + 1 if used else 2)  # quirk genotypes reach allele 14 ("20/20" clamps)
        n_multi += na > 2
        vs = orc.variant_stats(gt[v], na)
        assert int(t[4]) == vs.missing_alleles and int(t[5]) == vs.missing_genotypes
        assert_close([_fl(t[6])], [vs.hw_chi2], "hwe chi2")
        assert_close([_fl(t[7])], [vs.hw_p], "hwe p")
        assert [int(x) for x in t[8: 8 + na]] == list(vs.alleles_count)[:na], v
        assert [int(x) for x in t[8 + na: 8 + na + na * na]] == list(vs.genotypes_count)[: na * na], v
    assert n_multi > 0 and ("multiallelic=%d" % n_multi) in r.stdout
    # per-phenotype counters: group = id of the PED variable of the column's individual ("1" -> 0, "2" -> 1, else 2);
    # a VCF column without a PED row is in no group
    pheno = {p[1]: p[5] for p in people}
    group = np.array([{1: 0, 2: 1}.get(pheno[n], 2) if n in pheno else -1 for n in names])
    plines = [t for t in lines if t[0] == "P"]
    assert len(plines) == 3 * len(rows)
    for v in range(len(rows)):
        for k in range(3):
            t = plines[3 * v + k]
            vs = orc.variant_stats(np.ascontiguousarray(gt[v][group == k]), 2)
            assert int(t[1]) == k and [int(x) for x in t[2:6]] == list(vs.genotypes_count)[:4], (v, k)
            assert (int(t[6]), int(t[7]), int(t[8]), int(t[9])) == (vs.missing_genotypes, vs.missing_alleles, vs.alleles_count[0], vs.alleles_count[1])
            assert_close([_fl(t[10])], [vs.hw_chi2], "group hwe chi2"); assert_close([_fl(t[11])], [vs.hw_p], "group hwe p")
    miss = orc.sample_missing(gt)                          # get_sample_stats: per-sample missing genotypes
    # per-sample Mendelian errors: every sample whose two parents are VCF columns, checked on every variant
    col = {n: i for i, n in enumerate(names)}
    trios = [(col[p[2]], col[p[3]], col[p[1]], orc.MALE if p[4] == 1 else orc.FEMALE) for p in people if p[2] != "0"]
    is_x = np.array([1 if c == "X" else 0 for c, _, _ in rows], np.uint8)
    _, trio_err = orc.mendel_counts(gt, [t[0] for t in trios], [t[1] for t in trios], [t[2] for t in trios],
                                    [t[3] for t in trios], is_x)
    exp_mendel = np.zeros(len(names), np.int64)
    for t, e_ in zip(trios, trio_err):
        exp_mendel[t[2]] += e_
    assert exp_mendel.sum() > 0 or not trios
    for j, t in enumerate(slines):
        assert t[1] == names[j] and int(t[2]) == miss[j] and int(t[3]) == exp_mendel[j], (j, t)


def test_staging_matches_the_oracle_encoder():
    import ctypes as C
    from importlib import import_module
    b = import_module("hpg-variant_amd._build")
    L = C.CDLL(b.HOSTLIB)
    L.get_alleles.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for s in QUIRK_GTS + ["0/1:3", "5:1|0", "10/11", "./.:.", "", "/", "1/", "/1", "0/1/2"]:
        for pos in (0, 1):
            a1, a2 = C.c_int(), C.c_int()
            st = L.get_alleles(s.encode(), pos, C.byref(a1), C.byref(a2))
            assert (st, a1.value, a2.value) == orc.get_alleles(s, pos), (s, pos)
