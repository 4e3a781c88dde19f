"""The data files of the reference's own TDT integration test
(test/test_tdt_runner.c:436-498: 4000 SNPs x 147 samples = 49 trios), through
the oracle, an independent textbook TDT, and (gpu) the HIP path from raw text."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fx():
    text = gzip.open(os.path.join(HERE, "tdt_4k_147.vcf.gz")).read().decode()
    lines = text.splitlines()
    names = lines[0].split("\t")[9:]
    data = "\n".join(lines[1:]) + "\n"
    people = []
    for line in open(os.path.join(HERE, "tdt_4k_147.ped")):
        f = line.split()
        people.append((f[0], f[1], f[2], f[3], int(f[4]), int(f[5])))
    exp = json.load(open(os.path.join(HERE, "tdt_4k_147.json")))
    col = {n: i for i, n in enumerate(names)}
    trios = [(col[p[2]], col[p[3]], col[p[1]], p[4]) for p in people if p[2] != "0" and p[5] == 2]
    return dict(data=data, names=names, people=people, exp=exp, trios=trios)


def _csr(trios):
    f = [t[0] for t in trios]; m = [t[1] for t in trios]; c = [t[2] for t in trios]
    sex = [orc.MALE if t[3] == 1 else orc.FEMALE for t in trios]
    return f, m, list(range(len(trios) + 1)), c, sex


def _textbook_tdt(gt, trios):
    """Spielman TDT on biallelic calls, written from the genetics, not from tdt.c: every
    heterozygous parent of an affected child transmits one allele; count ref / alt."""
    n_alt = {0x00: 0, 0x01: 1, 0x10: 1, 0x11: 2}
    t1 = np.zeros(gt.shape[0], np.int64); t2 = np.zeros(gt.shape[0], np.int64)
    for v in range(gt.shape[0]):
        row = gt[v]
        for f, m, c, _ in trios:
            if row[f] not in n_alt or row[m] not in n_alt or row[c] not in n_alt:
                continue
            af, am, ac = n_alt[row[f]], n_alt[row[m]], n_alt[row[c]]
            # Mendelian consistency: child alt count must be reachable
            reach = {x + y for x in ({0} if af == 0 else {1} if af == 2 else {0, 1})
                     for y in ({0} if am == 0 else {1} if am == 2 else {0, 1})}
            if ac not in reach or (af != 1 and am != 1):
                continue
            if af == 1 and am == 1:
                alt_tx = ac                                  # two het parents: ac alt alleles transmitted in total
                t2[v] += alt_tx; t1[v] += 2 - alt_tx
            else:
                hom = am if af == 1 else af                  # the homozygous parent gives hom/2 alt alleles
                alt_tx = ac - hom // 2
                t2[v] += alt_tx; t1[v] += 1 - alt_tx
    return t1, t2


def test_oracle_on_reference_fixture_matches_textbook_tdt(fx):
    tok = orc.tokenize(fx["data"], len(fx["names"]), True)
    assert tok["n_lines"] == fx["exp"]["n_variants"] == 4000 and not tok["status"].any()
    hist = {"%02x" % k: int(v) for k, v in zip(*np.unique(tok["gt"], return_counts=True))}
    assert hist == fx["exp"]["genotype_histogram"]           # SURVEY section 4 fixture facts
    assert len(fx["trios"]) == 49
    t1, t2 = orc.tdt_counts(tok["gt"], *_csr(fx["trios"]), chrom_is_x=tok["is_x"])
    assert t1.tolist() == fx["exp"]["t1"] and t2.tolist() == fx["exp"]["t2"]
    b1, b2 = _textbook_tdt(tok["gt"], fx["trios"])           # independent derivation
    assert np.array_equal(t1, b1) and np.array_equal(t2, b2)


@pytest.mark.gpu
def test_hip_path_from_raw_text_on_reference_fixture(fx):
    from helpers import assert_close, hpgv
    e = hpgv.Engine(0)
    n = len(fx["names"])
    got = e.tokenize(fx["data"], n, True)
    assert np.array_equal(got["gt"], orc.tokenize(fx["data"], n, True)["gt"])
    e.set_families(n, *_csr(fx["trios"]))
    res = e.tdt_text(fx["data"])
    assert res["n_lines"] == 4000
    assert res["t1"].tolist() == fx["exp"]["t1"] and res["t2"].tolist() == fx["exp"]["t2"]
    odds, chisq, p = orc.tdt_stats(res["t1"], res["t2"])
    assert_close(res["chisq"], chisq, "chisq"); assert_close(res["p"], p, "p"); assert_close(res["odds"], odds, "odds")
    e.close()


@pytest.mark.gpu
def test_c_example_from_vcf_text_to_tsv(fx, tmp_path):
    """examples/assoc_from_text.c: plain C over include/hpgv.h, VCF file in, the reference's .chisq TSV out."""
    import subprocess
    from importlib import import_module
    from helpers import hpgv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = import_module("hpg-variant_amd._build")
    hpgv.build()
    exe = str(tmp_path / "assoc_from_text")
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "assoc_from_text.c"), "-o", exe,
                           "-L", b.LIBDIR, "-lhpgv", "-Wl,-rpath," + b.LIBDIR, "-lm"])
    vcf = tmp_path / "in.vcf"
    vcf.write_bytes(b"##fileformat=VCFv4.1\n" + gzip.open(os.path.join(HERE, "tdt_4k_147.vcf.gz")).read())
    pheno = {p[1]: p[5] for p in fx["people"]}
    cond = [1 if pheno[n] == 2 else 0 if pheno[n] == 1 else 2 for n in fx["names"]]
    (tmp_path / "cond.txt").write_text(" ".join(map(str, cond)))
    out = subprocess.run([exe, str(vcf), str(tmp_path / "cond.txt")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    assert lines[0].startswith("#CHR\tPOS\tID\tA1") and len(lines) == 4001
    tok = orc.tokenize(fx["data"], len(fx["names"]), True)
    A1, A2, U1, U2 = orc.assoc_counts(tok["gt"], np.array(cond, np.uint8), tok["is_x"])
    odds, chisq, p = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
    data = fx["data"].split("\n")
    for i in (0, 1, 2, 1999, 3999):
        t = lines[1 + i].split("\t")
        ref = data[i].split("\t")
        assert t[0:3] == ref[0:3] and t[3] == ref[3] and t[8] == ref[4]
        assert (int(t[4]), int(t[9]), int(t[5]), int(t[10])) == (A1[i], A2[i], U1[i], U2[i])
        for got, exp in zip(t[13:], (odds[i], chisq[i], p[i])):
            g = float("nan") if "nan" in got else float(got)
            assert (np.isnan(g) and np.isnan(exp)) or abs(g - exp) <= 6e-7 * max(1.0, abs(exp))
