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
