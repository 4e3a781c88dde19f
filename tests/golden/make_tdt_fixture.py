#!/usr/bin/env python3
"""Builds tests/golden/tdt_4k_147.* from the data files the reference's own TDT
integration test uses (test/test_tdt_runner.c:436-498 runs hpg-var-gwas tdt on
test/tdt_files/4K_variants_147_samples.{vcf,ped}).  Runs only where
/root/reference exists; the outputs are committed:

  tdt_4k_147.vcf.gz   the VCF data lines (with the #CHROM line), gzip
  tdt_4k_147.ped      the PED file (49 trios)
  tdt_4k_147.json     T/U per variant from the ORACLE on these inputs

The reference's expected output for this test (PLINK's plink.tdt) is not in the
reference checkout (.MISSING_LARGE_BLOBS), so the T/U table is a regression
pin of the oracle, not an independent golden.
"""
import gzip
import io
import json
import os
import sys
import tarfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as orc  # noqa: E402

SRC = "/root/reference/test/tdt_files/4K_variants_147_samples.tar.gz"


def load_ped(text):
    people = []
    for line in text.splitlines():
        f = line.split()
        if len(f) >= 6:
            people.append((f[0], f[1], f[2], f[3], int(f[4]), int(f[5])))
    return people


def families_csr(people, names):
    """CSR pedigree the way tdt.c:56-95,135-148 walks it (families in order of first appearance)."""
    col = {n: i for i, n in enumerate(names)}
    fams = {}
    for p in people:
        fams.setdefault(p[0], []).append(p)
    fcol, mcol, coff, ccol, csex = [], [], [0], [], []
    for fid in dict.fromkeys(p[0] for p in people):
        members = fams[fid]
        father = mother = None
        for p in members:
            if p[2] == "0" and p[3] == "0":
                if father and mother:
                    break
                if p[4] == 1:
                    father = p
                elif p[4] == 2:
                    mother = p
        ok = father and mother and father[1] in col and mother[1] in col
        fcol.append(col[father[1]] if ok else -1)
        mcol.append(col[mother[1]] if ok else -1)
        if ok:
            for p in members:
                if (p[2] == "0" and p[3] == "0") or p[5] != 2 or p[1] not in col:
                    continue
                ccol.append(col[p[1]])
                csex.append(orc.MALE if p[4] == 1 else orc.FEMALE)
        coff.append(len(ccol))
    return fcol, mcol, coff, ccol, csex


def main():
    tf = tarfile.open(SRC)
    vcf = tf.extractfile("4K_variants_147_samples.vcf").read().decode()
    ped = tf.extractfile("4K_variants_147_samples.ped").read().decode()
    lines = [l for l in vcf.splitlines() if not l.startswith("##")]
    header, data = lines[0], lines[1:]
    names = header.split("\t")[9:]
    with gzip.GzipFile(os.path.join(HERE, "tdt_4k_147.vcf.gz"), "wb", mtime=0) as g:
        g.write(("\n".join([header] + data) + "\n").encode())
    with open(os.path.join(HERE, "tdt_4k_147.ped"), "w") as f:
        f.write(ped)
    tok = orc.tokenize("\n".join(data) + "\n", len(names), True)
    assert tok["n_lines"] == len(data) and not tok["status"].any()
    people = load_ped(ped)
    fam = families_csr(people, names)
    t1, t2 = orc.tdt_counts(tok["gt"], *fam, chrom_is_x=tok["is_x"])
    out = {"n_variants": len(data), "n_samples": len(names), "n_families": len(fam[0]),
           "n_counted_children": len(fam[3]), "t1": t1.tolist(), "t2": t2.tolist(),
           "genotype_histogram": {"%02x" % k: int(v) for k, v in zip(*np.unique(tok["gt"], return_counts=True))}}
    with open(os.path.join(HERE, "tdt_4k_147.json"), "w") as f:
        json.dump(out, f)
    print({k: v for k, v in out.items() if k not in ("t1", "t2")}, "sum T/U", int(t1.sum()), int(t2.sum()))


if __name__ == "__main__":
    main()
