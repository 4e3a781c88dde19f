"""File-level runners (hpgv_run_assoc / hpgv_run_tdt: what run_association_test and
run_tdt_test do, assoc_runner.c:23-276, tdt_runner.c:23-279): VCF + PED files in, sorted
TSV out, with the GPU tokenizing the text."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import hpgv
from oracle import pyoracle as orc
from test_host_mirror_gpu import _families_csr, _fl, _parse_table, _write_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    hpgv.build()
    from importlib import import_module
    b = import_module("hpg-variant_amd._build")
    L = C.CDLL(b.HOSTLIB)
    L.hpgv_run_assoc.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_size_t, C.POINTER(C.c_long)]
    L.hpgv_run_tdt.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
    L.hpgv_host_last_error.restype = C.c_char_p
    yield L
    L.hpgv_host_shutdown()


def _vcf_from_batch(tmp, names, rows):
    with open(tmp / "in.vcf", "w") as f:
        f.write("##fileformat=VCFv4.1\n##source=test\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for v, (chrom, fmt, samples) in enumerate(rows):
            f.write("%s\t%d\trs%d\tA\tC\t.\tPASS\t.\t%s\t%s\n" % (chrom, 1000 + v, v, fmt, "\t".join(samples)))
    return str(tmp / "in.vcf")


def _codes(rows, strict):
    return np.array([[orc.encode_sample(s, fmt.split(":").index("GT"), strict) for s in samples]
                     for _, fmt, samples in rows], dtype=np.uint8)


@pytest.mark.parametrize("batch_bytes", [1 << 16, 1 << 22])
def test_run_assoc_from_files(host, tmp_path, batch_bytes):
    rng = np.random.default_rng(3)
    people, names, rows = _write_inputs(tmp_path, rng, 40, 30, 700)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    pheno = {p[1]: p[5] for p in people}
    cond = np.array([{2: orc.AFFECTED, 1: orc.UNAFFECTED}.get(pheno[n], orc.COND_OTHER) for n in names], np.uint8)
    gt = _codes(rows, True)
    is_x = np.array([1 if c == "X" else 0 for c, _, _ in rows], np.uint8)
    A1, A2, U1, U2 = orc.assoc_counts(gt, cond, is_x)
    lf = orc.logfact(len(names) * 10)
    for task, ext in ((1, "chisq"), (2, "fisher")):
        out = str(tmp_path / ("res." + ext))
        n = C.c_long(0)
        rc = host.hpgv_run_assoc(vcf.encode(), str(tmp_path / "ped.txt").encode(), out.encode(), task, batch_bytes, C.byref(n))
        assert rc == 0, host.hpgv_host_last_error()
        assert n.value == len(rows)
        gnu = subprocess.run(["sort", "-k1,1h", "-k2,2n", out], capture_output=True, text=True,
                             env=dict(os.environ, LC_ALL="C"), check=True).stdout
        assert open(out).read() == gnu                                  # sorted like the reference's file
        odds, chisq, p = orc.assoc_stats(task, A1, A2, U1, U2, lf)
        header, table = _parse_table(out)
        assert header[0] == "#CHR" and len(table) == len(rows)
        by_pos = {int(t[1]): t for t in table}
        for v in range(len(rows)):
            t = by_pos[1000 + v]
            assert t[0] == rows[v][0] and t[2] == "rs%d" % v
            assert (int(t[4]), int(t[9]), int(t[5]), int(t[10])) == (A1[v], A2[v], U1[v], U2[v]), v
            exp = [odds[v]] + ([chisq[v]] if task == 1 else []) + [p[v]]
            for g, e in zip([_fl(x) for x in t[13:]], exp):
                assert (np.isnan(g) and np.isnan(e)) or (np.isinf(g) and np.isinf(e)) or abs(g - e) <= 6e-7 * max(1.0, abs(e))


def test_run_tdt_from_files(host, tmp_path):
    rng = np.random.default_rng(4)
    people, names, rows = _write_inputs(tmp_path, rng, 80, 5, 500)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    out = str(tmp_path / "res.tdt")
    n = C.c_long(0)
    rc = host.hpgv_run_tdt(vcf.encode(), str(tmp_path / "ped.txt").encode(), out.encode(), 1 << 17, C.byref(n))
    assert rc == 0 and n.value == len(rows), host.hpgv_host_last_error()
    gt = _codes(rows, True)
    is_x = np.array([1 if c == "X" else 0 for c, _, _ in rows], np.uint8)
    t1, t2 = orc.tdt_counts(gt, *_families_csr(people, names), chrom_is_x=is_x)
    odds, chisq, p = orc.tdt_stats(t1, t2)
    header, table = _parse_table(out)
    assert header == "#CHR POS ID A1 A2 T U OR CHISQ P-VALUE".split() and len(table) == len(rows)
    by_pos = {int(t[1]): t for t in table}
    assert t1.sum() + t2.sum() > 0
    for v in range(len(rows)):
        t = by_pos[1000 + v]
        assert (int(t[5]), int(t[6])) == (t1[v], t2[v]), v
        for g, e in zip([_fl(x) for x in t[7:]], (odds[v], chisq[v], p[v])):
            assert (np.isnan(g) and np.isnan(e)) or (np.isinf(g) and np.isinf(e)) or abs(g - e) <= 6e-7 * max(1.0, abs(e))


def test_run_reports_bad_inputs(host, tmp_path):
    assert host.hpgv_run_assoc(b"/nonexistent.vcf", b"/nonexistent.ped", str(tmp_path / "o").encode(), 1, 1 << 20, None) != 0
    (tmp_path / "p.ped").write_text("f a 0 0 1 2\n")
    (tmp_path / "v.vcf").write_text("1\t5\trs\tA\tC\t.\t.\t.\tGT\t0/1\n")          # no #CHROM line
    assert host.hpgv_run_assoc(str(tmp_path / "v.vcf").encode(), str(tmp_path / "p.ped").encode(),
                               str(tmp_path / "o").encode(), 1, 1 << 20, None) != 0


@pytest.mark.parametrize("kind", ["gzip", "bgzf"])
def test_run_assoc_from_compressed_vcf(host, tmp_path, kind):
    # --compression gzip|bgzip (shared_options.c:60-61): same result file as from the plain text
    import gzip
    from test_host_logic_cpu import _bgzf
    rng = np.random.default_rng(5)
    people, names, rows = _write_inputs(tmp_path, rng, 60, 25, 1500)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    data = open(vcf, "rb").read()
    packed = str(tmp_path / "in.vcf.gz")
    open(packed, "wb").write(gzip.compress(data, 1) if kind == "gzip" else _bgzf(data, 0x4000))
    outs = []
    for path in (vcf, packed):
        out = str(tmp_path / ("res_" + os.path.basename(path)))
        n = C.c_long(0)
        rc = host.hpgv_run_assoc(path.encode(), str(tmp_path / "ped.txt").encode(), out.encode(), 1, 1 << 16, C.byref(n))
        assert rc == 0 and n.value == len(rows), host.hpgv_host_last_error()
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b"\n") == len(rows) + 1


def test_run_assoc_batches_of_short_lines(host, tmp_path):
    # lines much shorter than a complete record: more lines per batch than the arrays were sized for
    names = ["s%d" % j for j in range(50)]
    with open(tmp_path / "ped.txt", "w") as f:
        for j, nm in enumerate(names):
            f.write("F%d %s 0 0 1 %d\n" % (j, nm, 1 + j % 2))
    with open(tmp_path / "in.vcf", "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for v in range(6000):
            if v % 1500 == 7:
                f.write("1\t%d\trs%d\tA\tC\t.\t.\t.\tGT\t%s\n" % (100 + v, v, "\t".join(["0/1"] * 50)))
            else:
                f.write("1\t%d\trs%d\tA\tC\t.\t.\t.\tGT\n" % (100 + v, v))
    out = str(tmp_path / "res.chisq")
    n = C.c_long(0)
    rc = host.hpgv_run_assoc(str(tmp_path / "in.vcf").encode(), str(tmp_path / "ped.txt").encode(), out.encode(), 1, 1 << 16, C.byref(n))
    assert rc == 0 and n.value == 6000, host.hpgv_host_last_error()
    header, table = _parse_table(out)
    assert len(table) == 6000
    full = [t for t in table if int(t[4]) + int(t[9]) > 0]
    assert len(full) == 4 and all((int(t[4]), int(t[9]), int(t[5]), int(t[10])) == (25, 25, 25, 25) for t in full)


def test_run_vcf2epi_then_epistasis(host, tmp_path):
    # create_dataset_from_vcf (dataset_creator.c:24-222): VCF + PED -> binary dataset; then the epistasis run reads it
    import struct
    host.hpgv_run_vcf2epi.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
    host.hpgv_run_epistasis.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    rng = np.random.default_rng(8)
    people, names, rows = _write_inputs(tmp_path, rng, 30, 40, 300)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    out = str(tmp_path / "epistasis_dataset.bin")
    n = C.c_long(0)
    rc = host.hpgv_run_vcf2epi(vcf.encode(), str(tmp_path / "ped.txt").encode(), out.encode(), 1 << 16, C.byref(n))
    assert rc == 0 and n.value == len(rows), host.hpgv_host_last_error()
    blob = open(out, "rb").read()
    nv, nA, nU = struct.unpack("<III", blob[:12])
    pheno = {p[1]: p[5] for p in people}
    affected = np.array([pheno.get(nm) == 2 for nm in names])
    assert (nv, nA, nU) == (len(rows), int(affected.sum()), int((~affected).sum()))
    data = np.frombuffer(blob[12:], np.uint8).reshape(nv, nA + nU)
    # destination of VCF column k: cases in column order first, then the others (group_individuals_by_phenotype)
    dest = np.empty(len(names), np.int64)
    dest[affected] = np.arange(nA); dest[~affected] = nA + np.arange(nU)
    exp = np.zeros_like(data)
    for v, (_, fmt, samples) in enumerate(rows):
        pos = fmt.split(":").index("GT")
        for k_, s in enumerate(samples):
            st, a1, a2 = orc.get_alleles(s, pos)
            exp[v, dest[k_]] = 255 if st != 0 else (0 if (a1 == 0 and a2 == 0) else (1 if a1 != a2 else 2))
    assert np.array_equal(data, exp)
    rc = host.hpgv_run_epistasis(out.encode(), 4, 1, 5, 0, 1, str(tmp_path / "epi").encode())
    assert rc == 0, host.hpgv_host_last_error()
    report = open(str(tmp_path / "epi") + ".cv1.epi").read().splitlines()
    assert report[0] == "#CROSS VALIDATION 1" and len(report) == 5 + 5


class _Filters(C.Structure):
    _fields_ = [("min_maf", C.c_double), ("max_missing", C.c_double), ("max_mendel_errors", C.c_int),
                ("num_alleles", C.c_int), ("min_quality", C.c_double)]


@pytest.mark.parametrize("which", ["maf", "missing", "mendel", "alleles", "quality", "all"])
def test_run_assoc_with_record_filters(host, tmp_path, which):
    # filter_records on every batch (assoc_runner.c:191): --maf / --missing / --mendel on the GPU, --alleles / --quality
    # from the record's text; the surviving records carry the same statistics as in an unfiltered run
    rng = np.random.default_rng(31)
    people, names, rows = _write_inputs(tmp_path, rng, 25, 20, 400)
    n = len(names)
    # make the cohort interesting for the filters: rare variants, missing-heavy variants, multi-allelic ALT, QUAL values
    alts, quals = [], []
    for v, (chrom, fmt, samples) in enumerate(rows):
        pos = fmt.split(":").index("GT")
        if v % 5 == 0:                                             # rare alternative allele
            for k_ in range(n):
                if rng.random() < 0.9:
                    parts = samples[k_].split(":"); parts[pos] = "0/0"; samples[k_] = ":".join(parts)
        if v % 7 == 0:                                             # many missing calls
            for k_ in range(n):
                if rng.random() < 0.3:
                    parts = samples[k_].split(":"); parts[pos] = "./."; samples[k_] = ":".join(parts)
        alts.append(["C", "C,G", ".", "C,G,T"][v % 4]); quals.append([".", "10", "35.5", "90"][v % 4 if v % 3 else 3])
    with open(tmp_path / "in.vcf", "w") as f:
        f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for v, (chrom, fmt, samples) in enumerate(rows):
            f.write("%s\t%d\trs%d\tA\t%s\t%s\tPASS\t.\t%s\t%s\n" % (chrom, 1000 + v, v, alts[v], quals[v], fmt, "\t".join(samples)))
    vcf = str(tmp_path / "in.vcf")
    lax = _codes(rows, False)
    is_x = np.array([1 if c == "X" else 0 for c, _, _ in rows], np.uint8)
    col = {nm: i for i, nm in enumerate(names)}
    trios = [(col[p[2]], col[p[3]], col[p[1]], orc.MALE if p[4] == 1 else orc.FEMALE) for p in people
             if p[2] != "0" and p[3] != "0" and p[1] in col and p[2] in col and p[3] in col]
    merr, _ = orc.mendel_counts(lax, [t[0] for t in trios], [t[1] for t in trios], [t[2] for t in trios], [t[3] for t in trios], is_x)
    maf, miss = np.zeros(len(rows)), np.zeros(len(rows))
    for v in range(len(rows)):
        vs = orc.variant_stats(lax[v], 2)
        a0, a1 = vs.alleles_count[0], vs.alleles_count[1]
        maf[v] = min(a0, a1) / (a0 + a1) if a0 + a1 else 0.0
        miss[v] = vs.missing_genotypes / n
    n_alleles = np.array([1 if a == "." else 1 + len(a.split(",")) for a in alts])
    qual = np.array([-1.0 if q == "." else float(q) for q in quals])
    F = {"maf": _Filters(0.1, -1, -1, -1, -1), "missing": _Filters(-1, 0.1, -1, -1, -1), "mendel": _Filters(-1, -1, 1, -1, -1),
         "alleles": _Filters(-1, -1, -1, 2, -1), "quality": _Filters(-1, -1, -1, -1, 30.0), "all": _Filters(0.02, 0.4, 50, 2, 5.0)}[which]
    keep = np.ones(len(rows), bool)
    if F.min_maf >= 0: keep &= maf >= F.min_maf
    if F.max_missing >= 0: keep &= miss <= F.max_missing
    if F.max_mendel_errors >= 0: keep &= merr <= F.max_mendel_errors
    if F.num_alleles >= 0: keep &= n_alleles == F.num_alleles
    if F.min_quality >= 0: keep &= qual >= F.min_quality
    assert 0 < keep.sum() < len(rows)
    host.hpgv_run_set_filters.argtypes = [C.POINTER(_Filters)]
    ped = str(tmp_path / "ped.txt").encode()
    full, filt = str(tmp_path / "full.chisq"), str(tmp_path / "filt.chisq")
    cnt = C.c_long(0)
    host.hpgv_run_set_filters(None)
    assert host.hpgv_run_assoc(vcf.encode(), ped, full.encode(), 1, 1 << 16, C.byref(cnt)) == 0 and cnt.value == len(rows)
    host.hpgv_run_set_filters(C.byref(F))
    try:
        rc = host.hpgv_run_assoc(vcf.encode(), ped, filt.encode(), 1, 1 << 16, C.byref(cnt))
    finally:
        host.hpgv_run_set_filters(None)
    assert rc == 0, host.hpgv_host_last_error()
    assert cnt.value == int(keep.sum())
    all_lines = {int(l.split("\t")[1]): l for l in open(full).read().splitlines()[1:]}
    got = open(filt).read().splitlines()[1:]
    assert sorted(int(l.split("\t")[1]) for l in got) == [1000 + v for v in np.flatnonzero(keep)]
    assert all(l == all_lines[int(l.split("\t")[1])] for l in got)


def _stats_inputs(tmp_path, rng, n_fam, n_extra, n_variants):
    people, names, rows = _write_inputs(tmp_path, rng, n_fam, n_extra, n_variants)
    alts = [["C", "C,G", "CT", "C,G,T"][v % 4] for v in range(len(rows))]
    infos = [[".", "DP=10;AC=3;DB=1", "AF=0.5", "NS=3;AN=7"][v % 4] for v in range(len(rows))]
    quals = [[".", "10", "35.5", "90"][v % 4] for v in range(len(rows))]
    filts = [["PASS", "q10", ".", "PASS"][(v // 2) % 4] for v in range(len(rows))]
    with open(tmp_path / "in.vcf", "w") as f:
        f.write("##fileformat=VCFv4.1\n##source=test\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for v, (chrom, fmt, samples) in enumerate(rows):
            f.write("%s\t%d\trs%d\tA\t%s\t%s\t%s\t%s\t%s\t%s\n" % (chrom, 1000 + v, v, alts[v], quals[v], filts[v], infos[v], fmt, "\t".join(samples)))
    return people, names, rows, alts, infos, quals, filts


def _expected_counts(code_row, alt):
    """variant_stats_t of one record as get_variants_stats fills it (alleles seen in the calls extend num_alleles)."""
    na = 1 if alt == "." else 1 + len(alt.split(","))
    used = [n for code in code_row for n in (code >> 4, code & 0xF) if n != 0xF]
    na = max(2, na, (max(used) + 1) if used else 2)
    return na, orc.variant_stats(code_row, na)


def _gtc(vs, na):
    parts = []
    for i in range(na):
        for j in range(i, na):
            c = vs.genotypes_count[i * na + j] if i == j else vs.genotypes_count[i * na + j] + vs.genotypes_count[j * na + i]
            parts.append("%d/%d:%d" % (i, j, c))
    return ",".join(parts) + ",./.:%d" % vs.missing_genotypes


@pytest.mark.parametrize("overwrite", [0, 1])
def test_run_aggregate(host, tmp_path, overwrite):
    # run_aggregate (aggregate_runner.c:23-222): samples dropped, INFO extended by the allele / genotype counts
    host.hpgv_run_aggregate.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_size_t, C.POINTER(C.c_long)]
    rng = np.random.default_rng(61)
    people, names, rows, alts, infos, quals, filts = _stats_inputs(tmp_path, rng, 10, 30, 260)
    out = str(tmp_path / "in.vcf.aggregated")
    n = C.c_long(0)
    rc = host.hpgv_run_aggregate(str(tmp_path / "in.vcf").encode(), out.encode(), overwrite, 1 << 16, C.byref(n))
    assert rc == 0 and n.value == len(rows), host.hpgv_host_last_error()
    lines = open(out).read().splitlines()
    pre = "" if overwrite else "HPG_"
    assert lines[:2] == ["##fileformat=VCFv4.1", "##source=test"]
    assert [l.split(",")[0] for l in lines[2:6]] == ["##INFO=<ID=%sAC" % pre, "##INFO=<ID=%sAF" % pre, "##INFO=<ID=%sAN" % pre, "##INFO=<ID=HPG_GTC"]
    assert lines[6] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO" and len(lines) == 7 + len(rows)
    lax = _codes(rows, False)
    for v, line in enumerate(lines[7:]):
        t = line.split("\t")
        assert t[:7] == [rows[v][0], str(1000 + v), "rs%d" % v, "A", alts[v], quals[v], filts[v]], v
        na, vs = _expected_counts(lax[v], alts[v])
        total = sum(vs.alleles_count[:na])
        kept = [] if infos[v] == "." else [f for f in infos[v].split(";") if not (overwrite and f.split("=")[0] in ("AC", "AF", "AN"))]
        exp = kept + ["%sAC=%s" % (pre, ",".join(str(vs.alleles_count[k]) for k in range(1, na))),
                      "%sAF=%s" % (pre, ",".join("%.3f" % (np.float32(vs.alleles_count[k]) / np.float32(total) if total else 0.0) for k in range(1, na))),
                      "%sAN=%d" % (pre, total), "HPG_GTC=" + _gtc(vs, na)]
        assert t[7] == ";".join(exp), (v, t[7], ";".join(exp))


def test_run_stats_files(host, tmp_path):
    # run_stats (stats_runner.c:23-420): per-variant counters, per-sample counters, file summary
    host.hpgv_run_stats.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
    rng = np.random.default_rng(62)
    people, names, rows, alts, infos, quals, filts = _stats_inputs(tmp_path, rng, 20, 15, 330)
    prefix = str(tmp_path / "in.vcf")
    n = C.c_long(0)
    rc = host.hpgv_run_stats(prefix.encode(), str(tmp_path / "ped.txt").encode(), prefix.encode(), 1 << 16, C.byref(n))
    assert rc == 0 and n.value == len(rows), host.hpgv_host_last_error()
    lax = _codes(rows, False)
    is_x = np.array([1 if c == "X" else 0 for c, _, _ in rows], np.uint8)
    col = {nm: i for i, nm in enumerate(names)}
    trios = [(col[p[2]], col[p[3]], col[p[1]], orc.MALE if p[4] == 1 else orc.FEMALE) for p in people if p[2] != "0" and p[3] != "0"]
    merr, trio_err = orc.mendel_counts(lax, [t[0] for t in trios], [t[1] for t in trios], [t[2] for t in trios], [t[3] for t in trios], is_x)
    vl = open(prefix + ".stats-variants").read().splitlines()
    assert vl[0].split("\t")[:5] == ["#CHROM", "POS", "REF", "ALT", "NUM_ALLELES"] and len(vl) == 1 + len(rows)
    for v, line in enumerate(vl[1:]):
        t = line.split("\t")
        na, vs = _expected_counts(lax[v], alts[v])
        total = sum(vs.alleles_count[:na])
        assert t[:5] == [rows[v][0], str(1000 + v), "A", alts[v], str(na)]
        assert t[5] == ",".join(str(vs.alleles_count[k]) for k in range(na)) and t[7] == _gtc(vs, na)
        assert t[6] == ",".join("%.4f" % (np.float32(vs.alleles_count[k]) / np.float32(total) if total else 0.0) for k in range(na))
        assert (int(t[8]), int(t[9]), int(t[11])) == (vs.missing_alleles, vs.missing_genotypes, merr[v])
        for g, e in ((_fl(t[12]), vs.hw_chi2), (_fl(t[13]), vs.hw_p)):
            assert (np.isnan(g) and np.isnan(e)) or abs(g - e) <= 1e-5 * max(1.0, abs(e))
    sl = open(prefix + ".stats-samples").read().splitlines()
    miss = orc.sample_missing(lax)
    exp_err = np.zeros(len(names), np.int64)
    for t, e in zip(trios, trio_err):
        exp_err[t[2]] += e
    assert sl[0] == "#SAMPLE\tMISS_GT\tMEND_ER"
    assert [l.split("\t") for l in sl[1:]] == [[names[j], str(miss[j]), str(exp_err[j])] for j in range(len(names))]
    summary = open(prefix + ".stats-summary").read()
    n_multi = sum(1 for v in range(len(rows)) if _expected_counts(lax[v], alts[v])[0] > 2)
    assert "Number of variants = %d\nNumber of samples = %d\nNumber of biallelic variants = %d\nNumber of multiallelic variants = %d" % (
        len(rows), len(names), len(rows) - n_multi, n_multi) in summary
    snps = sum(1 for a in alts if all(len(x) == 1 for x in a.split(",")))
    assert "Number of SNP = %d\nNumber of indels = %d" % (snps, len(rows) - snps) in summary
    assert "Number of transitions = 0\nNumber of transversions = %d" % sum(1 for a in alts if a == "C") in summary      # A>C only
    assert "Percentage of PASS = %.2f%%" % (100.0 * sum(1 for f in filts if f == "PASS") / len(rows)) in summary
    qs = [float(q) for q in quals if q != "."]
    assert "Average quality = %.2f" % (sum(qs) / len(qs)) in summary
    # one file per phenotype value of the PED (stats_runner.c:267-297,319-323): the counters within the group
    pheno = {p[1]: str(p[5]) for p in people}
    values = []
    for p in people:
        if str(p[5]) not in values:
            values.append(str(p[5]))
    for val in values:
        cols = np.array([pheno.get(nm) == val for nm in names])
        gl = open("%s.phenotype-%s.stats-variants" % (prefix, val)).read().splitlines()
        assert gl[0].startswith("#CHROM\tPOS\tREF\tALT\tALLELES_COUNT") and len(gl) == 1 + len(rows)
        for v in range(0, len(rows), 3):
            t = gl[1 + v].split("\t")
            vs = orc.variant_stats(np.ascontiguousarray(lax[v][cols]), 2)
            g = vs.genotypes_count
            assert t[:4] == [rows[v][0], str(1000 + v), "A", alts[v]]
            assert t[4] == "%d,%d" % (vs.alleles_count[0], vs.alleles_count[1])
            assert t[6] == "0/0:%d,0/1:%d,1/1:%d,./.:%d" % (g[0], g[1] + g[2], g[3], vs.missing_genotypes), (val, v, t[6])
            assert (int(t[7]), int(t[8])) == (vs.missing_alleles, vs.missing_genotypes)
            for got, e in ((_fl(t[10]), vs.hw_chi2), (_fl(t[11]), vs.hw_p)):
                assert (np.isnan(got) and np.isnan(e)) or abs(got - e) <= 1e-5 * max(1.0, abs(e))


def test_run_assoc_from_bgzf_decoded_on_the_gpu(host, tmp_path, capfd):
    # a bgzip file of 256 blocks or more is decoded on the device (hpgv_inflate_blocks_dev) and tokenized from device memory:
    # same result file as from the plain text -- by that path, with every third block refused by the device decoder (decoded
    # by the host and patched in), and with the device path switched off; several pipeline batches, lines across blocks
    from test_host_logic_cpu import _bgzf
    rng = np.random.default_rng(15)
    people, names, rows = _write_inputs(tmp_path, rng, 50, 40, 4000)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    data = open(vcf, "rb").read()
    packed = str(tmp_path / "in.vcf.gz")
    open(packed, "wb").write(_bgzf(data, 0x700))
    assert len(data) // 0x700 > 300
    ped = str(tmp_path / "ped.txt").encode()

    def run(path, tag, env=None):
        for k, v in (env or {}).items():
            os.environ[k] = v
        try:
            out = str(tmp_path / ("res_" + tag))
            n = C.c_long(0)
            rc = host.hpgv_run_assoc(path.encode(), ped, out.encode(), 1, 1 << 17, C.byref(n))
            assert rc == 0 and n.value == len(rows), host.hpgv_host_last_error()
            return open(out, "rb").read()
        finally:
            for k in (env or {}):
                del os.environ[k]
    plain = run(vcf, "plain")
    assert plain.count(b"\n") == len(rows) + 1
    assert run(packed, "gpu") == plain
    assert run(packed, "serial_walk", {"HPGV_SERIAL_BGZF_WALK": "1"}) == plain        # the block table by one thread instead of the team
    # blocks of very different sizes: the team that builds the block table walks the file in 64 segments, and with 64 KB
    # blocks next to 1.8 KB ones some segments hold no block start at all; the trace says which walk built the table
    import struct
    import zlib

    def stored(ch):                                                  # one BGZF block holding `ch` uncompressed (deflate level 0)
        co = zlib.compressobj(0, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
                + comp + struct.pack("<II", zlib.crc32(ch), len(ch)))
    n_big, big = 10, 0xfe00
    mixed = str(tmp_path / "mixed.vcf.gz")
    open(mixed, "wb").write(b"".join(stored(data[i * big:(i + 1) * big]) for i in range(n_big)) + _bgzf(data[n_big * big:], 0x700))
    assert os.path.getsize(mixed) // 64 < big // 2                   # a stored block spans more than two of the 64 segments
    capfd.readouterr()
    assert run(mixed, "mixed", {"HPGV_RUN_TRACE": "1"}) == plain     # the block table found on the device, in the uploaded bytes
    err = capfd.readouterr().err
    assert "blocks found" in err and "stage: walk" not in err
    assert run(mixed, "mixed_team", {"HPGV_RUN_TRACE": "1", "HPGV_BGZF_HOST_TABLE": "1"}) == plain     # ... built by the host's team
    err = capfd.readouterr().err
    assert "stage: walk" in err and "serial walk" not in err and "blocks found" not in err
    assert run(mixed, "mixed_serial", {"HPGV_RUN_TRACE": "1", "HPGV_SERIAL_BGZF_WALK": "1"}) == plain  # ... by one thread
    assert "serial walk" in capfd.readouterr().err
    assert run(packed, "stretches", {"HPGV_TEST_SCAN_ROWS": "300", "HPGV_RUN_TRACE": "1"}) == plain      # eleven stretches through the four slots
    assert capfd.readouterr().err.count("blocks found") >= 10
    assert run(packed, "stretches_patched", {"HPGV_TEST_SCAN_ROWS": "300", "HPGV_TEST_GPU_INFLATE_REFUSE_EVERY": "7"}) == plain
    capfd.readouterr()
    # without a text buffer that grows the table has to be complete before the text is allocated: the host builds it
    assert run(packed, "fixed_text", {"HPGV_TEST_SCAN_ROWS": "300", "HPGV_NO_GROWING_TEXT": "1", "HPGV_RUN_TRACE": "1"}) == plain
    assert "stage: walk" in capfd.readouterr().err
    # blocks whose header carries a second extra field are not what the device scan looks for: those stretches are walked on
    # the host, the rest of the chain is found on the device again

    def odd(ch):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 12) + b"XY" + struct.pack("<H", 2) + b"zz" + b"BC"
                + struct.pack("<HH", 2, 12 + 12 + len(comp) + 8 - 1) + comp + struct.pack("<II", zlib.crc32(ch), len(ch)))
    pieces, at, k = [], 0, 0
    while at < len(data):
        n = 0x700 if k % 40 else 0x300
        pieces.append(odd(data[at:at + n]) if k % 40 == 0 else _bgzf(data[at:at + n], 0x700)[:-28])
        at += n; k += 1
    oddf = str(tmp_path / "odd.vcf.gz")
    open(oddf, "wb").write(b"".join(pieces) + _bgzf(b"", 0x700))
    assert run(oddf, "odd", {"HPGV_RUN_TRACE": "1"}) == plain
    err = capfd.readouterr().err
    assert "walked on the host" in err and "blocks found" in err
    assert run(packed, "patched", {"HPGV_TEST_GPU_INFLATE_REFUSE_EVERY": "3"}) == plain
    assert run(packed, "copied_back", {"HPGV_NO_DEVICE_WINDOWS": "1"}) == plain       # device decoding, whole windows copied back
    assert run(packed, "cpu", {"HPGV_NO_GPU_INFLATE": "1"}) == plain
    # the other runners read more of a line's head (QUAL .. INFO, FORMAT): tdt, vcf2epi, aggregate and the stats files
    # from the device-decoded file and from the plain text
    host.hpgv_run_vcf2epi.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
    host.hpgv_run_aggregate.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_size_t, C.POINTER(C.c_long)]
    host.hpgv_run_stats.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
    both = {}
    for tag, path in (("p", vcf), ("z", packed)):
        n = C.c_long(0)
        o = lambda name: str(tmp_path / (name + "_" + tag))
        assert host.hpgv_run_tdt(path.encode(), ped, o("tdt").encode(), 1 << 17, C.byref(n)) == 0, host.hpgv_host_last_error()
        assert host.hpgv_run_vcf2epi(path.encode(), ped, o("epi").encode(), 1 << 17, C.byref(n)) == 0, host.hpgv_host_last_error()
        assert host.hpgv_run_aggregate(path.encode(), o("agg").encode(), 1, 1 << 17, C.byref(n)) == 0, host.hpgv_host_last_error()
        assert host.hpgv_run_stats(path.encode(), ped, o("st").encode(), 1 << 17, C.byref(n)) == 0, host.hpgv_host_last_error()
        both[tag] = [open(o("tdt"), "rb").read(), open(o("epi"), "rb").read(), open(o("agg"), "rb").read()] + \
                    [open(o("st") + ext, "rb").read() for ext in (".stats-variants", ".stats-samples", ".stats-summary")]
    assert all(len(x) > 0 for x in both["p"])
    assert both["p"] == both["z"]
    # a file cut off inside a block, and one with bytes that are no block after its end: an error, by either way to the
    # block table, not a hang and not a result
    whole = open(packed, "rb").read()
    for tag, blob in (("cut", whole[:len(whole) * 2 // 3 + 5]), ("tail", whole + b"not a block" * 20)):
        bad = str(tmp_path / ("bad_" + tag + ".vcf.gz"))
        open(bad, "wb").write(blob)
        for env in ({}, {"HPGV_BGZF_HOST_TABLE": "1"}):
            os.environ.update(env)
            try:
                n = C.c_long(0)
                rc = host.hpgv_run_assoc(bad.encode(), ped, str(tmp_path / "bad_out").encode(), 1, 1 << 17, C.byref(n))
            finally:
                for k in env:
                    del os.environ[k]
            assert rc != 0, (tag, env, n.value)


def test_run_assoc_bgzf_text_buffer_grows_during_the_run(host, tmp_path, capfd):
    # a text that outgrows what was committed for it when its first blocks were seen (here: the estimate cut to 30 %): the
    # device buffer grows -- more pieces mapped into its address range -- while the pipeline reads what is already there
    import zlib
    from test_host_logic_cpu import _bgzf
    rng = np.random.default_rng(21)
    n_samples, n_variants = 60, 330_000
    names = ["S%d" % j for j in range(n_samples)]
    with open(tmp_path / "ped.txt", "w") as f:
        for j, nm in enumerate(names):
            f.write("F%d %s 0 0 %d %d\n" % (j, nm, 1 + j % 2, 1 + (j * 7 // 3) % 2))
    gts = np.array(["0/0", "0/1", "1/1", "./."])
    bodies = ["\t".join(gts[rng.choice(4, size=n_samples, p=[0.5, 0.3, 0.19, 0.01])]) for _ in range(997)]
    head = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"
    text = (head + "".join("%d\t%d\trs%d\tA\tG\t.\tPASS\t.\tGT\t%s\n" % (1 + i * 22 // n_variants, 1000 + i, i, bodies[i % 997])
                           for i in range(n_variants))).encode()
    assert len(text) > (80 << 20)                                   # more than one 64 MB piece of the buffer
    vcf, packed = str(tmp_path / "big.vcf"), str(tmp_path / "big.vcf.gz")
    open(vcf, "wb").write(text)
    out = bytearray()
    for i in range(0, len(text), 0xff00):                            # bgzip blocks, zlib level 1 for speed
        ch = text[i:i + 0xff00]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + (18 + len(comp) + 8 - 1).to_bytes(2, "little") + comp
        out += zlib.crc32(ch).to_bytes(4, "little") + len(ch).to_bytes(4, "little")
    open(packed, "wb").write(bytes(out) + _bgzf(b"", 0x700))
    ped = str(tmp_path / "ped.txt").encode()

    def run(path, tag, env=None, batch=8 << 20):
        os.environ.update(env or {})
        try:
            o = str(tmp_path / ("res_" + tag))
            n = C.c_long(0)
            rc = host.hpgv_run_assoc(path.encode(), ped, o.encode(), 1, batch, C.byref(n))
            assert rc == 0 and n.value == n_variants, host.hpgv_host_last_error()
            return open(o, "rb").read()
        finally:
            for k in (env or {}):
                del os.environ[k]
    plain = run(vcf, "plain")
    capfd.readouterr()
    assert run(packed, "grown", {"HPGV_TEST_SCAN_ROWS": "200", "HPGV_TEST_TEXT_ESTIMATE_PERCENT": "30", "HPGV_RUN_TRACE": "1"}) == plain
    err = capfd.readouterr().err
    assert "streaming" in err and err.count("blocks found") >= 6
    assert run(packed, "as_estimated") == plain
    # the windows of a text that stays on the device are about a 64th of the file whatever the caller's batch size: with
    # batches of 256 KB this text goes through in some 70 windows of 1.5 MB, or, with that switched off, in 370 of 256 KB
    capfd.readouterr()
    assert run(packed, "large_windows", {"HPGV_RUN_TRACE": "1"}, batch=1 << 18) == plain
    n_large = int(capfd.readouterr().err.split(" records, ")[1].split(" batches")[0])
    assert run(packed, "callers_windows", {"HPGV_RUN_TRACE": "1", "HPGV_NO_LARGE_WINDOWS": "1"}, batch=1 << 18) == plain
    n_small = int(capfd.readouterr().err.split(" records, ")[1].split(" batches")[0])
    assert 50 <= n_large <= 90 and n_small >= 4 * n_large, (n_large, n_small)


@pytest.mark.parametrize("chroms", [("1", "2", "10", "22"), ("1", "2", "22", "X"), ("1", "1", "1", "1")], ids=["numbered", "x_last", "one"])
def test_run_assoc_result_order_is_known_without_reading_the_file_back(host, tmp_path, chroms, capfd):
    # a position-sorted VCF gives a result file that is already in `sort -k1,1h -k2,2n` order, and the runner knows that
    # when it has written the last record; chromosome X after the numbered ones is NOT in that order (-h reads "X" as 0):
    # then the file is sorted afterwards, as for any other input; equal positions are ordered by the whole line
    rng = np.random.default_rng(31)
    people, names, rows = _write_inputs(tmp_path, rng, 30, 20, 6000, chroms=("1",))
    n = len(rows)
    with open(tmp_path / "sorted.vcf", "w") as f:
        f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for v, (_, fmt, samples) in enumerate(rows):
            pos = 1000 + v // 2 * 2                                # pairs of records share a position
            f.write("%s\t%d\trs%d\tA\tC\t.\tPASS\t.\t%s\t%s\n" % (chroms[v * len(chroms) // n], pos, v, fmt, "\t".join(samples)))
    out = str(tmp_path / "res.chisq")
    cnt = C.c_long(0)
    os.environ["HPGV_RUN_TRACE"] = "1"
    try:
        capfd.readouterr()
        rc = host.hpgv_run_assoc(str(tmp_path / "sorted.vcf").encode(), str(tmp_path / "ped.txt").encode(), out.encode(), 1, 1 << 16, C.byref(cnt))
        err = capfd.readouterr().err
    finally:
        del os.environ["HPGV_RUN_TRACE"]
    assert rc == 0 and cnt.value == n, host.hpgv_host_last_error()
    gnu = subprocess.run(["sort", "-k1,1h", "-k2,2n", out], capture_output=True, text=True, env=dict(os.environ, LC_ALL="C"), check=True).stdout
    assert open(out).read() == gnu
    assert ("in order as written" in err) == (chroms[-1] != "X")       # read back and sorted only where it had to be


def test_run_fails_on_a_bgzf_block_with_a_wrong_crc(host, tmp_path):
    # one literal byte flipped inside a STORED block -- the stream still inflates to ISIZE bytes, only the
    # CRC-32 of the trailer tells.  The run must fail (as htslib's reader does) by the device path and by the host path, and say
    # why; HPGV_BGZF_VERIFY=0 switches the check off and the damaged genotype goes into the statistics unnoticed.
    import struct
    import zlib
    from test_host_logic_cpu import _bgzf
    rng = np.random.default_rng(16)
    people, names, rows = _write_inputs(tmp_path, rng, 50, 40, 4000)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    data = open(vcf, "rb").read()
    ped = str(tmp_path / "ped.txt").encode()

    def stored(ch, damage_at=None):
        co = zlib.compressobj(0, zlib.DEFLATED, -15)
        comp = bytearray(co.compress(ch) + co.flush())
        if damage_at is not None:
            at = 5 + damage_at                                        # one stored DEFLATE block: 5 header bytes, then the text as it is
            assert comp[at] == ch[damage_at]
            comp[at] = ord("1") if comp[at] == ord("0") else ord("0")
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
                + bytes(comp) + struct.pack("<II", zlib.crc32(ch), len(ch)))
    cut = 0x700 * 150
    block = data[cut: cut + 0x3000]
    # a genotype character (the byte after a TAB inside the sample columns): 0 <-> 1 keeps the line well-formed
    line_start = block.index(b"\n") + 1
    tabs, at = 0, line_start
    while tabs < 12:
        at = block.index(b"\t", at) + 1
        tabs += 1
    assert block[at] in b"01"
    good = str(tmp_path / "good.vcf.gz")
    bad = str(tmp_path / "bad.vcf.gz")
    open(good, "wb").write(_bgzf(data[:cut], 0x700)[:-28] + stored(block) + _bgzf(data[cut + 0x3000:], 0x700))
    open(bad, "wb").write(_bgzf(data[:cut], 0x700)[:-28] + stored(block, at) + _bgzf(data[cut + 0x3000:], 0x700))

    def run(path, env):
        os.environ.update(env)
        try:
            n = C.c_long(0)
            out = str(tmp_path / "res")
            rc = host.hpgv_run_assoc(path.encode(), ped, out.encode(), 1, 1 << 17, C.byref(n))
            return rc, (open(out, "rb").read() if rc == 0 else None), host.hpgv_host_last_error()
        finally:
            for k in env:
                del os.environ[k]
    host.hpgv_host_last_error.restype = C.c_char_p
    rc, ref, _ = run(good, {})
    assert rc == 0
    for env in ({}, {"HPGV_BGZF_HOST_TABLE": "1"}, {"HPGV_NO_GPU_INFLATE": "1"}):
        rc, _, why = run(bad, env)
        assert rc != 0, env
        assert b"CRC-32" in why, (env, why)
        rc, res, _ = run(good, env)
        assert rc == 0 and res == ref
    rc, res, _ = run(bad, {"HPGV_BGZF_VERIFY": "0"})
    assert rc == 0 and res != ref                                     # unchecked: the wrong genotype is counted
