"""ctypes view of the CPU oracle (oracle/hpgv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhpgv_oracle.so")

MALE, FEMALE, SEX_UNKNOWN = 0, 1, 2
UNAFFECTED, AFFECTED, COND_OTHER = 0, 1, 2
TASK_CHISQ, TASK_FISHER = 1, 2
MAX_ALLELES = 15


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("hpgv_oracle.c", "hpgv_epi_oracle.c", "hpgv_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    # never build from inside a profiled process: the children (make, gcc, ld) would inherit the profiler's preload
    # and run under it on the GPU box; the profile scripts build before their first rocprofv3 line
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("oracle/_build is missing or stale and this process runs under a profiler: "
                           "run `python __graft_entry__.py` first")
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class VariantStats(C.Structure):
    _fields_ = [
        ("num_alleles", C.c_int32),
        ("alleles_count", C.c_int32 * MAX_ALLELES),
        ("genotypes_count", C.c_int32 * (MAX_ALLELES * MAX_ALLELES)),
        ("missing_alleles", C.c_int32),
        ("missing_genotypes", C.c_int32),
        ("maf", C.c_double), ("maf_allele", C.c_int32),
        ("mgf", C.c_double), ("mgf_genotype", C.c_int32),
        ("hw_n_AA", C.c_int32), ("hw_n_Aa", C.c_int32), ("hw_n_aa", C.c_int32),
        ("hw_chi2", C.c_double), ("hw_p", C.c_double),
    ]


_lib = None


def effective_cpus():
    """CPUs this process may really use: its affinity mask capped by the cgroup CPU quota (a container can see 256
    CPUs and be allowed 16; a team of 256 threads is then throttled to a crawl)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    L.orc_set_threads(effective_cpus())
    p_u8 = C.POINTER(C.c_uint8)
    p_i32 = C.POINTER(C.c_int32)
    p_f64 = C.POINTER(C.c_double)
    L.orc_get_field_position_in_format.restype = C.c_int
    L.orc_get_field_position_in_format.argtypes = [C.c_char_p, C.c_char_p]
    L.orc_get_alleles.restype = C.c_int
    L.orc_get_alleles.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_encode_sample.restype = C.c_uint8
    L.orc_encode_sample.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.orc_decode.restype = C.c_int
    L.orc_decode.argtypes = [C.c_uint8, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_assoc_count_individual.restype = None
    L.orc_assoc_count_individual.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 4
    L.orc_assoc_basic_test.restype = C.c_double
    L.orc_assoc_basic_test.argtypes = [C.c_int] * 4
    L.orc_assoc_odds_ratio.restype = C.c_double
    L.orc_assoc_odds_ratio.argtypes = [C.c_int] * 4
    L.orc_chisq_p_value.restype = C.c_double
    L.orc_chisq_p_value.argtypes = [C.c_double]
    L.orc_chrom_is_x.restype = C.c_int
    L.orc_chrom_is_x.argtypes = [C.c_char_p, C.c_int]
    L.orc_init_logarithm_array.restype = None
    L.orc_init_logarithm_array.argtypes = [C.c_int, p_f64]
    L.orc_fisher_two_sided.restype = C.c_double
    L.orc_fisher_two_sided.argtypes = [C.c_int] * 4 + [p_f64]
    L.orc_assoc_packed.restype = None
    L.orc_assoc_packed.argtypes = [p_u8, C.c_size_t, C.c_int, C.c_int, p_u8, p_u8] + [p_i32] * 4
    L.orc_assoc_stats.restype = None
    L.orc_assoc_stats.argtypes = [C.c_int, C.c_int] + [p_i32] * 4 + [p_f64] * 4
    L.orc_check_mendel.restype = C.c_int
    L.orc_check_mendel.argtypes = [C.c_char_p] + [C.c_int] * 7
    L.orc_tdt_packed.restype = None
    L.orc_tdt_packed.argtypes = [p_u8, C.c_size_t, C.c_int, p_u8, C.c_int,
                                 p_i32, p_i32, p_i32, p_i32, p_u8, p_i32, p_i32]
    L.orc_mendel_counts.restype = None
    L.orc_mendel_counts.argtypes = [p_u8, C.c_size_t, C.c_int, p_u8, C.c_int, p_i32, p_i32, p_i32, p_u8, p_i32, p_i32]
    L.orc_tdt_stats.restype = None
    L.orc_tdt_stats.argtypes = [C.c_int, p_i32, p_i32, p_f64, p_f64, p_f64]
    L.orc_variant_stats.restype = None
    L.orc_variant_stats.argtypes = [p_u8, C.c_int, C.c_int, C.POINTER(VariantStats)]
    L.orc_sample_missing.restype = None
    L.orc_sample_missing.argtypes = [p_u8, C.c_size_t, C.c_int, C.c_int, p_i32]
    L.orc_tokenize.restype = C.c_int
    L.orc_tokenize.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, p_u8, C.c_size_t, p_u8, p_i32]
    L.orc_hwe.restype = None
    L.orc_hwe.argtypes = [C.c_int] * 3 + [p_f64, p_f64]
    L.orc_splitmix64.restype = C.c_uint64
    L.orc_splitmix64.argtypes = [C.c_uint64]
    L.orc_synth_thresholds.restype = None
    L.orc_synth_thresholds.argtypes = [C.c_uint64, C.POINTER(C.c_uint32)]
    L.orc_synth_genotype.restype = C.c_uint8
    L.orc_synth_genotype.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    L.orc_synth_matrix.restype = None
    L.orc_synth_matrix.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_size_t, p_u8]
    L.orc_baseline_assoc_packed.restype = C.c_double
    L.orc_baseline_assoc_packed.argtypes = [p_u8, C.c_size_t, C.c_int, C.c_int, p_u8, C.c_int,
                                            C.POINTER(C.c_int)]
    L.orc_baseline_assoc_text.restype = C.c_double
    L.orc_baseline_assoc_text.argtypes = [C.c_uint64, C.c_int, C.c_int, p_u8, C.c_int,
                                          C.POINTER(C.c_int)]
    L.orc_baseline_assoc_text_task.restype = C.c_double
    L.orc_baseline_assoc_text_task.argtypes = [C.c_uint64, C.c_int, C.c_int, p_u8, C.c_int, p_f64, C.c_int,
                                               C.POINTER(C.c_int)]
    L.orc_baseline_tdt_text.restype = C.c_double
    L.orc_baseline_tdt_text.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, p_i32, p_i32, p_i32, p_i32, p_u8,
                                        C.c_int, C.POINTER(C.c_int)]
    L.orc_baseline_stats_text.restype = C.c_double
    L.orc_baseline_stats_text.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.orc_tdt_text.restype = None
    L.orc_tdt_text.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_char_p), p_u8, C.c_int,
                               p_i32, p_i32, p_i32, p_i32, p_u8, p_i32, p_i32]
    _lib = L
    return L


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def get_alleles(sample, gt_pos=0):
    a1, a2 = C.c_int(), C.c_int()
    st = lib().orc_get_alleles(sample.encode(), gt_pos, C.byref(a1), C.byref(a2))
    return st, a1.value, a2.value


def encode_sample(sample, gt_pos=0, strict=True):
    return lib().orc_encode_sample(sample.encode(), gt_pos, 1 if strict else 0)


def encode_matrix(rows, gt_pos=0, strict=True):
    """rows: list of lists of sample strings -> uint8 [n_variants, n_samples]"""
    out = np.empty((len(rows), len(rows[0]) if rows else 0), dtype=np.uint8)
    for i, r in enumerate(rows):
        for j, s in enumerate(r):
            out[i, j] = encode_sample(s, gt_pos, strict)
    return out


def logfact(n):
    t = np.zeros(n, dtype=np.float64)
    lib().orc_init_logarithm_array(n, _p(t, C.c_double))
    return t


def assoc_counts(gt, condition, chrom_is_x=None, n_samples=None):
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    nv, pitch = gt.shape
    ns = pitch if n_samples is None else n_samples
    condition = np.ascontiguousarray(condition, dtype=np.uint8)
    assert condition.shape[0] >= ns
    x = None if chrom_is_x is None else np.ascontiguousarray(chrom_is_x, dtype=np.uint8)
    out = [np.zeros(nv, dtype=np.int32) for _ in range(4)]
    lib().orc_assoc_packed(_p(gt, C.c_uint8), pitch, nv, ns, _p(condition, C.c_uint8),
                           _p(x, C.c_uint8), *[_p(o, C.c_int32) for o in out])
    return out


def assoc_stats(task, A1, A2, U1, U2, lf=None):
    n = len(A1)
    odds, chisq, p = (np.zeros(n) for _ in range(3))
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (A1, A2, U1, U2)]
    lib().orc_assoc_stats(task, n, *[_p(a, C.c_int32) for a in arrs], _p(lf, C.c_double),
                          _p(odds, C.c_double), _p(chisq, C.c_double), _p(p, C.c_double))
    return odds, (chisq if task == TASK_CHISQ else None), p


def fisher_terms_needed(A1, A2, U1, U2, lf, rel_cut=1e-22):
    """Per table: the hypergeometric terms its two-sided p-value needs under a relative tail cut (orc_fisher_terms_needed)."""
    L = lib()
    L.orc_fisher_terms_needed.restype = C.c_long
    L.orc_fisher_terms_needed.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double), C.c_double]
    lfp = _p(np.ascontiguousarray(lf, dtype=np.float64), C.c_double)
    return np.array([L.orc_fisher_terms_needed(int(a), int(b), int(c), int(d), lfp, rel_cut)
                     for a, b, c, d in zip(A1, A2, U1, U2)], dtype=np.int64)


def check_mendel(chrom, f1, f2, m1, m2, c1, c2, sex):
    return lib().orc_check_mendel(chrom.encode(), f1, f2, m1, m2, c1, c2, sex)


def tdt_counts(gt, father_col, mother_col, child_off, child_col, child_sex, chrom_is_x=None):
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    nv, pitch = gt.shape
    fc = np.ascontiguousarray(father_col, dtype=np.int32)
    mc = np.ascontiguousarray(mother_col, dtype=np.int32)
    co = np.ascontiguousarray(child_off, dtype=np.int32)
    cc = np.ascontiguousarray(child_col, dtype=np.int32)
    cs = np.ascontiguousarray(child_sex, dtype=np.uint8)
    x = None if chrom_is_x is None else np.ascontiguousarray(chrom_is_x, dtype=np.uint8)
    t1 = np.zeros(nv, dtype=np.int32)
    t2 = np.zeros(nv, dtype=np.int32)
    lib().orc_tdt_packed(_p(gt, C.c_uint8), pitch, nv, _p(x, C.c_uint8), len(fc),
                         _p(fc, C.c_int32), _p(mc, C.c_int32), _p(co, C.c_int32),
                         _p(cc, C.c_int32), _p(cs, C.c_uint8), _p(t1, C.c_int32), _p(t2, C.c_int32))
    return t1, t2


def mendel_counts(gt, father_col, mother_col, child_col, child_sex, chrom_is_x=None):
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    nv, pitch = gt.shape
    fc, mc, cc = (np.ascontiguousarray(a, dtype=np.int32) for a in (father_col, mother_col, child_col))
    cs = np.ascontiguousarray(child_sex, dtype=np.uint8)
    x = None if chrom_is_x is None else np.ascontiguousarray(chrom_is_x, dtype=np.uint8)
    errors = np.zeros(nv, np.int32)
    trio = np.zeros(len(cc), np.int32)
    lib().orc_mendel_counts(_p(gt, C.c_uint8), pitch, nv, _p(x, C.c_uint8), len(cc), _p(fc, C.c_int32), _p(mc, C.c_int32),
                            _p(cc, C.c_int32), _p(cs, C.c_uint8), _p(errors, C.c_int32), _p(trio, C.c_int32))
    return errors, trio


def tdt_stats(t1, t2):
    n = len(t1)
    t1 = np.ascontiguousarray(t1, dtype=np.int32)
    t2 = np.ascontiguousarray(t2, dtype=np.int32)
    odds, chisq, p = (np.zeros(n) for _ in range(3))
    lib().orc_tdt_stats(n, _p(t1, C.c_int32), _p(t2, C.c_int32), _p(odds, C.c_double),
                        _p(chisq, C.c_double), _p(p, C.c_double))
    return odds, chisq, p


def variant_stats(row, num_alleles=2):
    row = np.ascontiguousarray(row, dtype=np.uint8)
    vs = VariantStats()
    lib().orc_variant_stats(_p(row, C.c_uint8), row.shape[0], num_alleles, C.byref(vs))
    return vs


def sample_missing(gt):
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    out = np.zeros(gt.shape[1], dtype=np.int32)
    lib().orc_sample_missing(_p(gt, C.c_uint8), gt.shape[1], gt.shape[0], gt.shape[1], _p(out, C.c_int32))
    return out


def tokenize(text, n_samples, strict=True, max_lines=None):
    if isinstance(text, str):
        text = text.encode()
    if max_lines is None:
        max_lines = text.count(b"\n") + 1
    pitch = max(n_samples, 1)
    gt = np.zeros((max_lines, pitch), np.uint8)
    is_x = np.zeros(max_lines, np.uint8)
    status = np.zeros(max_lines, np.int32)
    n = lib().orc_tokenize(text, len(text), n_samples, 1 if strict else 0, max_lines, _p(gt, C.c_uint8), pitch,
                           _p(is_x, C.c_uint8), _p(status, C.c_int32))
    k = min(n, max_lines)
    return dict(n_lines=n, gt=gt[:k, :n_samples], is_x=is_x[:k], status=status[:k])


def hwe(n_AA, n_Aa, n_aa):
    chi2, p = C.c_double(), C.c_double()
    lib().orc_hwe(n_AA, n_Aa, n_aa, C.byref(chi2), C.byref(p))
    return chi2.value, p.value


def synth_thresholds(v):
    thr = (C.c_uint32 * 3)()
    lib().orc_synth_thresholds(v, thr)
    return [thr[0], thr[1], thr[2]]


def synth_matrix(v0, n_variants, n_samples, pitch=None):
    pitch = n_samples if pitch is None else pitch
    gt = np.empty((n_variants, pitch), dtype=np.uint8)
    lib().orc_synth_matrix(v0, n_variants, n_samples, pitch, _p(gt, C.c_uint8))
    return gt


def baseline_assoc_packed(gt, n_samples, condition, n_threads):
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    cond = np.ascontiguousarray(condition, dtype=np.uint8)
    used = C.c_int(0)
    sec = lib().orc_baseline_assoc_packed(_p(gt, C.c_uint8), gt.shape[1], gt.shape[0], n_samples,
                                          _p(cond, C.c_uint8), n_threads, C.byref(used))
    return sec, used.value


def baseline_assoc_text(v0, n_variants, n_samples, condition, n_threads):
    cond = np.ascontiguousarray(condition, dtype=np.uint8)
    used = C.c_int(0)
    sec = lib().orc_baseline_assoc_text(v0, n_variants, n_samples, _p(cond, C.c_uint8),
                                        n_threads, C.byref(used))
    return sec, used.value


def baseline_fisher_text(v0, n_variants, n_samples, condition, lf_table, n_threads):
    """assoc --fisher on sample strings: the scan of assoc.c:50-57 + assoc.c:69-75 on the caller's ln(i!) table."""
    cond = np.ascontiguousarray(condition, dtype=np.uint8)
    lf = np.ascontiguousarray(lf_table, dtype=np.float64)
    used = C.c_int(0)
    sec = lib().orc_baseline_assoc_text_task(v0, n_variants, n_samples, _p(cond, C.c_uint8), TASK_FISHER,
                                             _p(lf, C.c_double), n_threads, C.byref(used))
    return sec, used.value


def baseline_tdt_text(v0, n_variants, n_samples, father_col, mother_col, child_off, child_col, child_sex, n_threads):
    fc, mc, co, cc = (np.ascontiguousarray(a, dtype=np.int32) for a in (father_col, mother_col, child_off, child_col))
    cs = np.ascontiguousarray(child_sex, dtype=np.uint8)
    used = C.c_int(0)
    sec = lib().orc_baseline_tdt_text(v0, n_variants, n_samples, len(fc), _p(fc, C.c_int32), _p(mc, C.c_int32),
                                      _p(co, C.c_int32), _p(cc, C.c_int32), _p(cs, C.c_uint8), n_threads, C.byref(used))
    return sec, used.value


def baseline_stats_text(v0, n_variants, n_samples, n_threads):
    used = C.c_int(0)
    sec = lib().orc_baseline_stats_text(v0, n_variants, n_samples, n_threads, C.byref(used))
    return sec, used.value


def tdt_text(rows, father_col, mother_col, child_off, child_col, child_sex, chrom_is_x=None):
    """orc_tdt_text on rows of sample strings (all FORMAT "GT"): the text-faithful twin of tdt_counts."""
    nv, ns = len(rows), len(rows[0]) if rows else 0
    flat = (C.c_char_p * (nv * ns))(*[s.encode() for r in rows for s in r])
    fmts = (C.c_char_p * max(nv, 1))(*([b"GT"] * nv))
    fc, mc, co, cc = (np.ascontiguousarray(a, dtype=np.int32) for a in (father_col, mother_col, child_off, child_col))
    cs = np.ascontiguousarray(child_sex, dtype=np.uint8)
    x = None if chrom_is_x is None else np.ascontiguousarray(chrom_is_x, dtype=np.uint8)
    t1, t2 = np.zeros(nv, np.int32), np.zeros(nv, np.int32)
    lib().orc_tdt_text(flat, nv, ns, fmts, _p(x, C.c_uint8), len(fc), _p(fc, C.c_int32), _p(mc, C.c_int32),
                       _p(co, C.c_int32), _p(cc, C.c_int32), _p(cs, C.c_uint8), _p(t1, C.c_int32), _p(t2, C.c_int32))
    return t1, t2


# ---- epistasis / MDR (hpgv_epi_oracle.c) -------------------------------------------------------

def _rows(rows):
    rows = [np.ascontiguousarray(r, dtype=np.uint8) for r in rows]
    arr = (C.POINTER(C.c_uint8) * len(rows))(*[_p(r, C.c_uint8) for r in rows])
    return rows, arr


def epi_counts(rows, n_affected, n_unaffected):
    """combination_counts (model.c:76-124): rows = `order` genotype rows, cases first."""
    rows, arr = _rows(rows)
    cells = 3 ** len(rows)
    aff, unaff = np.zeros(cells, np.int32), np.zeros(cells, np.int32)
    lib().orc_epi_counts(len(rows), arr, n_affected, n_unaffected, _p(aff, C.c_int32), _p(unaff, C.c_int32))
    return aff, unaff


def epi_counts_all_folds(rows, n_affected, n_unaffected, fold_masks):
    """combination_counts_all_folds (model.c:126-206): fold_masks k x (nA+nU), 1 = training part."""
    rows, arr = _rows(rows)
    fm = np.ascontiguousarray(fold_masks, dtype=np.uint8)
    k, cells = fm.shape[0], 3 ** len(rows)
    aff, unaff = np.zeros((k, cells), np.int32), np.zeros((k, cells), np.int32)
    lib().orc_epi_counts_all_folds(len(rows), arr, n_affected, n_unaffected, _p(fm, C.c_uint8), k,
                                   _p(aff, C.c_int32), _p(unaff, C.c_int32))
    return aff, unaff


def mdr_high_risk(ca, cu, n_aff, n_unaff):
    return bool(lib().orc_mdr_high_risk(ca, cu, n_aff, n_unaff))


def mdr_high_risk2(ca, cu, n_aff, n_unaff):
    return bool(lib().orc_mdr_high_risk2(ca, cu, n_aff, n_unaff))


def epi_confusion(risky, rows, n_affected, n_unaffected, fold_mask, subset, sizes):
    """confusion_matrix (model.c:337-456): risky = list of cells (tuples of genotypes); subset 1 TRAINING, 0 TESTING."""
    rows, arr = _rows(rows)
    r = np.ascontiguousarray(np.array(risky, dtype=np.uint8).reshape(-1))
    fm = np.ascontiguousarray(fold_mask, dtype=np.uint8)
    sz = np.array(sizes, np.int32)
    m = np.zeros(4, np.uint32)
    lib().orc_epi_confusion(len(rows), _p(r, C.c_uint8), len(risky), arr, n_affected, n_unaffected, _p(fm, C.c_uint8),
                            subset, _p(sz, C.c_int32), _p(m, C.c_uint32))
    return [int(x) for x in m]


def epi_evaluate(matrix, function):
    lib().orc_epi_evaluate.restype = C.c_double
    m = np.array(matrix, np.uint32)
    return float(lib().orc_epi_evaluate(_p(m, C.c_uint32), function))


def epi_model(rows, n_affected, n_unaffected, fold_masks, subset):
    rows, arr = _rows(rows)
    fm = np.ascontiguousarray(fold_masks, dtype=np.uint8)
    k = fm.shape[0]
    acc, rm, mat = np.zeros(k, np.float64), np.zeros(k, np.uint32), np.zeros((k, 4), np.uint32)
    lib().orc_epi_model(len(rows), arr, n_affected, n_unaffected, _p(fm, C.c_uint8), k, subset,
                        _p(acc, C.c_double), _p(rm, C.c_uint32), _p(mat, C.c_uint32))
    return acc, rm, mat


def epi_model_wide(rows, n_affected, n_unaffected, fold_masks, subset, mask_words=8):
    """orc_epi_model_wide: any order; risky masks as (folds, mask_words) u32 (bit c % 32 of word c / 32 = cell c)."""
    rows, arr = _rows(rows)
    fm = np.ascontiguousarray(fold_masks, dtype=np.uint8)
    k = fm.shape[0]
    acc, rm, mat = np.zeros(k, np.float64), np.zeros((k, mask_words), np.uint32), np.zeros((k, 4), np.uint32)
    lib().orc_epi_model_wide(len(rows), arr, n_affected, n_unaffected, _p(fm, C.c_uint8), k, subset, mask_words,
                             _p(acc, C.c_double), _p(rm, C.c_uint32), _p(mat, C.c_uint32))
    return acc, rm, mat


def epi_scan_pairs(dataset, n_affected, n_unaffected, fold_masks, subset):
    """Every pair i < j in lexicographic order: accuracy[f][p], risky_mask[f][p]."""
    d = np.ascontiguousarray(dataset, dtype=np.uint8)
    fm = np.ascontiguousarray(fold_masks, dtype=np.uint8)
    v, k = d.shape[0], fm.shape[0]
    assert k <= 64 and d.shape[1] == n_affected + n_unaffected == fm.shape[1]
    n_pairs = v * (v - 1) // 2
    acc, rm = np.zeros((k, n_pairs), np.float64), np.zeros((k, n_pairs), np.uint32)
    lib().orc_epi_scan_pairs(_p(d, C.c_uint8), v, n_affected, n_unaffected, _p(fm, C.c_uint8), k, subset,
                             _p(acc, C.c_double), _p(rm, C.c_uint32))
    return acc, rm


def fold_masks_from_assignment(fold_of_sample, num_folds):
    """get_k_folds_masks (cross_validation.c:247-281) without the SSE padding: mask[f][s] = 0 when sample s
    is in the testing part of fold f, 1 otherwise."""
    f = np.asarray(fold_of_sample)
    return (f[None, :] != np.arange(num_folds)[:, None]).astype(np.uint8)
