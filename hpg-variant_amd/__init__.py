"""hpg-variant_amd -- MI355X-native per-variant statistics engine for HPG Variant.

Python is plumbing here: this module only loads the C-ABI shared library
(include/hpgv.h) with ctypes and gives numpy-friendly wrappers for the tests
and bench.  There is no CPU implementation behind it: if the HIP library is
missing or no device is present, calls raise.

The directory name has a hyphen, so import it with
    importlib.import_module("hpg-variant_amd")
"""
import ctypes as C
import os

import numpy as np

from . import _build

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_UNSUPPORTED = 1, 2, 3, 4, 5, 6
TASK_CHISQ, TASK_FISHER = 1, 2
EPI_TESTING, EPI_TRAINING = 0, 1
COND_UNAFFECTED, COND_AFFECTED, COND_OTHER = 0, 1, 2
SEX_MALE, SEX_FEMALE, SEX_UNKNOWN = 0, 1, 2
LAYOUT_ASSOC, LAYOUT_TDT, LAYOUT_STATS, LAYOUT_STATS_GROUPS, LAYOUT_MENDEL, LAYOUT_EPI = 0, 1, 2, 3, 4, 5
GT_MISSING = 0xFF

# every symbol include/hpgv.h declares (checked by the CPU suite)
SYMBOLS = [
    "hpgv_version", "hpgv_device_count", "hpgv_create", "hpgv_create_multi", "hpgv_group_size", "hpgv_group_member", "hpgv_member_device", "hpgv_destroy", "hpgv_last_error",
    "hpgv_set_option", "hpgv_set_cohort", "hpgv_assoc_layout", "hpgv_set_families",
    "hpgv_tdt_layout", "hpgv_set_logfact", "hpgv_set_stats_cohort", "hpgv_stats_layout",
    "hpgv_set_stats_groups", "hpgv_stats_groups_layout", "hpgv_stats_scan_group_dev",
    "hpgv_set_pedigree", "hpgv_mendel_layout", "hpgv_mendel_scan_dev", "hpgv_mendel_children_dev",
    "hpgv_dev_alloc", "hpgv_dev_free", "hpgv_memcpy_h2d", "hpgv_memcpy_d2h", "hpgv_stream_sync",
    "hpgv_device_numa_node", "hpgv_memcpy_h2d_async", "hpgv_inflate_blocks_dev", "hpgv_bgzf_verify_dev", "hpgv_bgzf_scan_dev", "hpgv_bgzf_scan_scratch_bytes", "hpgv_dev_reserve", "hpgv_dev_commit", "hpgv_dev_committed", "hpgv_dev_release", "hpgv_text_alias", "hpgv_stream_create", "hpgv_stream_create_low", "hpgv_stream_destroy", "hpgv_host_alloc", "hpgv_host_free",
    "hpgv_layout_dev", "hpgv_synth_dev", "hpgv_synth_raw_dev",
    "hpgv_assoc_scan_dev", "hpgv_assoc_chisq_dev", "hpgv_assoc_fisher_dev",
    "hpgv_tdt_scan_dev", "hpgv_tdt_stats_dev", "hpgv_stats_scan_dev", "hpgv_stats_hwe_dev",
    "hpgv_sample_missing_dev", "hpgv_genotype_table_dev", "hpgv_stats_filter_dev",
    "hpgv_mendel", "hpgv_epi_dataset", "hpgv_tokenize_dev", "hpgv_tokenize", "hpgv_assoc_text", "hpgv_tdt_text",
    "hpgv_last_kernel_ms", "hpgv_assoc", "hpgv_tdt", "hpgv_stats", "hpgv_stats_ex", "hpgv_stats_groups",
    "hpgv_epi_dataset_text", "hpgv_set_text_filters", "hpgv_stats_text", "hpgv_stats_text_groups", "hpgv_epi_set_dataset", "hpgv_epi_set_folds", "hpgv_epi_set_fold_masks", "hpgv_epi_counts",
    "hpgv_epi_counts_all_folds", "hpgv_epi_scan_pairs", "hpgv_epi_rank_pairs", "hpgv_epi_rank_pairs_rows", "hpgv_epi_scan_triples", "hpgv_epi_rank_triples", "hpgv_epi_eval_combs", "hpgv_epi_rank_order", "hpgv_epi_rank_order_rows", "hpgv_read_probe",
    "hpgv_group_comm_init", "hpgv_group_comm_ranks", "hpgv_group_rccl_probe", "hpgv_group_shard", "hpgv_group_assoc", "hpgv_group_tdt", "hpgv_group_stats", "hpgv_group_sync", "hpgv_group_epi_share", "hpgv_group_epi_rank", "hpgv_epi_rank_triples_rows", "hpgv_text_alias_tiles", "hpgv_text_tiles_bytes", "hpgv_bgzf_verify_tiles_dev", "hpgv_memset_dev",
]


class HpgvError(RuntimeError):
    pass


_lib = None


def build(force=False, verbose=False):
    _build.build_all(force=force, verbose=verbose)


def lib_path():
    """libhpgv.so of this tree; HPGV_LIB names another build of it (tools/build_ablation.py: the forms that lost their A/B)."""
    return os.environ.get("HPGV_LIB") or _build.LIB


def load():
    """Loads libhpgv.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise HpgvError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                        "there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
    L.hpgv_version.restype = C.c_char_p
    L.hpgv_last_error.restype = C.c_char_p
    L.hpgv_last_error.argtypes = [vp]
    L.hpgv_create.argtypes = [i32, C.POINTER(vp)]
    L.hpgv_create_multi.argtypes = [vp, i32, C.POINTER(vp)]
    L.hpgv_group_size.argtypes = [vp]
    L.hpgv_group_member.argtypes = [vp, i32]
    L.hpgv_group_member.restype = vp
    L.hpgv_member_device.argtypes = [vp, i32]
    L.hpgv_destroy.argtypes = [vp]
    L.hpgv_destroy.restype = None
    L.hpgv_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.hpgv_set_cohort.argtypes = [vp, vp, i32]
    L.hpgv_assoc_layout.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(sz)]
    L.hpgv_set_families.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.hpgv_tdt_layout.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(sz)]
    L.hpgv_set_logfact.argtypes = [vp, vp, sz]
    L.hpgv_set_stats_cohort.argtypes = [vp, i32]
    L.hpgv_stats_layout.argtypes = [vp, C.POINTER(sz)]
    L.hpgv_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.hpgv_dev_free.argtypes = [vp, vp]
    L.hpgv_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.hpgv_host_free.argtypes = [vp, vp]
    L.hpgv_memcpy_h2d.argtypes = [vp, vp, vp, sz, vp]
    L.hpgv_memcpy_d2h.argtypes = [vp, vp, vp, sz, vp]
    L.hpgv_stream_sync.argtypes = [vp, vp]
    L.hpgv_layout_dev.argtypes = [vp, i32, vp, sz, i32, vp, vp]
    L.hpgv_synth_dev.argtypes = [vp, i32, u64, i32, vp, vp]
    L.hpgv_synth_raw_dev.argtypes = [vp, u64, i32, i32, sz, vp, vp]
    L.hpgv_assoc_scan_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.hpgv_assoc_chisq_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.hpgv_assoc_fisher_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.hpgv_tdt_scan_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.hpgv_tdt_stats_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.hpgv_stats_scan_dev.argtypes = [vp, vp, i32, vp, vp]
    L.hpgv_stats_hwe_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.hpgv_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hpgv_assoc.argtypes = [vp, i32, vp, sz, i32, vp] + [vp] * 7
    L.hpgv_tdt.argtypes = [vp, vp, sz, i32, vp] + [vp] * 5
    L.hpgv_stats.argtypes = [vp, vp, sz, i32, vp, vp, vp]
    L.hpgv_stats_ex.argtypes = [vp, vp, sz, i32, vp, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.hpgv_stats_groups.argtypes = [vp, vp, sz, i32, vp, vp, vp]
    L.hpgv_sample_missing_dev.argtypes = [vp, vp, i32, vp, vp]
    L.hpgv_genotype_table_dev.argtypes = [vp, vp, sz, i32, vp, i32, vp, vp]
    L.hpgv_tokenize_dev.argtypes = [vp, vp, sz, i32, i32, i32, vp, vp, vp, vp, sz, vp, vp, vp]
    L.hpgv_inflate_blocks_dev.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.hpgv_bgzf_verify_dev.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.hpgv_bgzf_verify_tiles_dev.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, u64, vp]
    L.hpgv_text_tiles_bytes.restype = sz
    L.hpgv_text_tiles_bytes.argtypes = [u64]
    L.hpgv_text_alias_tiles.argtypes = [vp, vp, vp, vp, vp, u64]
    L.hpgv_bgzf_scan_scratch_bytes.argtypes = [u64, i32]
    L.hpgv_bgzf_scan_scratch_bytes.restype = sz
    L.hpgv_bgzf_scan_dev.argtypes = [vp, vp, u64, u64, u64, i32, vp, vp, vp, vp, vp, sz, vp, vp]
    L.hpgv_dev_reserve.argtypes = [vp, sz, C.POINTER(vp)]
    L.hpgv_dev_commit.argtypes = [vp, vp, sz]
    L.hpgv_dev_release.argtypes = [vp, vp]
    L.hpgv_dev_committed.argtypes = [vp, vp, C.POINTER(sz)]
    L.hpgv_tokenize.argtypes = [vp, C.c_char_p, sz, i32, i32, i32, C.POINTER(i32), vp, vp, vp, sz, vp, vp]
    L.hpgv_assoc_text.argtypes = [vp, i32, C.c_char_p, sz, i32, C.POINTER(i32), vp, vp, vp] + [vp] * 7
    L.hpgv_tdt_text.argtypes = [vp, C.c_char_p, sz, i32, C.POINTER(i32), vp, vp, vp] + [vp] * 5
    L.hpgv_stats_filter_dev.argtypes = [vp, vp, i32, C.c_double, C.c_double, C.c_double, vp, vp]
    L.hpgv_set_stats_groups.argtypes = [vp, vp, i32, i32]
    L.hpgv_stats_groups_layout.argtypes = [vp, C.POINTER(sz), vp]
    L.hpgv_stats_scan_group_dev.argtypes = [vp, vp, i32, i32, vp, vp]
    L.hpgv_set_pedigree.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    L.hpgv_mendel_layout.argtypes = [vp, C.POINTER(sz)]
    L.hpgv_mendel_scan_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.hpgv_mendel_children_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.hpgv_mendel.argtypes = [vp, vp, sz, i32, vp, vp, vp]
    L.hpgv_epi_dataset.argtypes = [vp, vp, sz, i32, vp]
    L.hpgv_stats_text.argtypes = [vp, C.c_char_p, sz, i32, C.POINTER(i32), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32), vp, vp]
    L.hpgv_stats_text_groups.argtypes = [vp, C.c_char_p, sz, i32, C.POINTER(i32), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32), vp, vp, vp, vp, vp]
    L.hpgv_set_text_filters.argtypes = [vp, C.c_double, C.c_double, C.c_long]
    L.hpgv_epi_dataset_text.argtypes = [vp, C.c_char_p, sz, i32, C.POINTER(i32), vp, vp, vp, vp]
    L.hpgv_epi_set_dataset.argtypes = [vp, vp, i32, i32, i32]
    L.hpgv_epi_set_folds.argtypes = [vp, vp, i32]
    L.hpgv_epi_set_fold_masks.argtypes = [vp, vp, i32]
    L.hpgv_epi_counts.argtypes = [vp, i32, vp, i32, vp, vp]
    L.hpgv_epi_counts_all_folds.argtypes = [vp, i32, vp, i32, vp, vp]
    L.hpgv_epi_scan_pairs.argtypes = [vp, i32, i32, i32, vp, vp, C.POINTER(C.c_ulonglong)]
    L.hpgv_epi_rank_pairs.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.hpgv_epi_scan_triples.argtypes = [vp, i32, vp, vp]
    L.hpgv_epi_rank_triples.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.hpgv_epi_eval_combs.argtypes = [vp, i32, vp, i32, i32, vp, vp]
    L.hpgv_epi_rank_order.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.hpgv_epi_rank_order_rows.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.hpgv_epi_rank_pairs_rows.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.hpgv_read_probe.argtypes = [vp, vp, sz, i32, C.POINTER(C.c_float)]
    i64 = C.c_int64
    L.hpgv_group_comm_init.argtypes = [vp]
    L.hpgv_group_comm_ranks.argtypes = [vp]
    L.hpgv_group_rccl_probe.argtypes = [C.c_char_p, C.c_size_t]
    L.hpgv_group_shard.argtypes = [vp, i64, i32, C.POINTER(i64), C.POINTER(i64)]
    L.hpgv_group_assoc.argtypes = [vp, i32, vp, vp, i64, vp, vp, vp, vp]
    L.hpgv_group_tdt.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
    L.hpgv_group_stats.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.hpgv_group_sync.argtypes = [vp]
    L.hpgv_group_epi_share.argtypes = [vp, i32, i32, vp, vp]
    L.hpgv_group_epi_rank.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.hpgv_epi_rank_triples_rows.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    _lib = L
    return L


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Engine:
    """One engine context on one device (thin wrapper over hpgv_ctx)."""

    def __init__(self, device=0):
        """device: one device id, or a list of ids for a group context (hpgv_create_multi)."""
        self.L = load()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*device)
            rc = self.L.hpgv_create_multi(ids, len(device), C.byref(h))
        else:
            rc = self.L.hpgv_create(device, C.byref(h))
        if rc != OK:
            raise HpgvError("hpgv_create(%s) -> %d: %s" % (device, rc, self.L.hpgv_last_error(None).decode()))
        self.h = h
        self.device = device
        self._bufs = []
        self._views = []

    def close(self):
        if getattr(self, "h", None):
            # member views of a group hold their member's raw handle: they go first (their buffers are freed while the
            # member context is alive, and their handle is cleared so that a later close()/__del__ of the view is a no-op)
            for v in getattr(self, "_views", []):
                v.close()
            self._views = []
            for b in self._bufs:
                self.L.hpgv_dev_free(self.h, b)
            self._bufs = []
            for b in getattr(self, "_host_bufs", []):
                self.L.hpgv_host_free(self.h, b)
            self._host_bufs = []
            if not getattr(self, "_view", False):
                self.L.hpgv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise HpgvError("hpgv error %d: %s" % (rc, self.L.hpgv_last_error(self.h).decode()))

    def set_option(self, key, value):
        self._chk(self.L.hpgv_set_option(self.h, key.encode(), int(value)))

    # ---- cohort -----------------------------------------------------------
    def set_cohort(self, condition):
        c = _np(condition, np.uint8)
        self._chk(self.L.hpgv_set_cohort(self.h, _ptr(c), len(c)))
        return self.assoc_layout()

    def assoc_layout(self):
        a, u, p = C.c_int(), C.c_int(), C.c_size_t()
        self._chk(self.L.hpgv_assoc_layout(self.h, C.byref(a), C.byref(u), C.byref(p)))
        return a.value, u.value, p.value

    def set_families(self, n_samples, father_col, mother_col, child_off, child_col, child_sex):
        f, m = _np(father_col, np.int32), _np(mother_col, np.int32)
        o, c, s = _np(child_off, np.int32), _np(child_col, np.int32), _np(child_sex, np.uint8)
        assert len(o) == len(f) + 1 and len(c) == len(s)
        self._chk(self.L.hpgv_set_families(self.h, n_samples, len(f), _ptr(f), _ptr(m), _ptr(o),
                                           _ptr(c), _ptr(s)))
        return self.tdt_layout()

    def tdt_layout(self):
        a, b, p = C.c_int(), C.c_int(), C.c_size_t()
        self._chk(self.L.hpgv_tdt_layout(self.h, C.byref(a), C.byref(b), C.byref(p)))
        return a.value, b.value, p.value

    def set_logfact(self, table):
        t = _np(table, np.float64)
        self._chk(self.L.hpgv_set_logfact(self.h, _ptr(t), len(t)))

    def set_stats_cohort(self, n_samples):
        self._chk(self.L.hpgv_set_stats_cohort(self.h, n_samples))
        p = C.c_size_t()
        self._chk(self.L.hpgv_stats_layout(self.h, C.byref(p)))
        return p.value

    def set_stats_groups(self, group_of_sample, n_groups):
        g = _np(group_of_sample, np.int32)
        self._chk(self.L.hpgv_set_stats_groups(self.h, _ptr(g), len(g), n_groups))
        p = C.c_size_t()
        sizes = np.zeros(n_groups, np.int32)
        self._chk(self.L.hpgv_stats_groups_layout(self.h, C.byref(p), _ptr(sizes)))
        return p.value, sizes

    def stats_scan_group(self, d_gt, n_variants, group, d_counts8, stream=None):
        self._chk(self.L.hpgv_stats_scan_group_dev(self.h, d_gt, n_variants, group, d_counts8, stream))

    def set_pedigree(self, n_samples, father_col, mother_col, child_col, child_sex):
        f, m, c = _np(father_col, np.int32), _np(mother_col, np.int32), _np(child_col, np.int32)
        s = _np(child_sex, np.uint8)
        self._chk(self.L.hpgv_set_pedigree(self.h, n_samples, len(c), _ptr(f), _ptr(m), _ptr(c), _ptr(s)))
        p = C.c_size_t()
        self._chk(self.L.hpgv_mendel_layout(self.h, C.byref(p)))
        return p.value

    def mendel_scan(self, d_gt, n_variants, d_errors, d_is_x=None, stream=None):
        self._chk(self.L.hpgv_mendel_scan_dev(self.h, d_gt, n_variants, d_is_x, d_errors, stream))

    def mendel_children(self, d_gt, n_variants, d_child_errors, d_is_x=None, stream=None):
        self._chk(self.L.hpgv_mendel_children_dev(self.h, d_gt, n_variants, d_is_x, d_child_errors, stream))

    # ---- device memory ------------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.hpgv_dev_alloc(self.h, nbytes, C.byref(p)))
        self._bufs.append(p)
        return p

    def free(self, p):
        self._bufs = [b for b in self._bufs if b.value != p.value]
        self._chk(self.L.hpgv_dev_free(self.h, p))

    def host_array(self, shape, dtype=np.uint8):
        """numpy array in page-locked host memory (hpgv_host_alloc): batches staged into it are read by the
        per-batch kernels in place, without a copy.  Freed with the engine."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self.L.hpgv_host_alloc(self.h, max(n, 16), C.byref(p)))
        self._host_bufs = getattr(self, "_host_bufs", []) + [p]
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.hpgv_memcpy_h2d(self.h, dptr, _ptr(arr), arr.nbytes, None))

    def d2h(self, dptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._chk(self.L.hpgv_memcpy_d2h(self.h, _ptr(out), dptr, out.nbytes, None))
        return out

    def sync(self, stream=None):
        self._chk(self.L.hpgv_stream_sync(self.h, stream))

    # ---- per-batch host entry points ---------------------------------------
    def assoc(self, task, gt, is_x=None):
        gt = _np(gt, np.uint8)
        nv, pitch = gt.shape
        x = None if is_x is None else _np(is_x, np.uint8)
        A1, A2, U1, U2 = (np.zeros(nv, np.int32) for _ in range(4))
        odds, chisq, p = (np.zeros(nv, np.float64) for _ in range(3))
        self._chk(self.L.hpgv_assoc(self.h, task, _ptr(gt), pitch, nv, _ptr(x), _ptr(A1), _ptr(A2),
                                    _ptr(U1), _ptr(U2), _ptr(odds),
                                    _ptr(chisq) if task == TASK_CHISQ else None, _ptr(p)))
        return dict(A1=A1, A2=A2, U1=U1, U2=U2, odds=odds,
                    chisq=chisq if task == TASK_CHISQ else None, p=p)

    def assoc_view(self, task, gt2d, n_samples, is_x=None):
        """hpgv_assoc on a 2-D uint8 VIEW as it lies in memory (row stride = pitch): nothing is copied on the way,
        so a view of page-locked memory (host_array) is read by the kernel in place."""
        nv, pitch = gt2d.shape[0], gt2d.strides[0]
        assert gt2d.dtype == np.uint8 and gt2d.strides[1] == 1 and pitch >= n_samples
        A1, A2, U1, U2 = (np.zeros(nv, np.int32) for _ in range(4))
        odds, chisq, p = (np.zeros(nv, np.float64) for _ in range(3))
        self._chk(self.L.hpgv_assoc(self.h, task, C.c_void_p(gt2d.ctypes.data), pitch, nv,
                                    None if is_x is None else C.c_void_p(is_x.ctypes.data), _ptr(A1), _ptr(A2),
                                    _ptr(U1), _ptr(U2), _ptr(odds), _ptr(chisq) if task == TASK_CHISQ else None, _ptr(p)))
        return dict(A1=A1, A2=A2, U1=U1, U2=U2, odds=odds, chisq=chisq if task == TASK_CHISQ else None, p=p)

    def tdt_view(self, gt2d, n_samples, is_x=None):
        nv, pitch = gt2d.shape[0], gt2d.strides[0]
        assert gt2d.dtype == np.uint8 and gt2d.strides[1] == 1 and pitch >= n_samples
        t1, t2 = np.zeros(nv, np.int32), np.zeros(nv, np.int32)
        odds, chisq, p = (np.zeros(nv, np.float64) for _ in range(3))
        self._chk(self.L.hpgv_tdt(self.h, C.c_void_p(gt2d.ctypes.data), pitch, nv,
                                  None if is_x is None else C.c_void_p(is_x.ctypes.data), _ptr(t1), _ptr(t2),
                                  _ptr(odds), _ptr(chisq), _ptr(p)))
        return dict(t1=t1, t2=t2, odds=odds, chisq=chisq, p=p)

    def tdt(self, gt, is_x=None):
        gt = _np(gt, np.uint8)
        nv, pitch = gt.shape
        x = None if is_x is None else _np(is_x, np.uint8)
        t1, t2 = np.zeros(nv, np.int32), np.zeros(nv, np.int32)
        odds, chisq, p = (np.zeros(nv, np.float64) for _ in range(3))
        self._chk(self.L.hpgv_tdt(self.h, _ptr(gt), pitch, nv, _ptr(x), _ptr(t1), _ptr(t2),
                                  _ptr(odds), _ptr(chisq), _ptr(p)))
        return dict(t1=t1, t2=t2, odds=odds, chisq=chisq, p=p)

    def stats(self, gt):
        gt = _np(gt, np.uint8)
        nv, pitch = gt.shape
        c8 = np.zeros((nv, 8), np.int32)
        chi2, p = np.zeros(nv, np.float64), np.zeros(nv, np.float64)
        self._chk(self.L.hpgv_stats(self.h, _ptr(gt), pitch, nv, _ptr(c8), _ptr(chi2), _ptr(p)))
        return dict(counts8=c8, hwe_chi2=chi2, hwe_p=p)

    def stats_groups(self, gt, n_groups):
        gt = _np(gt, np.uint8)
        nv, pitch = gt.shape
        c8 = np.zeros((n_groups, nv, 8), np.int32)
        chi2, p = np.zeros((n_groups, nv), np.float64), np.zeros((n_groups, nv), np.float64)
        self._chk(self.L.hpgv_stats_groups(self.h, _ptr(gt), pitch, nv, _ptr(c8), _ptr(chi2), _ptr(p)))
        return dict(counts8=c8, hwe_chi2=chi2, hwe_p=p)

    # ---- epistasis / MDR ------------------------------------------------------
    def epi_set_dataset(self, genotypes, n_affected, n_unaffected):
        d = _np(genotypes, np.uint8)
        assert d.ndim == 2 and d.shape[1] == n_affected + n_unaffected
        self._chk(self.L.hpgv_epi_set_dataset(self.h, _ptr(d), d.shape[0], n_affected, n_unaffected))
        self._epi = (d.shape[0], n_affected, n_unaffected, 1)

    def epi_set_folds(self, fold_of_sample, num_folds):
        f = _np(fold_of_sample, np.int32)
        self._chk(self.L.hpgv_epi_set_folds(self.h, _ptr(f), num_folds))
        self._epi = self._epi[:3] + (num_folds,)

    def epi_set_fold_masks(self, padded_masks, num_folds):
        m = _np(padded_masks, np.uint8)
        self._chk(self.L.hpgv_epi_set_fold_masks(self.h, _ptr(m), num_folds))
        self._epi = self._epi[:3] + (num_folds,)

    def epi_counts(self, combs, all_folds=False):
        c = _np(combs, np.int32)
        n, order = c.shape
        cells, k = 3 ** order, self._epi[3]
        shape = (k, n, cells) if all_folds else (n, cells)
        aff, unaff = np.zeros(shape, np.int32), np.zeros(shape, np.int32)
        fn = self.L.hpgv_epi_counts_all_folds if all_folds else self.L.hpgv_epi_counts
        self._chk(fn(self.h, order, _ptr(c), n, _ptr(aff), _ptr(unaff)))
        return aff, unaff

    def epi_scan_pairs(self, subset, i_begin=0, i_end=None):
        i_end = self._epi[0] if i_end is None else i_end
        n = C.c_ulonglong(0)
        self._chk(self.L.hpgv_epi_scan_pairs(self.h, i_begin, i_end, subset, None, None, C.byref(n)))
        k = self._epi[3]
        acc, mask = np.zeros((k, n.value), np.float64), np.zeros((k, n.value), np.uint16)
        if n.value:
            self._chk(self.L.hpgv_epi_scan_pairs(self.h, i_begin, i_end, subset, _ptr(acc), _ptr(mask), C.byref(n)))
        return acc, mask

    def epi_rank_pairs(self, subset, max_ranking_size, rows=None):
        k, n = self._epi[3], max_ranking_size
        ci, cj = np.zeros((k, n), np.int32), np.zeros((k, n), np.int32)
        acc, mask, cnt = np.zeros((k, n), np.float64), np.zeros((k, n), np.uint32), np.zeros(k, np.int32)
        ms = C.c_float(0)
        lo, hi = (0, self._epi[0]) if rows is None else rows
        self._chk(self.L.hpgv_epi_rank_pairs_rows(self.h, lo, hi, subset, n, _ptr(ci), _ptr(cj), _ptr(acc), _ptr(mask), _ptr(cnt),
                                                  C.byref(ms)))
        return dict(i=ci, j=cj, accuracy=acc, risky=mask, n=cnt, scan_ms=ms.value)

    def epi_scan_triples(self, subset):
        v, k = self._epi[0], self._epi[3]
        acc, mask = np.zeros((k, v, v, v), np.float64), np.zeros((k, v, v, v), np.uint32)
        self._chk(self.L.hpgv_epi_scan_triples(self.h, subset, _ptr(acc), _ptr(mask)))
        return acc, mask

    def epi_rank_triples(self, subset, max_ranking_size):
        k, n = self._epi[3], max_ranking_size
        ci, cj, ck = (np.zeros((k, n), np.int32) for _ in range(3))
        acc, mask, cnt = np.zeros((k, n), np.float64), np.zeros((k, n), np.uint32), np.zeros(k, np.int32)
        ms = C.c_float(0)
        self._chk(self.L.hpgv_epi_rank_triples(self.h, subset, n, _ptr(ci), _ptr(cj), _ptr(ck), _ptr(acc), _ptr(mask), _ptr(cnt), C.byref(ms)))
        return dict(i=ci, j=cj, k=ck, accuracy=acc, risky=mask, n=cnt, scan_ms=ms.value)

    def epi_eval_combs(self, combs, subset):
        """the MDR model of listed combinations of any order 2 .. 5: accuracy (n, folds), risky masks (n, folds, 8) u32"""
        c = _np(combs, np.int32)
        n, order = c.shape
        k = self._epi[3]
        acc, mask = np.zeros((n, k), np.float64), np.zeros((n, k, 8), np.uint32)
        self._chk(self.L.hpgv_epi_eval_combs(self.h, order, _ptr(c), n, subset, _ptr(acc), _ptr(mask)))
        return acc, mask

    def epi_rank_order(self, order, subset, max_ranking_size, rows=None):
        k, n = self._epi[3], max_ranking_size
        combs = np.zeros((k, n, order), np.int32)
        acc, mask, cnt = np.zeros((k, n), np.float64), np.zeros((k, n, 8), np.uint32), np.zeros(k, np.int32)
        ms = C.c_float(0)
        lo, hi = (0, self._epi[0]) if rows is None else rows
        self._chk(self.L.hpgv_epi_rank_order_rows(self.h, order, lo, hi, subset, n, _ptr(combs), _ptr(acc), _ptr(mask), _ptr(cnt), C.byref(ms)))
        return dict(combs=combs, accuracy=acc, risky=mask, n=cnt, scan_ms=ms.value)

    def epi_dataset(self, gt):
        gt = _np(gt, np.uint8)
        nv, pitch = gt.shape
        nA, nU, _ = self.assoc_layout()
        out = np.zeros((nv, nA + nU), np.uint8)
        self._chk(self.L.hpgv_epi_dataset(self.h, _ptr(gt), pitch, nv, _ptr(out)))
        return out

    def mendel(self, gt, is_x=None, child_errors=None):
        gt = _np(gt, np.uint8)
        nv, pitch = gt.shape
        x = None if is_x is None else _np(is_x, np.uint8)
        err = np.zeros(nv, np.int32)
        self._chk(self.L.hpgv_mendel(self.h, _ptr(gt), pitch, nv, _ptr(x), _ptr(err), _ptr(child_errors)))
        return err

    def stats_ex(self, gt, sample_missing=None, multi_cap=0):
        """hpgv_stats_ex: counters + HWE, per-sample missing counts accumulated into
        `sample_missing`, and the 256-bin genotype tables of multi-allelic variants."""
        gt = _np(gt, np.uint8)
        nv, pitch = gt.shape
        c8 = np.zeros((nv, 8), np.int32)
        chi2, p = np.zeros(nv, np.float64), np.zeros(nv, np.float64)
        midx = np.zeros(max(multi_cap, 1), np.int32)
        mtab = np.zeros((max(multi_cap, 1), 256), np.int32)
        nm = C.c_int(multi_cap)
        self._chk(self.L.hpgv_stats_ex(self.h, _ptr(gt), pitch, nv, _ptr(c8), _ptr(chi2), _ptr(p),
                                       _ptr(sample_missing), _ptr(midx), _ptr(mtab), C.byref(nm)))
        k = min(nm.value, multi_cap)
        return dict(counts8=c8, hwe_chi2=chi2, hwe_p=p, n_multi=nm.value, multi_idx=midx[:k], multi_table=mtab[:k])

    def tokenize(self, text, n_samples, strict=True, max_lines=None):
        """VCF data lines (bytes) -> dict(gt [n_lines, n_samples], is_x, status, line_off, field_off, n_lines)."""
        if isinstance(text, str):
            text = text.encode()
        if max_lines is None:
            max_lines = text.count(b"\n") + 1
        pitch = max(n_samples, 1)
        gt = np.zeros((max_lines, pitch), np.uint8)
        is_x = np.zeros(max_lines, np.uint8)
        status = np.zeros(max_lines, np.int32)
        line_off = np.zeros(max_lines + 1, np.uint64)
        field_off = np.zeros((max_lines, 10), np.uint32)
        nl = C.c_int(0)
        self._chk(self.L.hpgv_tokenize(self.h, text, len(text), n_samples, 1 if strict else 0, max_lines, C.byref(nl),
                                       _ptr(line_off), _ptr(field_off), _ptr(gt), pitch, _ptr(is_x), _ptr(status)))
        k = min(nl.value, max_lines)
        return dict(n_lines=nl.value, gt=gt[:k, :n_samples], is_x=is_x[:k], status=status[:k], line_off=line_off[:k + 1],
                    field_off=field_off[:k])

    def assoc_text(self, task, text, max_lines=None):
        if isinstance(text, str):
            text = text.encode()
        if max_lines is None:
            max_lines = text.count(b"\n") + 1
        m = max(max_lines, 1)
        A1, A2, U1, U2 = (np.zeros(m, np.int32) for _ in range(4))
        odds, chisq, p = (np.zeros(m, np.float64) for _ in range(3))
        status = np.zeros(m, np.int32)
        nl = C.c_int(0)
        self._chk(self.L.hpgv_assoc_text(self.h, task, text, len(text), max_lines, C.byref(nl), None, None, _ptr(status),
                                         _ptr(A1), _ptr(A2), _ptr(U1), _ptr(U2), _ptr(odds),
                                         _ptr(chisq) if task == TASK_CHISQ else None, _ptr(p)))
        k = min(nl.value, max_lines)
        return dict(n_lines=nl.value, status=status[:k], A1=A1[:k], A2=A2[:k], U1=U1[:k], U2=U2[:k], odds=odds[:k],
                    chisq=chisq[:k] if task == TASK_CHISQ else None, p=p[:k])

    def tdt_text(self, text, max_lines=None):
        if isinstance(text, str):
            text = text.encode()
        if max_lines is None:
            max_lines = text.count(b"\n") + 1
        m = max(max_lines, 1)
        t1, t2 = np.zeros(m, np.int32), np.zeros(m, np.int32)
        odds, chisq, p = (np.zeros(m, np.float64) for _ in range(3))
        status = np.zeros(m, np.int32)
        nl = C.c_int(0)
        self._chk(self.L.hpgv_tdt_text(self.h, text, len(text), max_lines, C.byref(nl), None, None, _ptr(status),
                                       _ptr(t1), _ptr(t2), _ptr(odds), _ptr(chisq), _ptr(p)))
        k = min(nl.value, max_lines)
        return dict(n_lines=nl.value, status=status[:k], t1=t1[:k], t2=t2[:k], odds=odds[:k], chisq=chisq[:k], p=p[:k])

    # ---- device-resident path (raw pointers; ints or c_void_p) ---------------
    def synth(self, which, v0, n_variants, d_dst, stream=None):
        self._chk(self.L.hpgv_synth_dev(self.h, which, v0, n_variants, d_dst, stream))

    def synth_raw(self, v0, n_variants, n_samples, pitch, d_dst, stream=None):
        self._chk(self.L.hpgv_synth_raw_dev(self.h, v0, n_variants, n_samples, pitch, d_dst, stream))

    def layout(self, which, d_src, src_pitch, n_variants, d_dst, stream=None):
        self._chk(self.L.hpgv_layout_dev(self.h, which, d_src, src_pitch, n_variants, d_dst, stream))

    def assoc_scan(self, d_gt, n_variants, d_counts, d_is_x=None, stream=None):
        self._chk(self.L.hpgv_assoc_scan_dev(self.h, d_gt, n_variants, d_is_x, d_counts, stream))

    def assoc_chisq(self, d_counts, n_variants, d_odds, d_chisq, d_p, stream=None):
        self._chk(self.L.hpgv_assoc_chisq_dev(self.h, d_counts, n_variants, d_odds, d_chisq, d_p, stream))

    def assoc_fisher(self, d_counts, n_variants, d_odds, d_p, stream=None):
        self._chk(self.L.hpgv_assoc_fisher_dev(self.h, d_counts, n_variants, d_odds, d_p, stream))

    def inflate_blocks(self, d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status, stream=None):
        self._chk(self.L.hpgv_inflate_blocks_dev(self.h, d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status, stream))

    def bgzf_verify_tiles(self, d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status, d_tiles, n_tiles, stream=None):
        self._chk(self.L.hpgv_bgzf_verify_tiles_dev(self.h, d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status, d_tiles, n_tiles, stream))

    def bgzf_verify(self, d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status, stream=None):
        self._chk(self.L.hpgv_bgzf_verify_dev(self.h, d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status, stream))

    def bgzf_scan(self, d_comp, lo, hi, text_base, max_rows, d_in_off, d_in_len, d_out_off, d_out_len, stream=None):
        """Rows of the decoder's tables for the chain of bgzip blocks from byte lo that end by hi -> (rows, chain end, text end, headers seen)."""
        need = self.L.hpgv_bgzf_scan_scratch_bytes(hi - lo, max_rows)
        scratch = self.alloc(need)
        res = np.zeros(4, np.uint64)
        try:
            self._chk(self.L.hpgv_bgzf_scan_dev(self.h, d_comp, lo, hi, text_base, max_rows, d_in_off, d_in_len, d_out_off, d_out_len,
                                                scratch, need, _ptr(res), stream))
        finally:
            self.free(scratch)
        return tuple(int(x) for x in res)

    def dev_reserve(self, max_bytes):
        p = C.c_void_p()
        self._chk(self.L.hpgv_dev_reserve(self.h, max_bytes, C.byref(p)))
        return p

    def dev_commit(self, p, nbytes):
        self._chk(self.L.hpgv_dev_commit(self.h, p, nbytes))

    def dev_release(self, p):
        self._chk(self.L.hpgv_dev_release(self.h, p))

    def tdt_scan(self, d_gt, n_variants, d_tu, d_is_x=None, stream=None):
        self._chk(self.L.hpgv_tdt_scan_dev(self.h, d_gt, n_variants, d_is_x, d_tu, stream))

    def tdt_stats(self, d_tu, n_variants, d_odds, d_chisq, d_p, stream=None):
        self._chk(self.L.hpgv_tdt_stats_dev(self.h, d_tu, n_variants, d_odds, d_chisq, d_p, stream))

    def stats_scan(self, d_gt, n_variants, d_counts8, stream=None):
        self._chk(self.L.hpgv_stats_scan_dev(self.h, d_gt, n_variants, d_counts8, stream))

    def stats_hwe(self, d_counts8, n_variants, d_chi2, d_p, stream=None):
        self._chk(self.L.hpgv_stats_hwe_dev(self.h, d_counts8, n_variants, d_chi2, d_p, stream))

    def sample_missing(self, d_gt, n_variants, d_missing, stream=None):
        self._chk(self.L.hpgv_sample_missing_dev(self.h, d_gt, n_variants, d_missing, stream))

    def genotype_table(self, d_raw, src_pitch, n_samples, d_idx, n_idx, d_table, stream=None):
        self._chk(self.L.hpgv_genotype_table_dev(self.h, d_raw, src_pitch, n_samples, d_idx, n_idx, d_table, stream))

    def stats_filter(self, d_counts8, n_variants, d_keep, min_maf=-1.0, max_maf=-1.0, max_missing=-1.0, stream=None):
        self._chk(self.L.hpgv_stats_filter_dev(self.h, d_counts8, n_variants, min_maf, max_maf, max_missing, d_keep, stream))

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._chk(self.L.hpgv_last_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- group context: the variant-sharded resident scan (hpgv_group_*) ---------------------------------
    def group_size(self):
        return self.L.hpgv_group_size(self.h)

    def member(self, i):
        """Engine view of member i of a group context (its device memory, generator, *_dev calls); owned by the group."""
        h = self.L.hpgv_group_member(self.h, i)
        if not h:
            raise HpgvError("no member %d" % i)
        m = Engine.__new__(Engine)
        m.L, m.h, m.device, m._bufs, m._host_bufs, m._view = self.L, C.c_void_p(h), self.L.hpgv_member_device(self.h, i), [], [], True
        m._views = []
        m._parent = self                # the view keeps its group alive; the group's close() closes its views first
        self._views.append(m)
        return m

    def group_comm_init(self):
        self._chk(self.L.hpgv_group_comm_init(self.h))
        return self.L.hpgv_group_comm_ranks(self.h)

    def group_shard(self, n_variants, member):
        lo, hi = C.c_int64(), C.c_int64()
        self._chk(self.L.hpgv_group_shard(self.h, n_variants, member, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    @staticmethod
    def _ptr_array(ptrs):
        if ptrs is None:
            return None
        return (C.c_void_p * len(ptrs))(*[p.value if isinstance(p, C.c_void_p) else p for p in ptrs])

    def group_assoc(self, task, d_gt, n_variants, d_counts, d_odds, d_chisq, d_p, d_is_x=None):
        self._chk(self.L.hpgv_group_assoc(self.h, task, self._ptr_array(d_gt), self._ptr_array(d_is_x), n_variants,
                                          d_counts, d_odds, d_chisq, d_p))

    def group_tdt(self, d_gt, n_variants, d_tu, d_odds, d_chisq, d_p, d_is_x=None):
        self._chk(self.L.hpgv_group_tdt(self.h, self._ptr_array(d_gt), self._ptr_array(d_is_x), n_variants, d_tu, d_odds, d_chisq, d_p))

    def group_stats(self, d_gt, n_variants, d_counts8, d_chi2, d_p, d_sample_missing=None):
        self._chk(self.L.hpgv_group_stats(self.h, self._ptr_array(d_gt), n_variants, d_counts8, d_chi2, d_p, d_sample_missing))

    def group_epi_share(self, order, member):
        lo, hi = C.c_int(), C.c_int()
        self._chk(self.L.hpgv_group_epi_share(self.h, order, member, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def group_epi_rank(self, order, subset, max_ranking_size):
        k, n = self._epi[3], max_ranking_size
        combs = np.zeros((k, n, order), np.int32)
        acc, mask, cnt = np.zeros((k, n), np.float64), np.zeros((k, n, 8), np.uint32), np.zeros(k, np.int32)
        ms = C.c_float(0)
        self._chk(self.L.hpgv_group_epi_rank(self.h, order, subset, n, _ptr(combs), _ptr(acc), _ptr(mask), _ptr(cnt), C.byref(ms)))
        return dict(combs=combs, accuracy=acc, risky=mask, n=cnt, scan_ms=ms.value)

    def group_sync(self):
        self._chk(self.L.hpgv_group_sync(self.h))

    def read_probe(self, d_buf, nbytes, iters=5):
        ms = C.c_float()
        self._chk(self.L.hpgv_read_probe(self.h, d_buf, nbytes, iters, C.byref(ms)))
        return ms.value
