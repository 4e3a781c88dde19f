#!/usr/bin/env python3
"""The text entry points on text that is already on the device (hpgv_text_alias, as after the GPU bgzip decoder):
hpgv_stats_text_groups with everything switched on (per-sample counters, Mendelian errors, phenotype groups),
hpgv_assoc_text, hpgv_tdt_text.  Diagnostic tool; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE
in its own pass) tools/pmc_sum.py adds the kernels' HBM traffic of one call up against text + matrix bytes.

  python tools/bench_text_entry.py [n_samples] [n_lines] [stats|assoc|tdt] [reps]
"""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 16_000
tool = sys.argv[3] if len(sys.argv) > 3 else "stats"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rng = np.random.default_rng(0)
codes = np.array(["0/0", "0/1", "1/1", "./."])
lines = []
for i in range(64):
    body = "\t".join(codes[rng.choice(4, size=n_samples, p=[0.5, 0.3, 0.19, 0.01])])
    lines.append("%d\t%d\trs%d\tA\tG\t.\tPASS\tAC=1;AN=2\tGT\t%s\n" % (1 + i % 22, 1000 + i, i, body))
n_lines = (n_lines // 64) * 64
text = ("".join(lines) * (n_lines // 64)).encode()
e = hpgv.Engine(0)
L = e.L
cond = (np.arange(n_samples) % 2).astype(np.uint8)
e.set_cohort(cond)
e.set_stats_cohort(n_samples)
nt = n_samples // 3
k = np.arange(nt)
e.set_families(n_samples, 3 * k, 3 * k + 1, np.arange(nt + 1), 3 * k + 2, (k % 2).astype(np.uint8))
e.set_pedigree(n_samples, 3 * k, 3 * k + 1, 3 * k + 2, (k % 2).astype(np.uint8))
n_groups = 3
e.set_stats_groups((np.arange(n_samples) % n_groups).astype(np.int32), n_groups)
# text on the device, a page-locked host buffer of the same size for the line heads
d_text = e.alloc(len(text) + 64)
e.h2d(d_text, np.frombuffer(text, np.uint8))
h_text = e.host_array((len(text) + 64,))
assert L.hpgv_text_alias(e.h, C.c_void_p(h_text.ctypes.data), d_text) == 0
m = n_lines
nl = C.c_int(0)
line_off = np.zeros(m + 2, np.uint64); field_off = np.zeros(m * 10, np.uint32); status = np.zeros(m, np.int32)
c8 = np.zeros(m * 8, np.int32); hw = np.zeros(2 * m); smiss = np.zeros(n_samples, np.int32); cerr = np.zeros(nt, np.int32)
merr = np.zeros(m, np.int32); midx = np.zeros(m, np.int32); mtab = np.zeros(256 * 64, np.int32)
gc8 = np.zeros(n_groups * m * 8, np.int32); ghw = np.zeros(2 * n_groups * m)
ints = np.zeros(4 * m, np.int32); dbl = np.zeros(3 * m)
p = lambda a: C.c_void_p(a.ctypes.data)


parts = os.environ.get("STATS_PARTS", "sm,me,ce,groups").split(",")      # diagnosis: which outputs are asked for


def call():
    if tool == "stats":
        nm = C.c_int(64)
        g = "groups" in parts
        rc = L.hpgv_stats_text_groups(e.h, C.cast(h_text.ctypes.data, C.c_char_p), len(text), m, C.byref(nl), p(line_off), p(field_off), p(status),
                                      p(c8), p(hw), p(hw[m:]), p(smiss) if "sm" in parts else None, p(midx), p(mtab), C.byref(nm),
                                      p(merr) if "me" in parts else None, p(cerr) if "ce" in parts else None,
                                      p(gc8) if g else None, p(ghw) if g else None, p(ghw[n_groups * m:]) if g else None)
    elif tool == "assoc":
        rc = L.hpgv_assoc_text(e.h, 1, C.cast(h_text.ctypes.data, C.c_char_p), len(text), m, C.byref(nl), p(line_off), p(field_off), p(status),
                               p(ints), p(ints[m:]), p(ints[2 * m:]), p(ints[3 * m:]), p(dbl), p(dbl[m:]), p(dbl[2 * m:]))
    else:
        rc = L.hpgv_tdt_text(e.h, C.cast(h_text.ctypes.data, C.c_char_p), len(text), m, C.byref(nl), p(line_off), p(field_off), p(status),
                             p(ints), p(ints[m:]), p(dbl), p(dbl[m:]), p(dbl[2 * m:]))
    assert rc == 0, L.hpgv_last_error(e.h)
    assert nl.value == n_lines


call()
t0 = time.perf_counter()
for _ in range(reps):
    call()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"tool": tool, "n_samples": n_samples, "n_lines": n_lines, "text_bytes": len(text), "matrix_bytes": n_lines * n_samples,
                  "calls": reps + 1, "fused": os.environ.get("HPGV_BATCH_FUSED", "1"), "parts": ",".join(parts), "ms_per_call": round(dt * 1e3, 3),
                  "text_GBps": round(len(text) / dt / 1e9, 1), "variants_per_s": round(n_lines / dt)}))
e.close()
