#!/usr/bin/env python3
"""Builds and runs tools/bench_adapter.c: the reference-named per-batch functions (assoc_test, tdt_test, get_variants_stats)
on vcf_record_t batches with one heap string per sample, from the reference's worker shape (1 / 2 / 4 / 8 OpenMP workers +
a draining thread) and from a lone caller; per call the split staging / engine / records.  Arguments go to the binary
(--call assoc|fisher|tdt|stats, --samples, --batch, --seconds, --workers, --stage-only).  Host strings + PCIe inclusive:
a diagnostic of the drop-in boundary, never bench.py's value."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
importlib.import_module("hpg-variant_amd").build()
exe = "/tmp/bench_adapter"
lib = os.path.join(ROOT, "hpg-variant_amd", "lib")
subprocess.check_call(["gcc", "-O2", "-g", "-std=gnu99", "-fopenmp", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tools", "bench_adapter.c"), "-L", lib, "-lhpgv_host", "-lhpgv", "-Wl,-rpath," + lib,
                       "-lm", "-lpthread", "-o", exe])
sys.exit(subprocess.call([exe] + sys.argv[1:]))
