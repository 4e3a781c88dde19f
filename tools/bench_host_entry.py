#!/usr/bin/env python3
"""Builds and runs tools/bench_host_entry.c: the latency of the synchronous per-batch entry points (hpgv_assoc,
hpgv_tdt, hpgv_stats) at the reference's batch size (200 variants, hpg-variant.conf:33) and larger, with the batch in
page-locked and in pageable host memory, with and without the fused per-batch kernel.  Diagnostic tool: these
PCIe-inclusive rates are never bench.py's `value`."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
importlib.import_module("hpg-variant_amd").build()
exe = "/tmp/bench_host_entry"
lib = os.path.join(ROOT, "hpg-variant_amd", "lib")
subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "bench_host_entry.c"),
                       "-L", lib, "-lhpgv", "-Wl,-rpath," + lib, "-lm", "-o", exe])
sys.exit(subprocess.call([exe] + sys.argv[1:]))
