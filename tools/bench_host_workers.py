#!/usr/bin/env python3
"""Builds and runs tools/bench_host_workers.c: hpgv_assoc from 1 / 2 / 4 / 8 concurrent worker threads (the reference's
num_threads, assoc_runner.c:106-112) on one context and on a [0, 0] group.  PCIe-inclusive diagnostic, never bench.py's value."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
importlib.import_module("hpg-variant_amd").build()
exe = "/tmp/bench_host_workers"
lib = os.path.join(ROOT, "hpg-variant_amd", "lib")
subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "bench_host_workers.c"),
                       "-L", lib, "-lhpgv", "-Wl,-rpath," + lib, "-lm", "-o", exe])
sys.exit(subprocess.call([exe] + sys.argv[1:]))
