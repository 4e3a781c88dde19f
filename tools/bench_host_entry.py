#!/usr/bin/env python3
"""PCIe-inclusive rate of the synchronous host entry point hpgv_assoc (host
matrix in, host results out) at several batch sizes.  Diagnostic tool: this
rate is never bench.py's `value`."""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")
N = 10_000
e = hpgv.Engine(0)
cond = (np.arange(N) % 2).astype(np.uint8)
e.set_cohort(cond)
rng = np.random.default_rng(0)
base = np.array([0x00, 0x01, 0x11, 0xFF], np.uint8)
for batch in (200, 2000, 20000, 100000):
    gt = base[rng.choice(4, size=(batch, N), p=[0.5, 0.3, 0.19, 0.01])]
    e.assoc(hpgv.TASK_CHISQ, gt)
    reps = max(3, 200000 // batch)
    t0 = time.perf_counter()
    for _ in range(reps):
        e.assoc(hpgv.TASK_CHISQ, gt)
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"batch_variants": batch, "samples": N, "ms_per_call": round(dt * 1e3, 3),
                      "variants_per_s": round(batch / dt), "host_to_device_GBps": round(batch * N / dt / 1e9, 2)}))
e.close()
