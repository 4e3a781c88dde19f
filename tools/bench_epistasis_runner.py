#!/usr/bin/env python3
"""The whole epistasis runner (hpgv_run_epistasis[_order] of libhpgv_host.so: dataset file in, k-fold cross-validation runs,
per-fold rankings merged, <prefix>.cv<r>.epi reports out; run_epistasis, singlenode/epistasis_runner.c) end to end.
  python tools/bench_epistasis_runner.py [V] [N] [k] [cv_runs] [order]
Diagnostic tool."""
import ctypes as C
import importlib
import json
import os
import struct
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
b = importlib.import_module("hpg-variant_amd._build")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ORDER = int(sys.argv[5]) if len(sys.argv) > 5 else 2
rng = np.random.default_rng(5)
nA = nU = N // 2
data = rng.choice(np.array([0, 1, 2, 255], np.uint8), size=(V, N), p=[0.5, 0.35, 0.14, 0.01])
L = C.CDLL(b.HOSTLIB)
L.hpgv_run_epistasis_order.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
L.hpgv_host_last_error.restype = C.c_char_p
assert L.hpgv_host_init(0) == 0
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "epi.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<III", V, nA, nU)); f.write(data.tobytes())
    out = {}
    for run in range(2):
        t0 = time.perf_counter()
        rc = L.hpgv_run_epistasis_order(path.encode(), ORDER, K, REPS, 10, 0, 1, os.path.join(d, "out").encode())
        assert rc == 0, L.hpgv_host_last_error()
        out["seconds_run_%d" % run] = round(time.perf_counter() - t0, 4)
combs = V * (V - 1) // 2 if ORDER == 2 else V * (V - 1) * (V - 2) // 6
print(json.dumps(dict(out, order=ORDER, V=V, samples=N, folds=K, cv_runs=REPS, dataset_MB=round(V * N / 1e6, 1),
                      combinations_x_cv_runs_per_s=combs * REPS / out["seconds_run_1"])))
