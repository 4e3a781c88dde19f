#!/usr/bin/env python3
"""Throughput of the any-order MDR ranking (hpgv_epi_rank_order: the listed-combination kernel k_epi_combs, one lane per cell of
the 3^order table): all C(V, order) combinations x N samples x k folds.  Diagnostic tool.
  python tools/bench_epistasis_order.py [order] [V] [N] [k]"""
import importlib
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")

order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
V = int(sys.argv[2]) if len(sys.argv) > 2 else 96
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
rng = np.random.default_rng(1)
nA = nU = N // 2
data = rng.choice(np.array([0, 1, 2, 255], np.uint8), size=(V, nA + nU), p=[0.5, 0.35, 0.14, 0.01])
fold = np.empty(nA + nU, np.int32)
fold[rng.permutation(nA)] = np.arange(nA) % K
fold[nA + rng.permutation(nU)] = np.arange(nU) % K
e = hpgv.Engine(0)
e.epi_set_dataset(data, nA, nU)
e.epi_set_folds(fold, K)
e.epi_rank_order(order, hpgv.EPI_TESTING, 10)
runs = []
for _ in range(2):
    t0 = time.perf_counter()
    res = e.epi_rank_order(order, hpgv.EPI_TESTING, 10)
    runs.append((time.perf_counter() - t0, res["scan_ms"]))
wall, scan_ms = min(runs)
combs = math.comb(V, order)
print(json.dumps({"order": order, "V": V, "samples": N, "folds": K, "combinations": combs, "wall_s": round(wall, 4), "scan_ms": round(scan_ms, 3),
                  "combinations_per_s": combs / (scan_ms * 1e-3), "cells_per_combination": 3 ** order,
                  "cell_samples_per_s": combs * 3 ** order * N / (scan_ms * 1e-3), "kernel": "k_epi_combs"}))
e.close()
