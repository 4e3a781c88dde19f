#!/usr/bin/env python3
"""Interleaved A/B of assoc-scan launch options in ONE process on ONE buffer
(cdna_hip_programming.md 5.4 rule 24): rounds x variants, median and min."""
import argparse
import importlib
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=int, default=1_000_000)
ap.add_argument("--samples", type=int, default=10_000)
ap.add_argument("--rounds", type=int, default=15)
ap.add_argument("--row-align", type=int, default=128)
ap.add_argument("--grid", type=str, default='{"scan_unroll":[4,8,10],"variants_per_wave":[2,4,8]}')
a = ap.parse_args()
grid = json.loads(a.grid)
keys = sorted(grid)
combos = [dict(zip(keys, c)) for c in itertools.product(*[grid[k] for k in keys])]

e = hpgv.Engine(0)
e.set_option("row_align", a.row_align)
e.set_option("profile", 1)
V, N = a.variants, a.samples
cond = (np.arange(N) % 2).astype(np.uint8)
nA, nU, pitch = e.set_cohort(cond)
d_gt = e.alloc(V * pitch)
d_counts = e.alloc(V * 16)
e.synth(hpgv.LAYOUT_ASSOC, 0, V, d_gt)
e.sync()
ts = {i: [] for i in range(len(combos))}
for r in range(a.rounds + 2):
    for i, c in enumerate(combos):
        for k, v in c.items():
            e.set_option(k, v)
        e.assoc_scan(d_gt, V, d_counts)
        ms, _ = e.last_kernel_ms()
        if r >= 2:
            ts[i].append(ms)
res = []
for i, c in enumerate(combos):
    med, mn = float(np.median(ts[i])), float(min(ts[i]))
    res.append((med, {"opts": c, "ms_med": round(med, 4), "ms_min": round(mn, 4),
                      "GBps_alg_med": round(V * (N + 40) / med / 1e6, 1), "frac": round(V * (N + 40) / med / 1e6 / 8000, 4)}))
for _, r in sorted(res, key=lambda t: t[0]):
    print(json.dumps(r), flush=True)
print(json.dumps({"pitch": pitch, "V": V, "N": N}))
e.close()
