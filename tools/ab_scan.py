#!/usr/bin/env python3
"""Interleaved A/B of scan launch options for any of the scans, in ONE process
on ONE buffer (cdna_hip_programming.md 5.4 rule 24).  Diagnostic tool.

  --kind assoc|tdt|stats|fisher
"""
import argparse
import importlib
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")
from oracle import pyoracle as orc  # noqa: E402  (only for the log-factorial table)

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="assoc")
ap.add_argument("--variants", type=int, default=1_000_000)
ap.add_argument("--samples", type=int, default=10_000)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--row-align", type=int, default=16)
ap.add_argument("--grid", type=str, default='{"variants_per_wave":[2]}')
a = ap.parse_args()
grid = json.loads(a.grid)
keys = sorted(grid)
combos = [dict(zip(keys, c)) for c in itertools.product(*[grid[k] for k in keys])]

e = hpgv.Engine(0)
e.set_option("row_align", a.row_align)
e.set_option("profile", 1)
V, N = a.variants, a.samples
if a.kind in ("assoc", "fisher"):
    cond = (np.arange(N) % 2).astype(np.uint8)
    _, _, pitch = e.set_cohort(cond)
    which, out_bytes, payload = hpgv.LAYOUT_ASSOC, 16, 40
elif a.kind == "tdt":
    n_tr = N // 3
    k = np.arange(n_tr)
    _, _, pitch = e.set_families(3 * n_tr, 3 * k, 3 * k + 1, np.arange(n_tr + 1), 3 * k + 2, (k % 2).astype(np.uint8))
    which, out_bytes, payload = hpgv.LAYOUT_TDT, 8, 32
else:
    pitch = e.set_stats_cohort(N)
    which, out_bytes, payload = hpgv.LAYOUT_STATS, 32, 48
d_gt = e.alloc(V * pitch)
d_out = e.alloc(V * out_bytes)
d_f = e.alloc(V * 24)
e.synth(which, 0, V, d_gt)
e.sync()
if a.kind == "fisher":
    e.set_logfact(orc.logfact(N * 10))
    e.assoc_scan(d_gt, V, d_out)
ts = {i: [] for i in range(len(combos))}
for r in range(a.rounds + 2):
    for i, c in enumerate(combos):
        for kk, v in c.items():
            e.set_option(kk, v)
        if a.kind == "assoc":
            e.assoc_scan(d_gt, V, d_out)
            ms, _ = e.last_kernel_ms()
        elif a.kind == "tdt":
            e.tdt_scan(d_gt, V, d_out)
            ms, _ = e.last_kernel_ms()
        elif a.kind == "stats":
            e.stats_scan(d_gt, V, d_out)
            ms, _ = e.last_kernel_ms()
        else:
            e.assoc_fisher(d_out, V, d_f.value, d_f.value + 8 * V)
            _, ms = e.last_kernel_ms()
        if r >= 2:
            ts[i].append(ms)
res = []
for i, c in enumerate(combos):
    med, mn = float(np.median(ts[i])), float(min(ts[i]))
    gb = V * (N + payload) / med / 1e6
    res.append((med, {"kind": a.kind, "opts": c, "ms_med": round(med, 4), "ms_min": round(mn, 4),
                      "Mvariants_per_s": round(V / med / 1e3, 1), "GBps_alg_med": round(gb, 1), "frac": round(gb / 8000, 4)}))
for _, r in sorted(res, key=lambda t: t[0]):
    print(json.dumps(r), flush=True)
print(json.dumps({"pitch": pitch, "V": V, "N": N}))
e.close()
