#!/usr/bin/env python3
"""Sustained (back-to-back, no host sync) A/B of assoc-scan options: each arm is
K launches queued at once, timed by one event pair; arms are interleaved over
rounds in ONE process.  This is the regime bench.py runs in."""
import argparse
import importlib
import itertools
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=int, default=1_000_000)
ap.add_argument("--samples", type=int, default=10_000)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--aligns", type=str, default="16,128")
ap.add_argument("--pads", type=str, default="0")
ap.add_argument("--grid", type=str, default='{"scan_unroll":[4,8],"variants_per_wave":[2,4]}')
ap.add_argument("--with-chisq", action="store_true")
a = ap.parse_args()
grid = json.loads(a.grid)
keys = sorted(grid)
combos = [dict(zip(keys, c)) for c in itertools.product(*[grid[k] for k in keys])]
V, N = a.variants, a.samples
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
sp = stream.cuda_stream
arms = []
for al, pad in [(int(x), int(y)) for x in a.aligns.split(",") for y in a.pads.split(",")]:
    e = hpgv.Engine(0)
    e.set_option("row_align", al)
    e.set_option("row_pad", pad)
    cond = (np.arange(N) % 2).astype(np.uint8)
    _, _, pitch = e.set_cohort(cond)
    gt = torch.empty(V * pitch, dtype=torch.uint8, device=dev)
    res = torch.empty(V * 40, dtype=torch.uint8, device=dev)
    e.synth(hpgv.LAYOUT_ASSOC, 0, V, gt.data_ptr(), sp)
    torch.cuda.synchronize()
    for c in combos:
        arms.append((al, c, e, gt, res, pitch))
ts = {i: [] for i in range(len(arms))}
for r in range(a.rounds + 1):
    for i, (al, c, e, gt, res, pitch) in enumerate(arms):
        for k, v in c.items():
            e.set_option(k, v)
        b = res.data_ptr()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record(stream)
        for _ in range(a.k):
            e.assoc_scan(gt.data_ptr(), V, b, None, sp)
            if a.with_chisq:
                e.assoc_chisq(b, V, b + 16 * V, b + 24 * V, b + 32 * V, sp)
        ev1.record(stream)
        torch.cuda.synchronize()
        if r >= 1:
            ts[i].append(ev0.elapsed_time(ev1) / a.k)
out = []
for i, (al, c, e, gt, res, pitch) in enumerate(arms):
    med = float(np.median(ts[i]))
    out.append((med, {"row_align": al, "opts": c, "pitch": pitch, "ms_per_launch_med": round(med, 4),
                      "ms_min": round(float(min(ts[i])), 4), "GBps_alg": round(V * (N + 40) / med / 1e6, 1),
                      "frac": round(V * (N + 40) / med / 1e6 / 8000, 4)}))
for _, o in sorted(out, key=lambda t: t[0]):
    print(json.dumps(o), flush=True)
