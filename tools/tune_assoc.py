#!/usr/bin/env python3
"""Sweeps the assoc scan's launch knobs on one GPU and prints achieved GB/s
(algorithmic bytes = V x (N + 40)).  Diagnostic tool, not part of the bench.

  python tools/tune_assoc.py --grid '{"scan_unroll":[8,10],"persistent":[0,1]}' [--variants ..]
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


def run(V, N, opts, iters=12, probe=False):
    e = hpgv.Engine(0)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_option("profile", 1)
    cond = (np.arange(N) % 2).astype(np.uint8)
    nA, nU, pitch = e.set_cohort(cond)
    d_gt = e.alloc(V * pitch)
    d_counts = e.alloc(V * 16)
    e.synth(hpgv.LAYOUT_ASSOC, 0, V, d_gt)
    e.sync()
    ts = []
    for i in range(iters + 3):
        e.assoc_scan(d_gt, V, d_counts)
        ms, _ = e.last_kernel_ms()
        if i >= 3:
            ts.append(ms)
    out = {"V": V, "N": N, "opts": opts, "pitch": pitch, "ms_med": round(float(np.median(ts)), 4),
           "ms_min": round(float(min(ts)), 4), "ms_max": round(float(max(ts)), 4)}
    out["GBps_alg_med"] = round(V * (N + 40) / out["ms_med"] / 1e6, 1)
    out["frac_of_8TBps"] = round(out["GBps_alg_med"] / 8000, 4)
    if probe:
        pm = e.read_probe(d_gt, V * pitch, 5)
        out["probe_GBps"] = round(V * pitch / pm / 1e6, 1)
    e.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=str, default="1000000")
    ap.add_argument("--samples", type=int, default=10_000)
    ap.add_argument("--grid", type=str, default='{"nontemporal":[1,0],"variants_per_wave":[4]}')
    ap.add_argument("--probe", action="store_true")
    a = ap.parse_args()
    grid = json.loads(a.grid)
    keys = sorted(grid)
    for V in [int(x) for x in a.variants.split(",")]:
        for combo in itertools.product(*[grid[k] for k in keys]):
            print(json.dumps(run(V, a.samples, dict(zip(keys, combo)), probe=a.probe)), flush=True)
