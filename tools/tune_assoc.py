#!/usr/bin/env python3
"""Sweeps the assoc scan's launch knobs on one GPU and prints achieved GB/s
(algorithmic bytes = V x (N + 40)).  Diagnostic tool, not part of the bench."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")


def run(V, N, opts, iters=10):
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
    for i in range(iters + 2):
        e.assoc_scan(d_gt, V, d_counts)
        ms, _ = e.last_kernel_ms()
        if i >= 2:
            ts.append(ms)
    probe = e.read_probe(d_gt, V * pitch, 5)
    e.close()
    ms = float(np.median(ts))
    return {"opts": opts, "pitch": pitch, "scan_ms_med": ms, "scan_ms_min": float(min(ts)),
            "GBps_alg": V * (N + 40) / ms / 1e6, "GBps_raw": V * pitch / ms / 1e6,
            "probe_ms": probe, "probe_GBps": V * pitch / probe / 1e6}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=1_000_000)
    ap.add_argument("--samples", type=int, default=10_000)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    grid = []
    for nt in (1, 0):
        for vpw in ((4,) if a.quick else (1, 2, 4, 8, 16)):
            grid.append({"nontemporal": nt, "variants_per_wave": vpw})
    if not a.quick:
        grid.append({"nontemporal": 1, "variants_per_wave": 4, "row_align": 16})
        grid.append({"nontemporal": 1, "variants_per_wave": 4, "row_align": 256})
    for g in grid:
        print(json.dumps(run(a.variants, a.samples, g)), flush=True)
