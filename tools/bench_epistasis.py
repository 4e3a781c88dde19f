#!/usr/bin/env python3
"""Throughput of the epistasis / MDR pair scan (hpgv_epi_rank_pairs): all V(V-1)/2 pairs x N samples x k folds.
Bound: VALU integer issue -- 18 operations (9 v_and_b32 + 9 v_bcnt_u32_b32) per pair and 32 samples; peak =
256 CUs x 64 lanes x clock.  Diagnostic tool (not the headline metric).

  python tools/bench_epistasis.py [V] [N] [k] [--cpu] [--complete] [--order=3] [--unbalanced] [--affected=n] [--option=key=value]

--complete: a dataset without missing calls (the scan then counts four cells per pair and derives the other five:
8 operations per pair and word; the roofline line keeps the 18 of the general case so that the two are comparable).
"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")

args = [a for a in sys.argv[1:] if not a.startswith("--")]
V = int(args[0]) if len(args) > 0 else 4096
N = int(args[1]) if len(args) > 1 else 10000
K = int(args[2]) if len(args) > 2 else 10
CLOCK_GHZ = 2.4
rng = np.random.default_rng(1)
nA = nU = N // 2
if "--unbalanced" in sys.argv:                              # cases != controls: the threshold of a high-risk cell is a ratio, not 1
    nA = (2 * N) // 5; nU = N - nA
for a in sys.argv[1:]:
    if a.startswith("--affected="):                         # any split, e.g. one whose ratio never ties a cell (4987 of 10000)
        nA = int(a[len("--affected="):]); nU = N - nA
COMPLETE = "--complete" in sys.argv
data = rng.choice(np.array([0, 1, 2, 255], np.uint8), size=(V, nA + nU), p=[0.5, 0.35, 0.15, 0.0] if COMPLETE else [0.5, 0.35, 0.14, 0.01])
fold = np.empty(nA + nU, np.int32)
fold[rng.permutation(nA)] = np.arange(nA) % K
fold[nA + rng.permutation(nU)] = np.arange(nU) % K
e = hpgv.Engine(0)
for kv in [a[len('--option='):] for a in sys.argv[1:] if a.startswith('--option=')]:
    e.set_option(kv.split('=')[0], int(kv.split('=')[1]))
t0 = time.perf_counter()
e.epi_set_dataset(data, nA, nU)
e.epi_set_folds(fold, K)
t_setup = time.perf_counter() - t0
ORDER = 3 if "--order=3" in sys.argv else 2
if ORDER == 3:
    # order 3: V(V-1)(V-2)/6 triples; 27 cells x (one three-input AND, v_bitop3_b32, + one v_bcnt_u32_b32) = 54 operations
    # per triple and 32 samples (hpgv_epi_kernels.h: the one-pass kernel; the two-pass form used above 10 folds does 108)
    e.epi_rank_triples(hpgv.EPI_TESTING, 10)
    runs = []
    for _ in range(2):
        t0 = time.perf_counter()
        res = e.epi_rank_triples(hpgv.EPI_TESTING, 10)
        runs.append((time.perf_counter() - t0, res["scan_ms"]))
    wall, scan_ms = min(runs)
    triples = V * (V - 1) * (V - 2) // 6
    words = (-(-(nA // K) // 128) + -(-(nU // K) // 128)) * 4 * K
    wave_instr = triples * words * 54 / 64
    peak = 256 * 4 * CLOCK_GHZ * 1e9 / 4
    line = {"order": 3, "V": V, "samples": N, "affected": nA, "folds": K, "triples": triples, "wall_s": round(wall, 4), "scan_ms": round(scan_ms, 3),
            "triples_per_s": triples / (scan_ms * 1e-3),
            "roofline": {"bound": "valu", "achieved": wave_instr / (scan_ms * 1e-3) / 1e9, "peak": peak / 1e9,
                         "unit": "G wave64 VALU instructions/s", "frac": wave_instr / (scan_ms * 1e-3) / peak,
                         "algorithmic_ops_per_triple_word": 54}}
    opts = dict(a[len('--option='):].split('=') for a in sys.argv[1:] if a.startswith('--option='))
    if int(opts.get("epi_triples_mfma", 1)) and K <= 16:
        # the matrix-core scan (k_epi_triples_mfma): 27 cells x samples (padded) x 2 flop per triple are the algorithm's; dense FP4
        # peak from MI355X_MICROARCH.md (see the pair line's note on the issue rate measured here)
        flops = triples * 27 * words * 32 * 2
        line["kernel"] = "k_epi_triples_mfma"
        line["roofline_vector_scan_equivalent"] = line["roofline"]
        line["roofline"] = {"bound": "mfma", "achieved": flops / (scan_ms * 1e-3) / 1e12, "peak": 10000.0, "unit": "TFLOP/s (FP4, dense)",
                            "frac": flops / (scan_ms * 1e-3) / 1e16, "algorithmic_flop_per_triple_and_sample": 54}
    else:
        line["kernel"] = "k_epi_triples3" if K <= 10 and int(opts.get("epi_triples_1pass", 1)) else "k_epi_triples"
    print(json.dumps(line))
    e.close()
    sys.exit(0)
e.epi_rank_pairs(hpgv.EPI_TESTING, 10)                        # warm
runs = []
for _ in range(3):
    t0 = time.perf_counter()
    res = e.epi_rank_pairs(hpgv.EPI_TESTING, 10)
    runs.append((time.perf_counter() - t0, res["scan_ms"]))
wall, scan_ms = min(runs)
pairs = V * (V - 1) // 2
words = (-(-(nA // K) // 128) + -(-(nU // K) // 128)) * 4 * K   # words per row after padding every (fold, class) run to 4 words
ops = pairs * words * 18
out = {"V": V, "samples": N, "affected": nA, "folds": K, "missing_calls": not COMPLETE, "pairs": pairs, "setup_s": round(t_setup, 3), "wall_s": round(wall, 4), "scan_ms": round(scan_ms, 3),
       "pairs_per_s": pairs / (scan_ms * 1e-3), "pair_samples_per_s": pairs * N / (scan_ms * 1e-3),
       "valu_lane_ops_per_s": ops * 32 / (scan_ms * 1e-3) / 32 * 1.0,
       "roofline": {"bound": "valu", "achieved_Tops": ops / (scan_ms * 1e-3) / 1e12 * 64 / 64,
                    "peak_Tops": 256 * 4 * CLOCK_GHZ * 1e9 / 4 / 1e12 * 4, "unit": "T wave-instructions x 64 lanes /s"}}
# peak: each SIMD issues one wave64 VALU instruction per 4 cycles: 256 CUs x 4 SIMDs x 2.4e9 / 4 = 614 G wave-instr/s
peak_wave_instr = 256 * 4 * CLOCK_GHZ * 1e9 / 4
wave_instr = ops / 64
out["roofline"] = {"bound": "valu", "achieved": wave_instr / (scan_ms * 1e-3) / 1e9, "peak": peak_wave_instr / 1e9,
                   "unit": "G wave64 VALU instructions/s", "frac": wave_instr / (scan_ms * 1e-3) / peak_wave_instr,
                   "algorithmic_ops_per_pair_word": 18}
opts = dict(a[len('--option='):].split('=') for a in sys.argv[1:] if a.startswith('--option='))
if int(opts.get("epi_pairs_mfma", 1)):                        # (data without missing calls takes the matrix-core scan too)
    # the matrix-core scan (hpgv_epi_mfma_kernels.h): 9 cells x samples (padded to 128 per group) x 2 flop per pair are the
    # algorithm's; the kernel walks the samples twice.  Peak: dense FP4 (MI355X_MICROARCH.md); the kernel is vector-issue
    # bound (16.7 k vector instructions per 256 pairs) and v_mfma_scale_f32_16x16x128_f8f6f4 issues every 33 cycles here
    # back to back (tools/exp/mfma_rate.hip), about half the rate that peak assumes
    flops = pairs * 9 * words * 32 * 2
    out["kernel"] = "k_epi_pairs_mfma"
    out["roofline_vector_scan_equivalent"] = out["roofline"]
    out["roofline"] = {"bound": "mfma", "achieved": flops / (scan_ms * 1e-3) / 1e12, "peak": 10000.0, "unit": "TFLOP/s (FP4, dense)",
                       "frac": flops / (scan_ms * 1e-3) / 1e16, "algorithmic_flop_per_pair_and_sample": 18}
else:
    out["kernel"] = "k_epi_pairs"
if COMPLETE and out["kernel"] == "k_epi_pairs":
    out["roofline"]["note"] = "complete data: 8 operations per pair and word are executed; frac is quoted at the general case's 18"
    out["roofline"]["frac_at_8_ops"] = out["roofline"]["frac"] * 8 / 18
del out["valu_lane_ops_per_s"]
if "--cpu" in sys.argv:
    from oracle import pyoracle as orc
    vs = 48
    masks = orc.fold_masks_from_assignment(fold, K)
    t0 = time.perf_counter()
    orc.epi_scan_pairs(data[:vs], nA, nU, masks, 0)
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": vs * (vs - 1) // 2 / dt, "unit": "pairs/s", "kind": "port", "cores": orc.effective_cpus(),
                           "sample": "oracle scan of the first %d SNPs (%d pairs) x %d samples x %d folds, OpenMP over rows, %.1f s" % (vs, vs * (vs - 1) // 2, N, K, dt)}
print(json.dumps(out))
e.close()
