#!/usr/bin/env python3
"""Per-launch averages of PMC counters per kernel -> profiles/<tag>_pmc_valu_<workload>.json (bench.py reads
`valu_insts_per_variant` of k_assoc_fisher from it for the c3 roofline).

  python tools/pmc_valu.py <workload> <bench_line.json> <out.json> <counter_collection.csv> [...]

SQ_INSTS_* count wave instructions summed over the dispatch; FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE doubled on
gfx950 for wide coalesced streams, MI355X_MICROARCH.md HBM section -- the Fisher kernel's reads are 8-byte table
look-ups and 16-byte count records, so its FETCH_SIZE is reported raw AND doubled)."""
import csv
import json
import sys

KERNELS = ("k_assoc_scan", "k_assoc_chisq", "k_assoc_fisher", "k_tdt_scan", "k_tdt_stats", "k_stats_scan", "k_stats_hwe", "k_batch")


def main():
    workload, bench_json, out = sys.argv[1:4]
    cfg = json.loads(open(bench_json).read().strip().splitlines()[-1])["config"]
    V, N, pitch = cfg.get("variants_per_tile", cfg["variants_per_gpu"]), cfg["samples"], cfg["row_pitch_bytes"]
    acc = {}
    for path in sys.argv[4:]:
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            short = name.split("(")[0].split("<")[0].replace("void ", "").replace("hpgv::", "")
            for k in KERNELS:
                if short.startswith(k):
                    a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    kernels = {}
    for k, counters in acc.items():
        d = {c: v[0] / v[1] for c, v in counters.items()}
        d["launches_seen"] = max(v[1] for v in counters.values())
        d["variants_per_launch"] = V
        if "SQ_INSTS_VALU" in d:
            d["valu_insts_per_variant"] = d["SQ_INSTS_VALU"] / V
        if "SQ_INSTS_SALU" in d:
            d["salu_insts_per_variant"] = d["SQ_INSTS_SALU"] / V
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in d:
                    d[c + "_frac_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
            d["hbm_read_bytes_doubled"] = d["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024
        kernels[k] = d
    doc = {"workload": workload, "variants": V, "samples": N, "row_pitch_bytes": pitch,
           "method": "rocprofv3 --kernel-trace --pmc <counters>, separate passes of `python3 bench.py --workload %s --steps 3 --warmup 1 "
                     "--no-cpu-baseline`; per-launch averages" % workload,
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: {c: round(v, 3) for c, v in d.items() if "per_variant" in c or "frac" in c} for k, d in kernels.items()}))


if __name__ == "__main__":
    main()
