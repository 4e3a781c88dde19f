#!/usr/bin/env python3
"""Derives profiles/<tag>_pmc_traffic_<workload>.json from two rocprofv3 counter passes.

  python tools/pmc_traffic.py <workload> <fetch_counter_collection.csv> <write_counter_collection.csv> <bench_line.json> <out.json>

The passes are SEPARATE runs of `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` around
`python3 bench.py --workload <w> --steps 3 --warmup 1 --no-cpu-baseline` (tools/pmc_traffic.sh).  FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced stream at 64 B, so it is
doubled (MI355X_MICROARCH.md, HBM section).  bench.py reads the result for roofline.traffic.
"""
import csv
import json
import sys

SCAN_KERNELS = ("k_assoc_scan", "k_assoc_chisq", "k_assoc_fisher", "k_tdt_scan", "k_tdt_stats", "k_stats_scan")


def averages(path, counter):
    acc = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        short = name.split("(")[0].split("<")[0].replace("void ", "").replace("hpgv::", "")
        for k in SCAN_KERNELS:
            if short.startswith(k):
                a = acc.setdefault(k, {"sum": 0.0, "n": 0, "name": name[:80]})
                a["sum"] += float(row["Counter_Value"])
                a["n"] += 1
    return acc


def main():
    workload, fetch_csv, write_csv, bench_json, out = sys.argv[1:6]
    cfg = json.loads(open(bench_json).read().strip().splitlines()[-1])["config"]     # the bench line of the same pass
    V, N, pitch = cfg.get("variants_per_tile", cfg["variants_per_gpu"]), cfg["samples"], cfg["row_pitch_bytes"]    # variants per LAUNCH
    fetch = averages(fetch_csv, "FETCH_SIZE")
    write = averages(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in fetch:
        f = fetch[k]["sum"] / fetch[k]["n"]
        w = write[k]["sum"] / write[k]["n"] if k in write else 0.0
        kernels[k] = {"FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w, "launches": fetch[k]["n"],
                      "kernel_name": fetch[k]["name"],
                      "hbm_read_bytes": f * 1024 * 2, "hbm_write_bytes": w * 1024,
                      "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024}
    doc = {"workload": workload, "variants": V, "samples": N, "row_pitch_bytes": pitch,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes of `python3 bench.py "
                     "--workload %s --steps 3 --warmup 1 --no-cpu-baseline`; KiB units; FETCH_SIZE doubled (gfx950 tallies "
                     "128-B requests of a wide coalesced stream at 64 B: MI355X_MICROARCH.md, HBM)" % workload,
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in kernels.items()}))


if __name__ == "__main__":
    main()
