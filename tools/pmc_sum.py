#!/usr/bin/env python3
"""Sums FETCH_SIZE / WRITE_SIZE over ALL kernels of a run (rocprofv3 --pmc counter_collection.csv files) and divides by the
number of calls the profiled tool made: the HBM traffic of one call.
  python tools/pmc_sum.py <tool_line.json> <fetch.csv> <write.csv>"""
import csv
import json
import sys

line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tot = {}
per_kernel = {}
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path)):
        c = row["Counter_Name"]
        v = float(row["Counter_Value"])
        tot[c] = tot.get(c, 0.0) + v
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("hpgv::", "")[:40]
        per_kernel.setdefault(k, {}).setdefault(c, 0.0)
        per_kernel[k][c] += v
calls = line["calls"]
fetch = tot.get("FETCH_SIZE", 0.0) * 2048 / calls          # KiB, doubled on gfx950 (wide coalesced streams)
write = tot.get("WRITE_SIZE", 0.0) * 1024 / calls
alg = line["text_bytes"] + line["matrix_bytes"]
out = dict(line, hbm_read_bytes_per_call=fetch, hbm_write_bytes_per_call=write, text_plus_matrix_bytes=alg,
           read_over_text_plus_matrix=round(fetch / alg, 3),
           per_kernel_GB_per_call={k: {c: round(v * (2048 if c == "FETCH_SIZE" else 1024) / calls / 1e9, 4) for c, v in d.items()} for k, d in per_kernel.items()})
print(json.dumps(out))
