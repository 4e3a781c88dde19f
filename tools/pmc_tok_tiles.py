#!/usr/bin/env python3
"""HBM traffic of ONE tokenizer call on a window of decoded text, from two rocprofv3 counter passes around
tools/bench_tokenize_tiles.py (FETCH_SIZE and WRITE_SIZE, passes of their own): the k_tok_* kernels' bytes divided by the six
calls the tool makes per mode, and the CRC kernel's bytes per launch beside them.
  python tools/pmc_tok_tiles.py <tool_line.json> <mode> <fetch.csv> <write.csv>"""
import csv
import json
import sys

line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
mode = sys.argv[2]
CALLS = 6                                                # one checked call + five timed ones
tok, crc = {}, {}
for path in sys.argv[3:]:
    for row in csv.DictReader(open(path)):
        c = row["Counter_Name"]
        if c not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        v = float(row["Counter_Value"]) * (2048 if c == "FETCH_SIZE" else 1024)      # KiB; FETCH_SIZE doubled on gfx950 (wide coalesced streams)
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("hpgv::", "")[:48]
        if k.startswith("k_tok_"):
            d = tok.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
            d[c] += v
            d["launches"] += c == "FETCH_SIZE"
        elif k.startswith("k_bgzf_crc"):
            d = crc.setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})
            d[c].append(v)
alg = (line["window_text_GB"] + line["matrix_GB"]) * 1e9
reads = sum(d["FETCH_SIZE"] for d in tok.values()) / CALLS
writes = sum(d["WRITE_SIZE"] for d in tok.values()) / CALLS
out = {"mode": mode, "n_samples": line["n_samples"], "n_lines": line["n_lines"], "window_text_bytes": line["window_text_GB"] * 1e9,
       "matrix_bytes": line["matrix_GB"] * 1e9, "device_ms_unprofiled": None,
       "hbm_read_bytes_per_call": reads, "hbm_write_bytes_per_call": writes, "text_plus_matrix_bytes": alg,
       "read_over_text_plus_matrix": round(reads / alg, 3), "read_plus_write_over_text_plus_matrix": round((reads + writes) / alg, 3),
       "tokenizer_kernels_GB_per_call": {k: {"FETCH_SIZE": round(d["FETCH_SIZE"] / CALLS / 1e9, 4), "WRITE_SIZE": round(d["WRITE_SIZE"] / CALLS / 1e9, 4),
                                             "launches_per_call": d["launches"] / CALLS} for k, d in tok.items()},
       "crc_kernel_GB_per_launch": {k: {c: [round(x / 1e9, 4) for x in v] for c, v in d.items()} for k, d in crc.items()},
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes of `python3 tools/bench_tokenize_tiles.py "
                 "N L %s`; KiB units; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of a wide coalesced stream at 64 B: "
                 "MI355X_MICROARCH.md, HBM)" % mode}
print(json.dumps(out))
