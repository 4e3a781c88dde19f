#!/usr/bin/env python3
"""Rate of the GPU raw-DEFLATE decoder (hpgv_inflate_blocks_dev) on BGZF-sized blocks of genotype text: n_blocks payloads of
65 280 bytes of text each (64 distinct ones, repeated), compressed with zlib at the given level, all decoded in one launch.

  python tools/bench_inflate.py [n_blocks] [level] [inflate_wave: 2 = one wave per block (default here), 0 = one lane per block, 1 = the library's choice by size] [text buffer: 0 = hpgv_dev_alloc (default), 1 = a reserved range backed by hpgv_dev_commit]
"""
import importlib
import json
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")
n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wave = int(sys.argv[3]) if len(sys.argv) > 3 else 2
grows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(0)
codes = np.array(["0/0", "0/1", "1/1", "./."])
raw, comp = [], []
for _ in range(64):
    t = ("\t".join(codes[rng.choice(4, size=17000, p=[0.5, 0.3, 0.19, 0.01])]) + "\n").encode()[:65280]
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    raw.append(t); comp.append(co.compress(t) + co.flush())
pick = np.arange(n_blocks) % 64
clen = np.array([len(c) for c in comp], np.uint32)
in_len = clen[pick]; out_len = np.full(n_blocks, 65280, np.uint32)
base = np.concatenate([[0], np.cumsum(clen[:-1], dtype=np.uint64)]).astype(np.uint64)
# every block gets its own copy of the compressed bytes (as in a file), laid out back to back
in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
cat = np.frombuffer(b"".join(comp), np.uint8)
cbytes = np.empty(int(in_len.sum()) + 16, np.uint8)
for j in range(64):
    idx = np.flatnonzero(pick == j)
    src = cat[int(base[j]): int(base[j]) + int(clen[j])]
    for i in idx[:0]:
        pass
# vectorised fill: blocks of the same kind have the same length
for j in range(64):
    idx = np.flatnonzero(pick == j)
    src = cat[int(base[j]): int(base[j]) + int(clen[j])]
    pos = in_off[idx].astype(np.int64)[:, None] + np.arange(int(clen[j]))[None, :]
    cbytes[pos] = src[None, :]
out_off = (np.arange(n_blocks, dtype=np.uint64) * 65280).astype(np.uint64)
total = n_blocks * 65280
e = hpgv.Engine(0)
e.set_option("inflate_wave", wave)
d_comp = e.alloc(len(cbytes))
if grows:
    d_text = e.dev_reserve(56 << 30)
    e.dev_commit(d_text, total + 16)
else:
    d_text = e.alloc(total + 16)
d_io, d_il, d_oo, d_ol, d_st = e.alloc(8 * n_blocks), e.alloc(4 * n_blocks), e.alloc(8 * n_blocks), e.alloc(4 * n_blocks), e.alloc(4 * n_blocks)
for d, a in ((d_comp, cbytes), (d_io, in_off), (d_il, in_len), (d_oo, out_off), (d_ol, out_len)):
    e.h2d(d, a)
runs = []
for _ in range(3):
    e.sync(); t0 = time.perf_counter()
    e.inflate_blocks(d_comp, d_io, d_il, d_oo, d_ol, n_blocks, d_text, d_st)
    e.sync(); runs.append(time.perf_counter() - t0)
status = e.d2h(d_st, (n_blocks,), np.int32)
chk = e.d2h(d_text.value + (n_blocks - 1) * 65280, (65280,), np.uint8).tobytes()
ok = bool((status == 0).all()) and chk == raw[int(pick[-1])]
dt = min(runs)
print(json.dumps({"blocks": n_blocks, "zlib_level": level, "decoder": {0: "lane per block", 1: "by size", 2: "wave per block", 3: "lane per block, tables in LDS", 4: "wave per block, several symbols per round"}[wave], "text_buffer": "reserved range" if grows else "allocation", "text_GB": total / 1e9, "compressed_GB": float(in_len.sum()) / 1e9,
                  "seconds": round(dt, 4), "text_GBps": total / dt / 1e9, "compressed_GBps": float(in_len.sum()) / dt / 1e9, "ok": ok}))
if grows:
    e.dev_release(d_text)
e.close()
