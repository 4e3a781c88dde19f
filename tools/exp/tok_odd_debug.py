#!/usr/bin/env python3
"""Which three-byte fields the GPU tokenizers and the oracle disagree on (diagnostic)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hpgv = importlib.import_module("hpg-variant_amd")
from oracle import pyoracle as orc
odd = bytes([0x00, 0x0D, 0x20, 0x2A, 0x2B, 0x2C, 0x2D, 0x2E, 0x2F, 0x30, 0x39, 0x3A, 0x3B, 0x3F, 0x40, 0x50, 0x52, 0x53, 0x54, 0x5C,
             0x6F, 0x7C, 0x7D, 0x7F, 0x80, 0xAE, 0xAF, 0xCA, 0xD0, 0xF9, 0xFA, 0xFB, 0xFF])
rng = np.random.default_rng(5)
n_samples = 520
lines, fields = [], []
for i in range(60):
    cols = []
    for j in range(n_samples):
        f = bytearray([b"0123456789"[int(rng.integers(0, 10))], b"/|"[int(rng.integers(0, 2))], b"0123456789"[int(rng.integers(0, 10))]])
        if rng.random() < 0.05:
            f[int(rng.integers(0, 3))] = odd[int(rng.integers(0, len(odd)))]
        cols.append(bytes(f))
    fields.append(cols)
    lines.append(b"\t".join([b"3", b"9", b"r", b"A", b"C", b"5", b"P", b"I" * (i % 37), b"GT"]) + b"\t" + b"\t".join(cols))
text = b"\n".join(lines) + b"\n"
for strict in (0, 1):
    exp = orc.tokenize(text, n_samples, strict)
    for tiles in (0, 1, 2):
        e = hpgv.Engine(0); e.set_option("tokenizer_tiles", tiles)
        got = e.tokenize(text, n_samples, strict)
        bad = np.argwhere(got["gt"] != exp["gt"])
        seen = {}
        for (l, c) in bad:
            seen.setdefault(fields[l][c], (int(got["gt"][l, c]), int(exp["gt"][l, c])))
        print("strict", strict, "tiles", tiles, "mismatches", len(bad), dict(list(seen.items())[:12]))
        e.close()
