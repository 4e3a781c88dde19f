#!/usr/bin/env python3
"""The tokenizer on WINDOWS OF TEXT THE BGZIP DECODER LEFT ON THE DEVICE: with the decoder's tile records (the CRC kernel
writes them while it checks the blocks: hpgv_bgzf_verify_tiles_dev, hpgv_text_alias_tiles) against the ordinary two-sweep call
on the same window.  The text is compressed here into BGZF-sized blocks (65 280 bytes, zlib level 1), inflated and checked on
the device, then a window that starts and ends inside the text (at line starts, not at tile boundaries) is tokenized both
ways: device ms, text GB/s, and -- under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` in passes of their own -- the bytes moved.
  python tools/bench_tokenize_tiles.py [n_samples] [n_lines] [tiles|plain|both]
Diagnostic tool."""
import ctypes as C
import importlib
import json
import os
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 16_000
which = sys.argv[3] if len(sys.argv) > 3 else "both"
rng = np.random.default_rng(0)
codes = np.array(["0/0", "0/1", "1/1", "./."])
lines = []
for i in range(64):
    body = "\t".join(codes[rng.choice(4, size=n_samples, p=[0.5, 0.3, 0.19, 0.01])])
    lines.append("%d\t%d\trs%d\tA\tG\t.\tPASS\tAC=1;AN=2\tGT\t%s\n" % (1 + i % 22, 1000 + i, i, body))
n_lines = (n_lines // 64) * 64
head = "".join(lines[:3]).encode()                       # three lines in front of the window: it begins inside a tile
text = head + ("".join(lines) * (n_lines // 64)).encode() + "".join(lines[:2]).encode()
a, b = len(head), len(text) - len("".join(lines[:2]).encode())
BS = 65280
raws = [text[p: p + BS] for p in range(0, len(text), BS)]
uniq = {}
comps = []
for r in raws:                                           # (the text repeats: each distinct block is compressed once)
    c = uniq.get(r)
    if c is None:
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        c = co.compress(r) + co.flush() + zlib.crc32(r).to_bytes(4, "little") + len(r).to_bytes(4, "little")
        uniq[r] = c
    comps.append(c)
n = len(raws)
in_len = np.array([len(c) - 8 for c in comps], np.uint32); out_len = np.array([len(r) for r in raws], np.uint32)
in_off = np.concatenate([[0], np.cumsum([len(c) for c in comps[:-1]], dtype=np.uint64)]).astype(np.uint64)
out_off = np.concatenate([[0], np.cumsum(out_len[:-1], dtype=np.uint64)]).astype(np.uint64)
cbytes = np.frombuffer(b"".join(comps) + b"\0" * 16, np.uint8)
e = hpgv.Engine(0)
L = e.L
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
d_comp = torch.from_numpy(cbytes.copy()).to(dev)
d_tab = [torch.from_numpy(x.view(np.uint8).copy()).to(dev) for x in (in_off, in_len, out_off, out_len)]
d_text = torch.empty(len(text) + 64, dtype=torch.uint8, device=dev)
d_status = torch.zeros(n, dtype=torch.int32, device=dev)
tiles_bytes = int(L.hpgv_text_tiles_bytes(len(text))); n_tiles = tiles_bytes // 32 - 1
d_tiles = torch.zeros(tiles_bytes, dtype=torch.uint8, device=dev)
ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
assert L.hpgv_inflate_blocks_dev(e.h, d_comp.data_ptr(), d_tab[0].data_ptr(), d_tab[1].data_ptr(), d_tab[2].data_ptr(), d_tab[3].data_ptr(), n, d_text.data_ptr(), d_status.data_ptr(), st.cuda_stream) == 0
out = {"n_samples": n_samples, "n_lines": n_lines, "window_text_GB": (b - a) / 1e9, "matrix_GB": n_lines * n_samples / 1e9, "blocks": n,
       "window_begins_inside_a_tile_at_byte": a % 2048}
for tiles in (0, 1):                                     # the CRC check without and with the tile records
    d_tiles.zero_(); torch.cuda.synchronize()
    ea.record(st)
    assert L.hpgv_bgzf_verify_tiles_dev(e.h, d_comp.data_ptr(), d_tab[0].data_ptr(), d_tab[1].data_ptr(), d_tab[2].data_ptr(), d_tab[3].data_ptr(), n, d_text.data_ptr(),
                                        d_status.data_ptr(), d_tiles.data_ptr() if tiles else None, n_tiles if tiles else 0, st.cuda_stream) == 0
    eb.record(st); torch.cuda.synchronize()
    out["crc_check_ms" if not tiles else "crc_check_with_tile_records_ms"] = round(ea.elapsed_time(eb), 3)
assert int(d_status.abs().sum().item()) == 0
pitch = n_samples
d_gt = torch.empty(n_lines * pitch, dtype=torch.uint8, device=dev)
d_isx = torch.empty(n_lines, dtype=torch.uint8, device=dev)
d_n = torch.zeros(1, dtype=torch.int32, device=dev)
d_st2 = torch.empty(n_lines, dtype=torch.int32, device=dev)
key = C.create_string_buffer(16)
win = d_text.data_ptr() + a


def run():
    rc = L.hpgv_tokenize_dev(e.h, win, b - a, n_samples, 1, n_lines, d_n.data_ptr(), None, None, d_gt.data_ptr(), pitch, d_isx.data_ptr(), d_st2.data_ptr(), st.cuda_stream)
    assert rc == 0, L.hpgv_last_error(e.h)


ref = None
for mode in (("tiles", "plain") if which == "both" else (which,)):
    if mode == "tiles":
        assert L.hpgv_text_alias_tiles(e.h, key, C.c_void_p(win), C.c_void_p(d_text.data_ptr()), C.c_void_p(d_tiles.data_ptr()), n_tiles) == 0
    else:
        L.hpgv_text_alias(e.h, key, None)
    d_gt.zero_(); run(); torch.cuda.synchronize()
    assert int(d_n.item()) == n_lines and int(d_st2.abs().sum().item()) == 0
    if ref is None:
        ref = d_gt.clone()
    else:
        assert torch.equal(ref, d_gt), "the two ways differ"
    ea.record(st)
    for _ in range(5):
        run()
    eb.record(st); torch.cuda.synchronize()
    ms = ea.elapsed_time(eb) / 5
    out[mode] = {"device_ms": round(ms, 4), "text_GBps": round((b - a) / ms / 1e6, 1), "text_plus_matrix_GBps": round(((b - a) + n_lines * n_samples) / ms / 1e6, 1)}
print(json.dumps(out))
e.close()
