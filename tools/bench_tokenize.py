#!/usr/bin/env python3
"""Throughput of the GPU VCF-text tokenizer on device-resident text, and of the
host entry point including PCIe.  Diagnostic tool."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
rng = np.random.default_rng(0)
codes = np.array(["0/0", "0/1", "1/1", "./."])
lines = []
for i in range(64):                                      # 64 distinct lines, repeated
    body = "\t".join(codes[rng.choice(4, size=n_samples, p=[0.5, 0.3, 0.19, 0.01])])
    lines.append("%d\t%d\trs%d\tA\tG\t.\tPASS\tAC=1;AN=2\tGT\t%s\n" % (1 + i % 22, 1000 + i, i, body))
text = ("".join(lines) * (n_lines // 64)).encode()
n_lines = (n_lines // 64) * 64
e = hpgv.Engine(0)
if os.environ.get("TOK_TILES"):                          # 1: two sweeps (default), 2: one sweep with look-back, 0: the three-sweep form
    e.set_option("tokenizer_tiles", int(os.environ["TOK_TILES"]))
dev = torch.device("cuda", 0)
d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
pitch = n_samples
d_gt = torch.empty(n_lines * pitch, dtype=torch.uint8, device=dev)
d_isx = torch.empty(n_lines, dtype=torch.uint8, device=dev)
d_n = torch.zeros(1, dtype=torch.int32, device=dev)
d_status = torch.empty(n_lines, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream()
L = e.L
def run():
    rc = L.hpgv_tokenize_dev(e.h, d_text.data_ptr(), len(text), n_samples, 1, n_lines, d_n.data_ptr(), None, None,
                             d_gt.data_ptr(), pitch, d_isx.data_ptr(), d_status.data_ptr(), st.cuda_stream)
    assert rc == 0, L.hpgv_last_error(e.h)
run(); torch.cuda.synchronize()
assert int(d_n.item()) == n_lines and int(d_status.abs().sum().item()) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(5):
    run()
b.record(st); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
# host entry point: text in host memory -> matrix in host memory (PCIe both ways)
t0 = time.perf_counter()
res = e.tokenize(text, n_samples, True, n_lines)
host_s = time.perf_counter() - t0
print(json.dumps({"n_samples": n_samples, "n_lines": n_lines, "text_GB": len(text) / 1e9,
                  "device_ms": round(ms, 3), "device_text_GBps": round(len(text) / ms / 1e6, 1),
                  "device_Mgenotypes_per_s": round(n_lines * n_samples / ms / 1e3, 1),
                  "host_entry_s": round(host_s, 3), "host_entry_text_GBps": round(len(text) / host_s / 1e9, 2)}))
e.close()
