#!/usr/bin/env python3
"""End-to-end rate of the file-level runner: synthetic VCF + PED files in, sorted
.chisq TSV out (file in the page cache -> PCIe -> GPU tokenizer -> scan -> writer ->
in-process sort), with the stage times the runner reports.  Diagnostic tool.

  python tools/bench_file_runner.py [n_samples] [n_variants] [plain,bgzf,gzip] [batch_MB,...] [ENV=v,ENV=v|ENV=v ...]

The fifth argument repeats the timed run of every input under other settings of the runner's environment switches (one
more run per '|'-separated group), e.g. "HPGV_IO_THREADS=8|HPGV_BGZF_HOST_TABLE=1".
"""
import ctypes as C
import importlib
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hpgv = importlib.import_module("hpg-variant_amd")
b = importlib.import_module("hpg-variant_amd._build")

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
n_variants = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
kinds = (sys.argv[3] if len(sys.argv) > 3 else "plain").split(",")
batches = [int(x) << 20 for x in (sys.argv[4] if len(sys.argv) > 4 else "64,256").split(",")]
variants_env = [dict(kv.split("=", 1) for kv in grp.split(",") if kv) for grp in (sys.argv[5].split("|") if len(sys.argv) > 5 else [])]
rng = np.random.default_rng(0)
codes = np.array(["0/0", "0/1", "1/1", "./."])
d = tempfile.mkdtemp()
vcf, ped, out = os.path.join(d, "in.vcf"), os.path.join(d, "in.ped"), os.path.join(d, "out.chisq")
bodies = ["\t".join(codes[rng.choice(4, size=n_samples, p=[0.5, 0.3, 0.19, 0.01])]) for _ in range(32)]
with open(vcf, "w", buffering=8 << 20) as f:      # large writes: the page cache then holds large folios, as after a copy or a read from disk
    f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
            "\t".join("S%d" % j for j in range(n_samples)) + "\n")
    for i in range(n_variants):
        # position-sorted like a real VCF (chromosomes in blocks), so the result file needs no re-ordering
        f.write("%d\t%d\trs%d\tA\tG\t.\tPASS\t.\tGT\t%s\n" % (1 + i * 22 // n_variants, 1000 + i, i, bodies[i % 32]))
with open(ped, "w") as f:
    for j in range(n_samples):
        f.write("F%d S%d 0 0 %d %d\n" % (j, j, 1 + j % 2, 1 + j % 2))
size = os.path.getsize(vcf)


bgzf_level = int(os.environ.get("BENCH_BGZF_LEVEL", "6"))      # bgzip's default is zlib's (6); round 1 measured files of level 1
BGZF_BLOCK = 0xff00


def bgzf_blocks(args):
    src, off, n = args
    with open(src, "rb") as fi:
        fi.seek(off)
        data = fi.read(n)
    parts = []
    for k in range(0, len(data), BGZF_BLOCK):
        ch = data[k:k + BGZF_BLOCK]
        co = zlib.compressobj(bgzf_level, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        parts.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
                     + comp + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(parts)


def bgzf_file(src, dst):
    """what `bgzip -l level` writes: blocks of 0xff00 bytes of text and the empty block at the end; compressed by a pool of
    processes (level 6 of 32 GB would take minutes on one core)"""
    import multiprocessing as mp
    piece = BGZF_BLOCK * 256
    tasks = [(src, off, piece) for off in range(0, os.path.getsize(src), piece)]
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool, open(dst, "wb", buffering=8 << 20) as fo:
        for out_bytes in pool.imap(bgzf_blocks, tasks, chunksize=1):
            fo.write(out_bytes)
        fo.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")


# every input file before the library is loaded: the compressing processes are forked from one that has not touched the device
paths = {"plain": vcf}
if "bgzf" in kinds:
    paths["bgzf"] = vcf + ".bgz"
    bgzf_file(vcf, paths["bgzf"])
if "gzip" in kinds:
    paths["gzip"] = vcf + ".gz"
    os.system("gzip -1 -c %s > %s" % (vcf, paths["gzip"]))
L = C.CDLL(b.HOSTLIB)
L.hpgv_run_assoc.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_size_t, C.POINTER(C.c_long)]
L.hpgv_host_last_error.restype = C.c_char_p
L.hpgv_host_last_run_times.argtypes = [C.POINTER(C.c_double)]
import hashlib
res = []
digests = set()
for kind in kinds:
    path = paths[kind]
    for batch in batches:
        n = C.c_long(0)
        # warm: engine, buffers, and the page cache -- the SECOND read of a freshly written file is a slow one (its pages move
        # to the kernel's active list, under one lock for all reader threads), so two runs before the timed ones
        for _ in range(2):
            L.hpgv_run_assoc(path.encode(), ped.encode(), out.encode(), 1, batch, C.byref(n))
        all_dt = []
        for _ in range(3):
            if os.path.exists(out) and not os.environ.get("BENCH_KEEP_RESULT"):
                os.remove(out)                                          # a run writes a new file (truncating the last run's 160 MB of page cache costs 15 - 35 ms)
            time.sleep(float(os.environ.get("BENCH_PAUSE_S", "0")))      # idle time between the runs (diagnosis: the first H2D copies of a run that follows another at once are slower)
            t0 = time.perf_counter()
            rc = L.hpgv_run_assoc(path.encode(), ped.encode(), out.encode(), 1, batch, C.byref(n))
            all_dt.append(time.perf_counter() - t0)
            assert rc == 0 and n.value == n_variants, L.hpgv_host_last_error()
        dt = sorted(all_dt)[1]                                         # the median of three
        digests.add(hashlib.md5(open(out, "rb").read()).hexdigest())      # every input form must give the same result file
        tm = (C.c_double * 6)()
        L.hpgv_host_last_run_times(tm)
        res.append({"input": kind, **({"bgzf_level": bgzf_level} if kind == "bgzf" else {}), "file_GB": round(os.path.getsize(path) / 1e9, 3), "batch_MB": batch >> 20, "seconds": round(dt, 3),
                    "seconds_of_3_runs": [round(x, 3) for x in all_dt],
                    "variants_per_s": round(n_variants / dt), "vcf_text_GBps": round(size / dt / 1e9, 2),
                    "stages_s_last_run": {k: round(v, 3) for k, v in zip(("read", "engine", "write", "sort", "total"), tm)}, "batches": int(tm[5])})
        for env in variants_env:
            os.environ.update(env)
            if os.path.exists(out) and not os.environ.get("BENCH_KEEP_RESULT"):
                os.remove(out)
            try:
                t0 = time.perf_counter()
                rc = L.hpgv_run_assoc(path.encode(), ped.encode(), out.encode(), 1, batch, C.byref(n))
                dt = time.perf_counter() - t0
            finally:
                for k in env:
                    del os.environ[k]
            assert rc == 0 and n.value == n_variants, L.hpgv_host_last_error()
            digests.add(hashlib.md5(open(out, "rb").read()).hexdigest())
            L.hpgv_host_last_run_times(tm)
            res.append({"input": kind, "env": env, "batch_MB": batch >> 20, "seconds": round(dt, 3), "variants_per_s": round(n_variants / dt),
                        "vcf_text_GBps": round(size / dt / 1e9, 2),
                        "stages_s": {k: round(v, 3) for k, v in zip(("read", "engine", "write", "sort", "total"), tm)}})
    if path != vcf:
        os.remove(path)
print(json.dumps({"n_samples": n_samples, "n_variants": n_variants, "vcf_GB": round(size / 1e9, 2),
                  "io_threads": os.environ.get("HPGV_IO_THREADS", "default"), "result_files_identical": len(digests) == 1, "runs": res}))
for p in (vcf, ped, out):
    os.remove(p)
os.rmdir(d)
