"""CPU suite: the host mirror's own logic (containers, thread-safe result list,
GT-text staging) built with AddressSanitizer + UBSan and run without a GPU."""
import os
import subprocess

import numpy as np
import pytest

from helpers import QUIRK_GTS, hpgv
from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_sources():
    hdir = os.path.join(ROOT, "hpg-variant_amd", "host")
    return [os.path.join(hdir, f) for f in sorted(os.listdir(hdir)) if f.endswith(".c")]


def _compile(tmp_path_factory, name, flags):
    hpgv.build()
    from importlib import import_module
    b = import_module("hpg-variant_amd._build")
    out = str(tmp_path_factory.mktemp("hostcpu") / name)
    # the adapters' source is compiled in directly so that it is instrumented too
    subprocess.check_call(["gcc", "-g", "-std=gnu99", "-fopenmp"] + flags +
                          ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "hpg-variant_amd", "host"),
                           os.path.join(ROOT, "tests", "c", "host_cpu_check.c")] + _host_sources() +
                          ["-o", out, "-L", b.LIBDIR, "-lhpgv", "-Wl,-rpath," + b.LIBDIR, "-lm", "-lz", "-lpthread"])
    return out


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return _compile(tmp_path_factory, "host_cpu_check", ["-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"])


@pytest.fixture(scope="module")
def exe_plain(tmp_path_factory):
    """the shipped optimisation level, no sanitizer: the word-at-a-time staging path (which the ASan build never takes)"""
    return _compile(tmp_path_factory, "host_cpu_check_plain", ["-O2"])


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run([exe] + list(args), capture_output=True, text=True, env=env)


def test_containers_and_result_list_under_sanitizers(exe):
    r = _run(exe, "containers")
    assert r.returncode == 0 and "CONTAINERS OK" in r.stdout and "FAIL" not in r.stdout, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


@pytest.mark.parametrize("build", ["asan", "plain"])
def test_staging_matches_the_oracle_encoder(exe, exe_plain, tmp_path, build):
    exe = exe if build == "asan" else exe_plain
    rng = np.random.default_rng(4)
    pool = QUIRK_GTS + ["0/1:12:99", ".:3", "1/", "/1", "0|0:.", "10/2", "a/1", "-1/0", "+1/1", "0/1/2", "2", "<empty>",
                        "./.:0,0", "0/0:1,2:3", "7:1/0", "7:./1:5", "7", "7:"]
    pool += ["1/.", "./1", "1|.", ".|.", "9/9", "0/1:", "0/:", "1/1x", "0/1\t", "./.:", ".", "0", "1:3"]
    hot = ["0/0", "0/1", "1|0", "1/1", "./.", "./1", "1/.", "9/9", "0/1:12:99", "./.:0,0", ".|.", "2/0", "0|9:."]
    long_fmt = ":".join("K%d" % k for k in range(120)) + ":GT"            # > 255 bytes in front of GT (assoc.c:45 strndup's it all)
    fmts = ["GT", "GT:DP", "DP:GT", "DP:GQ:GT", "DP:GQ", "GTX:GT", "XGT", long_fmt]
    chroms = ["1", "X", "XY", "x", "23"]
    n, v = 37, 60
    rows = []
    with open(tmp_path / "in.txt", "w") as f:
        f.write("%d %d\n" % (n, v))
        for i in range(v):
            fmt, chrom = fmts[i % len(fmts)], chroms[i % len(chroms)]
            # two rows in three hold only strings of the hot shape ("a/b", one-character alleles): those rows take the
            # word-at-a-time path of the plain build; a row with any other string is staged the general way
            src = pool if i % 3 == 0 else hot
            ss = [src[int(k)] for k in rng.integers(0, len(src), size=n)]
            f.write("%s %s %s\n" % (chrom, fmt, " ".join(ss)))
            rows.append((chrom, fmt, ss))
    r = _run(exe, "stage", str(tmp_path / "in.txt"))
    assert r.returncode == 0, r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(lines) == 2 * v
    for k, t in enumerate(lines):
        strict, i = int(t[0]) == 1, k % v
        chrom, fmt, ss = rows[i]
        assert int(t[1]) == orc.lib().orc_chrom_is_x(chrom.encode(), len(chrom))
        keys = fmt.split(":")
        if "GT" not in keys:
            assert all(x == "ff" for x in t[2:])                     # no GT in FORMAT: nothing usable
            continue
        pos = keys.index("GT")
        exp = ["%02x" % orc.encode_sample("" if s == "<empty>" else s, pos, strict) for s in ss]
        assert t[2:] == exp, (i, strict, list(zip(ss, t[2:], exp)))


@pytest.mark.parametrize("build", ["asan", "plain"])
def test_staging_never_reads_over_a_page_edge(exe, exe_plain, build):
    r = _run(exe if build == "asan" else exe_plain, "pageedge")
    assert r.returncode == 0 and "PAGEEDGE OK" in r.stdout, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_in_process_sort_equals_gnu_sort(exe, tmp_path):
    # assoc_runner.c:255-258 shells out to `sort -k1,1h -k2,2n`; the in-process replacement must give
    # the same file (C locale), header line included
    rng = np.random.default_rng(12)
    chroms = [str(c) for c in range(1, 23)] + ["X", "Y", "MT", "chr1", "1_KI270706v1_random", "GL000192.1", "10", "010", "2K"]
    lines = ["#CHR\tPOS\tID\tA1\tA2\tT\tU\tOR\tCHISQ\tP-VALUE"]
    for i in range(3000):
        c = chroms[int(rng.integers(0, len(chroms)))]
        pos = int(rng.integers(1, 5000)) if i % 10 else int(rng.integers(1, 250_000_000))
        lines.append("%s\t%d\trs%d\tA\tC\t%d\t%d\t%6f\t%6f\t%6f" % (c, pos, int(rng.integers(0, 50)), i % 7, i % 5, 1.5, 2.25, 0.125))
    order = rng.permutation(len(lines))
    body = "\n".join(lines[i] for i in order) + "\n"
    a, b = tmp_path / "mine.tsv", tmp_path / "gnu_in.tsv"
    a.write_text(body); b.write_text(body)
    r = _run(exe, "sort", str(a))
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, r.stderr
    env = dict(os.environ, LC_ALL="C")
    gnu = subprocess.run(["sort", "-k1,1h", "-k2,2n", str(b)], capture_output=True, text=True, env=env, check=True).stdout
    assert a.read_text() == gnu
    assert a.read_text().splitlines()[0].startswith("#CHR") or "X" in chroms      # header sorts with the text keys


def _bgzf(data, block=0xff00):
    """BGZF as bgzip writes it: gzip members with a 'BC' extra field holding the block size, then the
    28-byte end-of-file marker."""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]
    for ch in chunks:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(ch) + co.flush()
        bsize = 12 + 6 + len(comp) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6)
        out += b"BC" + struct.pack("<HH", 2, bsize - 1) + comp + struct.pack("<II", zlib.crc32(ch), len(ch))
    return bytes(out)


@pytest.mark.parametrize("kind", ["plain", "gzip", "bgzf"])
@pytest.mark.parametrize("header", [False, True])
@pytest.mark.parametrize("batch", [1 << 16, 200_000, 1 << 22])
def test_line_reader_sources(exe, tmp_path, kind, batch, header):
    # --compression gzip|bgzip (shared_options.c:60-61): every source hands out the same bytes in whole lines
    import gzip
    rng = np.random.default_rng(7)
    lines = []
    for i in range(9000):
        n = int(rng.integers(0, 300)) if i % 50 else int(rng.integers(20_000, 60_000))
        lines.append(bytes(rng.integers(48, 58, size=n, dtype=np.uint8)) + b"\n")
    data = b"".join(lines)[:-1]                                      # the last line has no newline
    head = b""
    if header:                                                       # the runners take the header off first; what
        head = b"##fileformat=VCFv4.1\n" + b"##x=" + b"y" * 100_000 + b"\n"          # was read past it is carried over
        head += b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"s%d" % i for i in range(3000)) + b"\n"
    src = tmp_path / "in"
    full = head + data
    src.write_bytes({"plain": full, "gzip": gzip.compress(full, 1), "bgzf": _bgzf(full)}[kind])
    r = _run(exe, "copy", str(src), str(tmp_path / "out"), str(batch), *(["vcf"] if header else []))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    assert (tmp_path / "out").read_bytes() == data
    assert int(r.stdout.split()[-1]) >= len(data) // batch


def test_line_reader_rejects_damaged_bgzf(exe, tmp_path):
    data = b"".join(b"%d\n" % i for i in range(200_000))
    blob = bytearray(_bgzf(data))
    blob[len(blob) // 2] ^= 0x55
    (tmp_path / "in").write_bytes(bytes(blob))
    r = _run(exe, "copy", str(tmp_path / "in"), str(tmp_path / "out"), str(1 << 20))
    assert r.returncode != 0
    assert "AddressSanitizer" not in r.stderr, r.stderr


@pytest.mark.parametrize("kind", ["plain", "bgzf"])
def test_line_reader_header_longer_than_its_buffer(exe, tmp_path, kind):
    head = b"##big=" + b"z" * (5 << 20) + b"\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\n"
    data = b"".join(b"1\t%d\trs\tA\tC\t.\t.\t.\tGT\t0/1\t1/1\n" % i for i in range(50_000))
    (tmp_path / "in").write_bytes(head + data if kind == "plain" else _bgzf(head + data))
    r = _run(exe, "copy", str(tmp_path / "in"), str(tmp_path / "out"), str(1 << 18), "vcf")
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, r.stdout + r.stderr
    assert (tmp_path / "out").read_bytes() == data
    (tmp_path / "nohdr").write_bytes(data)
    assert _run(exe, "copy", str(tmp_path / "nohdr"), str(tmp_path / "out2"), str(1 << 18), "vcf").returncode != 0


def test_deflate_decoder_against_zlib_under_sanitizers(exe):
    # hpgv_host_inflate_raw (the bgzip reader's decoder): 750 valid streams (text, incompressible, runs, every zlib strategy
    # incl. fixed codes and stored blocks, sizes 0 .. 1 MiB) decode to the original; damaged / truncated streams and wrong
    # output sizes are refused or decoded without any access outside the buffers
    r = _run(exe, "inflate")
    assert r.returncode == 0 and "inflate ok" in r.stdout, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
