"""GPU text staging (VCF data lines -> HPGV8 on the device)
against the oracle's TAB-split + get_alleles tokenizer."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import QUIRK_GTS, hpgv, set_or_skip, shipped
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu

WEIRD = QUIRK_GTS + ["", "0/1:12:99", ".:3", "1/", "/1", "0|0:.", "10/2", "a/1", "-1/0", "+1/1", " 1/0", "0/1/2", "2",
                     "./.:0,0", "0/0:1,2:3",
                     # allele indices no writer produces: atoi() is (int) strtol() -- cut to 32 bits, stuck at LONG_MAX past 64
                     "11111111111/0", "4294967297/1", "-4294967295/0", "99999999999999999999/1", "1/-99999999999999999999", "2147483648/0"]


@pytest.fixture(scope="module", params=[2, 1, 0], ids=["one-sweep", "tile-parallel", "line-by-line"])
def eng(request):
    """The three tokenizers: one sweep with look-back (default), tile-parallel in two sweeps (both hpgv_text2_kernels.h), count / mark /
    parse per line."""
    from helpers import set_or_skip
    e = hpgv.Engine(0)
    try:
        set_or_skip(e, "tokenizer_tiles", request.param)
    except BaseException:
        e.close()
        raise
    yield e
    e.close()


def _line(rng, n_samples, fmt, chrom, info_len=8, gts=None, n_cols=None):
    k = fmt.split(":").index("GT") if "GT" in fmt.split(":") else 0
    n_cols = n_samples if n_cols is None else n_cols
    cols = []
    for j in range(n_cols):
        g = gts[j] if gts is not None else (WEIRD[int(rng.integers(0, len(WEIRD)))] if rng.random() < 0.3
                                            else ["0/0", "0/1", "1/1", "1|0"][int(rng.integers(0, 4))])
        parts = ["7"] * k + [g]
        cols.append(":".join(parts))
    info = "I=" + "x" * info_len
    return "\t".join([chrom, "12345", "rs1", "A", "C,T", "50", "PASS", info, fmt] + cols)


def _check(eng, text, n_samples, strict, max_lines=None):
    got = eng.tokenize(text, n_samples, strict, max_lines)
    exp = orc.tokenize(text, n_samples, strict, max_lines)
    assert got["n_lines"] == exp["n_lines"]
    assert np.array_equal(got["status"], exp["status"]), (got["status"], exp["status"])
    assert np.array_equal(got["is_x"], exp["is_x"])
    assert np.array_equal(got["gt"], exp["gt"]), np.argwhere(got["gt"] != exp["gt"])[:5]
    return got


@pytest.mark.parametrize("n_samples", [0, 1, 3, 64, 257, 1500])
def test_random_lines_all_shapes(eng, n_samples):
    rng = np.random.default_rng(n_samples + 1)
    lines = []
    for i in range(40):
        fmt = ["GT", "GT:DP", "DP:GT", "DP:GQ:GT:PL", "DP:GQ"][i % 5]
        chrom = ["1", "X", "22", "XY", "chrX", "Y"][i % 6]
        info_len = [3, 100, 5000, 9000][i % 4]                  # INFO longer than one 4 KiB tile
        n_cols = n_samples if i % 7 else max(0, n_samples - 2)    # some lines lack trailing samples
        if i % 11 == 0:
            n_cols = n_samples + 3                                # some have extra columns
        lines.append(_line(rng, n_samples, fmt, chrom, info_len, n_cols=n_cols))
    lines.insert(5, "")                                           # an empty line
    lines.insert(9, "1\t5\trs\tA\tC")                            # a truncated line
    for strict in (True, False):
        _check(eng, "\n".join(lines) + "\n", n_samples, strict)
        _check(eng, "\n".join(lines), n_samples, strict)          # last line without newline
        _check(eng, "\r\n".join(lines) + "\r\n", n_samples, strict)   # CRLF files


def test_both_tokenizers_give_the_same_offsets():
    """line_off / field_off (what the host cuts CHROM .. FORMAT with) from the tile-parallel tokenizer equal the
    line-by-line one's, on lines of every shape: INFO longer than a tile, empty and truncated lines, CRLF, no final newline."""
    rng = np.random.default_rng(77)
    lines = []
    for i in range(60):
        fmt = ["GT", "GT:DP", "DP:GT", "DP:GQ:GT:PL", "DP:GQ"][i % 5]
        lines.append(_line(rng, 300, fmt, ["1", "X", "", "chrX"][i % 4], [3, 100, 5000, 9000, 20000][i % 5], n_cols=[300, 300, 298, 303, 0][i % 5]))
    lines.insert(3, ""); lines.insert(4, ""); lines.insert(17, "1\t5\trs\tA\tC"); lines.insert(18, "X")
    outs = []
    for tiles in shipped("tokenizer_tiles", (2, 1, 0)):
        e = hpgv.Engine(0)
        e.set_option("tokenizer_tiles", tiles)
        res = []
        for text in ("\n".join(lines) + "\n", "\n".join(lines), "\r\n".join(lines) + "\r\n", "\n\n\n", "\t\t\n", "abc"):
            for max_lines in (None, 7):
                res.append(e.tokenize(text, 300, False, max_lines))
        outs.append(res)
        e.close()
    for forms in zip(*outs):                                          # every form this build holds gives the same (one form: the oracle tests pin it)
        a = forms[0]
        for b in forms[1:]:
            assert a["n_lines"] == b["n_lines"]
            for k in ("gt", "is_x", "status", "line_off", "field_off"):
                assert np.array_equal(a[k], b[k]), k


def test_empty_and_capacity(eng):
    assert eng.tokenize(b"", 5)["n_lines"] == 0
    rng = np.random.default_rng(3)
    text = "\n".join(_line(rng, 10, "GT", "1") for _ in range(20)) + "\n"
    got = _check(eng, text, 10, True, max_lines=7)               # capacity smaller than the text
    assert got["n_lines"] == 20 and got["gt"].shape[0] == 7
    # offsets let the host cut the fixed columns without re-scanning
    full = eng.tokenize(text, 10)
    for i in range(20):
        ls = int(full["line_off"][i]); fo = full["field_off"][i]
        assert text[ls + fo[0]: ls + fo[1] - 1] == "1" and text[ls + fo[8]: ls + fo[9] - 1] == "GT"
        assert text[ls + fo[2]: ls + fo[3] - 1] == "rs1"


def test_large_batch_feeds_the_assoc_path(eng):
    # 1500 lines x 4000 samples of plain diploid calls: text -> HPGV8 on the GPU -> assoc vs oracle on the
    # oracle-tokenized matrix (the whole "text batch in, statistics out" chain)
    rng = np.random.default_rng(8)
    n_samples, n_lines = 4000, 1500
    codes = np.array(["0/0", "0/1", "1/0", "1/1", "./."])
    idx = rng.choice(5, size=(n_lines, n_samples), p=[0.5, 0.2, 0.1, 0.15, 0.05])
    body = ["\t".join(codes[r]) for r in idx]
    text = "\n".join("%s\t%d\trs%d\tA\tG\t.\tPASS\t.\tGT\t%s" % (["1", "X"][i % 2], i, i, body[i]) for i in range(n_lines)) + "\n"
    got = _check(eng, text, n_samples, True)
    cond = (np.arange(n_samples) % 2).astype(np.uint8)
    e = hpgv.Engine(0)
    e.set_cohort(cond)
    res = e.assoc(hpgv.TASK_CHISQ, got["gt"], got["is_x"])
    A1, A2, U1, U2 = orc.assoc_counts(orc.tokenize(text, n_samples)["gt"], cond, got["is_x"])
    assert np.array_equal(res["A1"], A1) and np.array_equal(res["U2"], U2)
    e.close()


def test_text_in_statistics_out(eng):
    # hpgv_assoc_text / hpgv_tdt_text: tokenize + layout + scan + statistics in one call
    from helpers import check_assoc, oracle_assoc, assert_close, make_families
    rng = np.random.default_rng(21)
    n_samples = 900
    lines = [_line(rng, n_samples, ["GT", "GT:DP", "DP:GT"][i % 3], ["1", "X", "7"][i % 3], [5, 4500][i % 2]) for i in range(120)]
    text = "\n".join(lines) + "\n"
    tok = orc.tokenize(text, n_samples, True)
    cond = rng.choice([0, 1, 2], size=n_samples).astype(np.uint8)
    e = hpgv.Engine(0)
    e.set_cohort(cond)
    lf = orc.logfact(n_samples * 10)
    e.set_logfact(lf)
    for task, otask in ((hpgv.TASK_CHISQ, orc.TASK_CHISQ), (hpgv.TASK_FISHER, orc.TASK_FISHER)):
        res = e.assoc_text(task, text)
        assert res["n_lines"] == 120 and np.array_equal(res["status"], tok["status"])
        check_assoc(res, oracle_assoc(otask, tok["gt"], cond, tok["is_x"], lf), task)
    short = e.assoc_text(hpgv.TASK_CHISQ, text, max_lines=50)
    assert short["n_lines"] == 120 and len(short["A1"]) == 50
    fam = make_families(rng, n_samples, 200, 3)
    e.set_families(n_samples, *fam)
    res = e.tdt_text(text)
    t1, t2 = orc.tdt_counts(tok["gt"], *fam, chrom_is_x=tok["is_x"])
    assert np.array_equal(res["t1"], t1) and np.array_equal(res["t2"], t2)
    odds, chisq, p = orc.tdt_stats(t1, t2)
    assert_close(res["p"], p, "tdt p"); assert_close(res["chisq"], chisq, "tdt chisq")
    assert e.assoc_text(hpgv.TASK_CHISQ, b"")["n_lines"] == 0
    e.close()


def test_text_entry_point_from_concurrent_threads(eng):
    # the file runners keep two text batches in flight (two engine threads): the tokenizer's device
    # scratch must not be shared between concurrent calls
    import threading
    from helpers import check_assoc, oracle_assoc
    rng = np.random.default_rng(33)
    n_samples = 700
    cond = rng.choice([0, 1, 2], size=n_samples).astype(np.uint8)
    e = hpgv.Engine(0)
    e.set_cohort(cond)
    texts, expected = [], []
    for t in range(4):
        lines = [_line(rng, n_samples, "GT", "1", 5) for _ in range(300 + 117 * t)]
        text = "\n".join(lines) + "\n"
        tok = orc.tokenize(text, n_samples, True)
        texts.append(text)
        expected.append(oracle_assoc(orc.TASK_CHISQ, tok["gt"], cond, tok["is_x"], None))
    errors = []

    def work(t):
        try:
            for _ in range(25):
                res = e.assoc_text(hpgv.TASK_CHISQ, texts[t])
                assert res["n_lines"] == 300 + 117 * t
                check_assoc(res, expected[t], hpgv.TASK_CHISQ)
        except Exception as ex:                                     # noqa: BLE001
            errors.append((t, repr(ex)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    e.close()
    assert not errors, errors


@pytest.mark.parametrize("wave", [2, 4, 0, 3], ids=["wave_per_block", "wave_per_block_several_symbols", "lane_per_block", "lane_per_block_lds_tables"])
def test_inflate_blocks_on_the_gpu_against_zlib(wave):
    # hpgv_inflate_blocks_dev: raw-DEFLATE payloads (as in BGZF blocks) of genotype text, incompressible bytes (stored
    # blocks), runs, every zlib strategy incl. fixed codes, sizes 0 .. 65 280 -- one wave per block (the default) and one
    # lane per block; a damaged block gets a non-zero status and leaves the others alone
    import zlib
    rng = np.random.default_rng(11)
    codes = np.array(["0/0", "0/1", "1/1", "./.", "0|1"])
    blobs = []
    for n in (0, 1, 17, 300, 4096, 65280, 65280, 30000):
        txt = ("\t".join(codes[rng.choice(5, size=n // 4 + 1, p=[0.5, 0.3, 0.15, 0.01, 0.04])]) + "\n").encode()[:n]
        blobs += [txt, bytes(rng.integers(0, 256, n, dtype=np.uint8)), b"\0" * n, (b"ACGT\t.\tPASS\n" * (n // 12 + 1))[:n]]
    # matches from further back than the wave decoder's 4 KiB ring, up to the longest (258 bytes): a 20 KB random chunk
    # three times over; several DEFLATE blocks in one stream, with the empty stored block a sync flush leaves between them
    chunk = bytes(rng.integers(0, 256, 20000, dtype=np.uint8))
    blobs += [chunk * 3 + chunk[:5280], (chunk[:9000] + b"\t0/1" * 2000) * 2]
    comp, raw = [], []
    for blob in blobs[:3]:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        parts = [co.compress(blob[:len(blob) // 3]), co.flush(zlib.Z_SYNC_FLUSH), co.compress(blob[len(blob) // 3:len(blob) // 2]),
                 co.flush(zlib.Z_FULL_FLUSH), co.compress(blob[len(blob) // 2:]), co.flush()]
        comp.append(b"".join(parts)); raw.append(blob)
    for k, blob in enumerate(blobs):
        for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_FILTERED), (6, zlib.Z_HUFFMAN_ONLY),
                                (6, zlib.Z_RLE), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            comp.append(co.compress(blob) + co.flush()); raw.append(blob)
    damaged = len(comp) // 2
    good = comp[damaged]
    comp[damaged] = bytes([good[0] | 0x06]) + good[1:]                     # block type 3: invalid
    n = len(comp)
    in_len = np.array([len(c) for c in comp], np.uint32); out_len = np.array([len(r) for r in raw], np.uint32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(out_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    cbytes = np.frombuffer(b"".join(comp) + b"\0" * 16, np.uint8)
    total = int(out_len.sum())
    e = hpgv.Engine(0)
    set_or_skip(e, "inflate_wave", wave)
    d_comp, d_text = e.alloc(len(cbytes)), e.alloc(total + 16)
    d_io, d_il, d_oo, d_ol, d_st = e.alloc(8 * n), e.alloc(4 * n), e.alloc(8 * n), e.alloc(4 * n), e.alloc(4 * n)
    for d, a in ((d_comp, cbytes), (d_io, in_off), (d_il, in_len), (d_oo, out_off), (d_ol, out_len)):
        e.h2d(d, a)
    e.h2d(d_text, np.full(total + 16, 0x5A, np.uint8))
    e.inflate_blocks(d_comp, d_io, d_il, d_oo, d_ol, n, d_text, d_st)
    e.sync()
    status = e.d2h(d_st, (n,), np.int32)
    text = e.d2h(d_text, (total,), np.uint8).tobytes()
    assert status[damaged] != 0
    for k in range(n):
        if k == damaged:
            continue
        assert status[k] == 0, (k, int(status[k]), len(raw[k]))
        assert text[int(out_off[k]): int(out_off[k]) + len(raw[k])] == raw[k], k
    e.close()


def _bgzf_bytes(payloads, raws):
    """BGZF blocks as bgzip writes them (18-byte header with the BC field, CRC32, ISIZE) around raw-DEFLATE payloads."""
    import struct
    import zlib
    out, rows, pos = [], [], 0
    for c, r in zip(payloads, raws):
        bsize = 18 + len(c) + 8
        out.append(struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize - 1) + c + struct.pack("<II", zlib.crc32(r), len(r)))
        rows.append((pos + 18, len(c), len(r)))
        pos += bsize
    return b"".join(out), rows


def test_bgzf_block_table_from_the_device_copy():
    # hpgv_bgzf_scan_dev: the chain of bgzip blocks found in the compressed bytes on the device, piece by piece as a file
    # arrives, against the rows a host walk gives; a header's 16 fixed bytes planted inside a block's data must not join
    import zlib
    rng = np.random.default_rng(5)
    codes = np.array(["0/0", "0/1", "1/1", "./."])
    raws, comps = [], []
    for k in range(700):
        n = int(rng.choice([0, 1, 40, 700, 5000, 65280, 65280, 30000]))
        txt = ("\t".join(codes[rng.choice(4, size=n // 4 + 1, p=[0.6, 0.25, 0.14, 0.01])]) + "\n").encode()[:n]
        if k % 97 == 5:                                                    # incompressible, and a whole fake block inside it
            fake = bytes([31, 139, 8, 4, 1, 2, 3, 4, 0, 255, 6, 0, 66, 67, 2, 0, 40, 0]) + bytes(19) + bytes([5, 0, 0, 0])
            txt = (bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) + fake + bytes(rng.integers(0, 256, 3000, dtype=np.uint8)))
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
        else:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
        raws.append(txt); comps.append(co.compress(txt) + co.flush())
    data, rows = _bgzf_bytes(comps, raws)
    size = len(data)
    e = hpgv.Engine(0)
    d_comp = e.alloc(size + 16)
    e.h2d(d_comp, np.frombuffer(data + b"\0" * 16, np.uint8))
    cap = 256
    d_io, d_il, d_oo, d_ol = e.alloc(8 * cap), e.alloc(4 * cap), e.alloc(8 * cap), e.alloc(4 * cap)
    got, lo, text, hi, calls, fakes = [], 0, 0, 0, 0, 0
    step = size // 7 + 1
    while lo < size:
        hi = min(size, hi + step)                                          # the file arrives in seven pieces
        while True:
            n, lo2, text2, hits = e.bgzf_scan(d_comp, lo, hi, text, cap, d_io, d_il, d_oo, d_ol)
            calls += 1
            io = e.d2h(d_io, (cap,), np.uint64)[:n]; il = e.d2h(d_il, (cap,), np.uint32)[:n]
            oo = e.d2h(d_oo, (cap,), np.uint64)[:n]; ol = e.d2h(d_ol, (cap,), np.uint32)[:n]
            for a, b, c, d in zip(io, il, oo, ol):
                got.append((int(a), int(b), int(d)))
                assert int(c) == text
                text += int(d)
            assert text == text2 and lo2 >= lo and (n > 0) == (lo2 > lo)
            fakes += 1 if 0 < n < min(cap, hits) else 0                     # the chain stopped before the range's hits ran out
            lo = lo2
            if n == 0:
                break
        assert calls < 200
    assert got == rows, (len(got), len(rows))
    assert fakes >= 7                                                       # every planted header was seen, and left out
    # a file whose first block carries another extra field: no chain, the caller walks it itself
    odd = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 12, 0]) + b"XY\x02\x00zz" + data[12:]
    e.h2d(d_comp, np.frombuffer(odd[:size] + b"\0" * 16, np.uint8))
    assert e.bgzf_scan(d_comp, 0, size, 0, cap, d_io, d_il, d_oo, d_ol)[0] == 0
    e.close()


def test_device_memory_that_grows_in_place():
    # hpgv_dev_reserve / hpgv_dev_commit / hpgv_dev_release: a range backed piece by piece behaves like one allocation for
    # copies that cross the pieces, and what is written stays where it is while the range grows
    e = hpgv.Engine(0)
    p = e.dev_reserve(3 << 30)
    a = np.arange(1 << 20, dtype=np.uint32)
    done = []
    for mb in (100, 200, 300, 456, 1024, 1100, 2048):                       # pieces of 128, 128, 64, 192, 576, 128 and 896 MB
        e.dev_commit(p, mb << 20)
        at = (mb << 20) - a.nbytes - (1 << 20)                              # near the end of what is there now
        e.h2d(p.value + at, a + mb)
        done.append((at, mb))
        for at0, mb0 in done:                                               # everything written so far, across all piece ends
            assert np.array_equal(e.d2h(p.value + at0, a.shape, np.uint32), a + mb0)
    e.h2d(p.value + (128 << 20) - (2 << 20), a)                             # across the first two pieces
    assert np.array_equal(e.d2h(p.value + (128 << 20) - (2 << 20), a.shape, np.uint32), a)
    with pytest.raises(Exception):
        e.dev_commit(p, 4 << 30)                                            # more than was reserved
    e.dev_release(p)
    e.close()


@pytest.mark.parametrize("wave", [2, 4, 0, 3], ids=["wave_per_block", "wave_per_block_several_symbols", "lane_per_block", "lane_per_block_lds_tables"])
def test_inflate_random_streams_against_zlib(wave):
    # 1 500 streams over the parameters zlib offers: level, strategy, window size (256 bytes .. 32 KiB), memLevel (1: a
    # new dynamic block every 128 symbols, so a stream holds hundreds of code tables), on data of alphabets from 2 to 256
    # symbols, periodic data and text, sizes 0 .. 65 280; every stream must come back bit for bit
    import zlib
    rng = np.random.default_rng(2024)
    raws, comps = [], []
    for k in range(1500):
        n = int(rng.choice([0, 1, 2, 3, 31, 257, 258, 259, 1000, 4095, 4096, 4097, 20000, 65280])) if k % 3 == 0 else int(rng.integers(0, 65281))
        kind = k % 6
        if kind == 0:
            raw = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 1:
            raw = bytes(rng.integers(0, int(rng.choice([2, 3, 4, 16])), n, dtype=np.uint8) + 48)
        elif kind == 2:
            period = int(rng.integers(1, 300))
            raw = (bytes(rng.integers(0, 256, period, dtype=np.uint8)) * (n // period + 1))[:n]
        elif kind == 3:
            gts = np.array(["0/0", "0/1", "1/1", "./.", "0|1", "1|0"])
            raw = ("\t".join(gts[rng.choice(6, size=n // 4 + 1, p=[0.55, 0.2, 0.1, 0.02, 0.08, 0.05])]) + "\n").encode()[:n]
        elif kind == 4:
            far = bytes(rng.integers(0, 256, int(rng.integers(300, 9000)), dtype=np.uint8))
            raw = (far + bytes(rng.integers(0, 4, int(rng.integers(0, 20000)), dtype=np.uint8)) + far * 2)[:n]
        else:
            raw = (b"chr1\t%d\trs%d\tA\tG\t.\tPASS\tAC=%d;AF=0.%d\tGT:DP\t" % (k, k, k % 7, k % 97) * (n // 40 + 1))[:n]
        co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -int(rng.integers(9, 16)), int(rng.integers(1, 10)),
                              int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
        raws.append(raw); comps.append(co.compress(raw) + co.flush())
    n = len(comps)
    in_len = np.array([len(c) for c in comps], np.uint32); out_len = np.array([len(r) for r in raws], np.uint32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(out_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    cbytes = np.frombuffer(b"".join(comps) + b"\0" * 16, np.uint8)
    total = int(out_len.sum())
    e = hpgv.Engine(0)
    set_or_skip(e, "inflate_wave", wave)
    d_comp, d_text = e.alloc(len(cbytes)), e.alloc(total + 16)
    d_io, d_il, d_oo, d_ol, d_st = e.alloc(8 * n), e.alloc(4 * n), e.alloc(8 * n), e.alloc(4 * n), e.alloc(4 * n)
    for d, a in ((d_comp, cbytes), (d_io, in_off), (d_il, in_len), (d_oo, out_off), (d_ol, out_len)):
        e.h2d(d, a)
    e.h2d(d_text, np.full(total + 16, 0xA5, np.uint8))
    e.inflate_blocks(d_comp, d_io, d_il, d_oo, d_ol, n, d_text, d_st)
    e.sync()
    status = e.d2h(d_st, (n,), np.int32)
    text = e.d2h(d_text, (total + 16,), np.uint8).tobytes()
    assert not status.any(), (np.flatnonzero(status)[:10], status[status != 0][:10])
    assert text[:total] == b"".join(raws)
    assert text[total:] == b"\xa5" * 16                                  # nothing written past the last block's text
    e.close()


@pytest.mark.parametrize("wave", [4, 2], ids=["wave_per_block_several_symbols", "wave_per_block"])
def test_inflate_damaged_streams_end_with_a_status_or_a_text_of_the_right_size(wave):
    # streams with one to three bytes overwritten: the decoder may refuse a block (non-zero status: the host decodes it and
    # reports the damage) or decode what the changed bits say, but it writes nothing outside the block's own text, leaves the
    # undamaged neighbours alone and always comes back
    import zlib
    rng = np.random.default_rng(99)
    gts = np.array(["0/0", "0/1", "1/1", "./.", "0|1", "1|0"])
    raws, comps, hurt = [], [], []
    for k in range(400):
        n = int(rng.integers(200, 65281))
        if k % 3 == 0:
            raw = bytes(rng.integers(0, int(rng.choice([2, 4, 64, 256])), n, dtype=np.uint8))
        else:
            raw = ("\t".join(gts[rng.choice(6, size=n // 4 + 1, p=[0.55, 0.2, 0.1, 0.02, 0.08, 0.05])]) + "\n").encode()[:n]
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15, int(rng.integers(1, 10)),
                              int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_FIXED])))
        comp = bytearray(co.compress(raw) + co.flush())
        damaged = k % 4 != 3
        if damaged:
            for _ in range(int(rng.integers(1, 4))):
                comp[int(rng.integers(0, len(comp)))] = int(rng.integers(0, 256))
        raws.append(raw); comps.append(bytes(comp)); hurt.append(damaged)
    n = len(comps)
    in_len = np.array([len(c) for c in comps], np.uint32); out_len = np.array([len(r) for r in raws], np.uint32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    gap = 64                                                            # guard bytes between the blocks' texts
    out_off = (np.concatenate([[0], np.cumsum(out_len[:-1].astype(np.uint64) + gap)])).astype(np.uint64)
    cbytes = np.frombuffer(b"".join(comps) + b"\0" * 16, np.uint8)
    total = int(out_off[-1]) + int(out_len[-1]) + gap
    e = hpgv.Engine(0)
    set_or_skip(e, "inflate_wave", wave)
    d_comp, d_text = e.alloc(len(cbytes)), e.alloc(total + 16)
    d_io, d_il, d_oo, d_ol, d_st = e.alloc(8 * n), e.alloc(4 * n), e.alloc(8 * n), e.alloc(4 * n), e.alloc(4 * n)
    for d, a in ((d_comp, cbytes), (d_io, in_off), (d_il, in_len), (d_oo, out_off), (d_ol, out_len)):
        e.h2d(d, a)
    e.h2d(d_text, np.full(total + 16, 0xA5, np.uint8))
    e.inflate_blocks(d_comp, d_io, d_il, d_oo, d_ol, n, d_text, d_st)
    e.sync()
    status = e.d2h(d_st, (n,), np.int32)
    text = e.d2h(d_text, (total + 16,), np.uint8).tobytes()
    refused = 0
    for k in range(n):
        a, m = int(out_off[k]), int(out_len[k])
        assert text[a + m:a + m + gap] == b"\xa5" * gap, k               # nothing behind the block's text
        if not hurt[k]:
            assert status[k] == 0 and text[a:a + m] == raws[k], k
        elif status[k] != 0:
            refused += 1
        else:                                                           # taken: then it is what zlib makes of the same bytes, if zlib takes them
            try:
                want = zlib.decompressobj(-15).decompress(comps[k])
            except zlib.error:
                want = None
            if want is not None and len(want) == m:
                assert text[a:a + m] == want, k
    assert refused > 100
    e.close()


@pytest.mark.parametrize("strict", [0, 1])
def test_everyday_genotype_stretches_and_what_breaks_them(eng, strict):
    # the tile-parallel tokenizer decodes 32-byte stretches of four-byte genotype fields (d/d, d|d, ./.) eight at a time;
    # every alignment of the fields against the threads' 32 bytes (head lengths 0 .. 40), every digit, and the fields that
    # end such a stretch -- a sub-field, a half-missing or two-digit allele, a haploid call, a short or a long last column --
    # must come out as the oracle's TAB-split + get_alleles gives them
    rng = np.random.default_rng(77 + strict)
    plain = ["%d%s%d" % (a, s, b) for a in range(10) for b in range(10) for s in "/|"] + ["./.", ".|."] * 20
    breakers = ["0/1:7", "./1", "1/.", "10/2", "1", ".", "0/12", "3|4:0,1", "", "0/1/1", "x/1", "1/x"]
    n_samples = 700
    lines = []
    for i in range(90):
        gts = [plain[int(k)] for k in rng.integers(0, len(plain), n_samples)]
        for _ in range(i % 4):                                     # 0 .. 3 fields that break the pattern, anywhere
            gts[int(rng.integers(0, n_samples))] = breakers[int(rng.integers(0, len(breakers)))]
        if i % 9 == 4:
            gts[-1] = "0/1:99"                                     # the line's last field carries a sub-field
        n_cols = n_samples if i % 10 else n_samples - 5           # some lines lack trailing samples
        lines.append(_line(rng, n_samples, "GT" if i % 6 else "GT:DP", "X" if i % 5 == 0 else "7", info_len=i % 41, gts=gts, n_cols=n_cols))
    text = "\n".join(lines) + ("\n" if strict else "")              # with and without a final newline
    got = _check(eng, text, n_samples, strict)
    assert got["n_lines"] == len(lines)
    # the same with a FORMAT that does not begin with GT: nothing is everyday, and (tile-parallel form) lines are re-done
    lines2 = [_line(rng, n_samples, "DP:GT", "7", info_len=i, gts=[plain[int(k)] for k in rng.integers(0, len(plain), n_samples)]) for i in range(12)]
    _check(eng, "\n".join(lines2) + "\n", n_samples, strict)


@pytest.mark.parametrize("strict", [0, 1])
def test_three_byte_fields_of_every_kind(eng, strict):
    # the shape recogniser of the tile-parallel forms (tok_pattern) checks a genotype as one dword: [TAB g s g] with carries, XORs and
    # nibble masks.  Every byte that could slip through such arithmetic is put into the three positions of otherwise everyday
    # columns: the neighbours of the digits ('/' 0x2F, ':' 0x3A ... '?'), 0x2A .. 0x2F (which + 6 turns into 0x3?), bytes that wrap
    # under + 6 (0xFA ..), bytes that differ from '/' or '|' in the separators' own bits (0x7C ^ 0x2F = 0x53), NUL, CR, and mixtures
    # of '/' and '|' inside one stretch of eight.  The oracle's TAB-split + get_alleles says what each field is.
    rng = np.random.default_rng(1234 + strict)
    odd = bytes([0x00, 0x0D, 0x20, 0x2A, 0x2B, 0x2C, 0x2D, 0x2E, 0x2F, 0x30, 0x39, 0x3A, 0x3B, 0x3F, 0x40, 0x50, 0x52, 0x53, 0x54, 0x5C,
                 0x6F, 0x7C, 0x7D, 0x7F, 0x80, 0xAE, 0xAF, 0xCA, 0xD0, 0xF9, 0xFA, 0xFB, 0xFF])
    n_samples = 520
    lines = []
    for i in range(120):
        cols = []
        for j in range(n_samples):
            a, b = b"0123456789"[int(rng.integers(0, 10))], b"0123456789"[int(rng.integers(0, 10))]
            f = bytearray([a, b"/|"[int(rng.integers(0, 2)) if i % 3 == 0 else (i // 3) % 2], b])
            if rng.random() < 0.02:
                f = bytearray(b"./." if rng.random() < 0.5 else b".|.")
            if rng.random() < 0.01 * (1 + i % 5):                   # one position gets an odd byte
                f[int(rng.integers(0, 3))] = odd[int(rng.integers(0, len(odd)))]
            if rng.random() < 0.002:                                 # all three do
                f = bytearray(odd[int(k)] for k in rng.integers(0, len(odd), 3))
            cols.append(bytes(f))
        head = b"\t".join([b"X" if i % 7 == 0 else b"3", b"9", b"r", b"A", b"C", b"5", b"P", b"I" * (i % 37), b"GT"])
        lines.append(head + b"\t" + b"\t".join(cols))
    text = b"\n".join(lines) + (b"\n" if strict else b"")
    got = _check(eng, text, n_samples, strict)
    assert got["n_lines"] == len(lines)


@pytest.mark.parametrize("strict", [0, 1])
def test_lines_of_empty_fields(eng, strict):
    # lines whose header fields are empty or one byte long: several lines -- and several FORMAT fields -- inside one thread's 32
    # bytes.  The tile-parallel forms derive what a TAB or a newline does from the thread's masks; a thread with eight or more TABs
    # behind its first newline goes through them one by one instead, and each FORMAT it meets decides for its own line
    rng = np.random.default_rng(99 + strict)
    n_samples = 5
    fmts = ["GT", "DP:GT", "A:B:GT", "DP", "", "GT:DP"]
    gts = ["0/1", "1|1", "./.", ".", "", "0/1:3", "7:1/0", "1:2:0|1", "2/3:x"]
    lines = []
    for i in range(400):
        head = [["", "X", "1"][int(rng.integers(0, 3))]] + [["", "7", "ab"][int(rng.integers(0, 3))] for _ in range(7)]
        n_cols = int(rng.integers(0, 8))
        cols = [gts[int(rng.integers(0, len(gts)))] for _ in range(n_cols)]
        k = int(rng.integers(0, 12))
        fields = head + [fmts[int(rng.integers(0, len(fmts)))]] + cols
        lines.append("\t".join(fields[:k] if k < 9 and i % 5 == 0 else fields))      # some lines stop inside the header
    text = "\n".join(lines) + ("\n" if strict else "")
    _check(eng, text, n_samples, strict)
    _check(eng, text, 0, strict)
    _check(eng, text, n_samples, strict, max_lines=137)


def test_random_bytes_from_a_small_alphabet(eng):
    # texts no VCF writer would produce: bytes drawn from the characters the tokenizer gives a meaning to (TAB, newline, digits, the
    # separators, '.', ':', the letters of "GT" and "X", CR), in proportions that make every shape of line likely -- headers of
    # empty fields, several lines and several FORMAT fields inside 32 bytes, lines longer than a tile, sample columns of any form.
    # Whatever the oracle's TAB-split + get_alleles makes of them, the three GPU forms must make the same.
    rng = np.random.default_rng(int(os.environ.get("HPGV_FUZZ_SEED", "20260401")))
    for tiles in shipped("tokenizer_tiles", (2, 1, 0)):                                          # a context whose FIRST call is a text of a few bytes (its scratch is sized by that call)
        fresh = hpgv.Engine(0)
        fresh.set_option("tokenizer_tiles", tiles)
        for tiny in (b"1", b"\n", b"1\t2\n", b""):
            _check(fresh, tiny, 3, 1)
        fresh.close()
    alphabets = [(b"\t\n0123./|:GTX", 0.30, 0.03), (b"\t\n01/|.:GT\r9", 0.24, 0.01), (b"\t\n0/1", 0.25, 0.02), (b"\t\nGT:0/1|.", 0.35, 0.08),
                 (b"\t\n01/", 0.50, 0.05), (b"\t\n0:1/GT.", 0.20, 0.004), (b"\t\n1|0", 0.26, 0.0005)]
    for it in range(int(os.environ.get("HPGV_FUZZ_TEXTS", "1200"))):     # (a soak run sets more)
        chars, p_tab, p_nl = alphabets[it % len(alphabets)]          # (the first two characters are TAB and newline)
        p = np.array([p_tab, p_nl] + [(1.0 - p_tab - p_nl) / (len(chars) - 2)] * (len(chars) - 2))
        n = int(rng.choice([0, 1, 7, 31, 32, 33, 200, 1500, 9000, 20000, 70000], p=[0.05, 0.05, 0.1, 0.1, 0.1, 0.1, 0.2, 0.15, 0.1, 0.04, 0.01]))
        text = bytes(np.frombuffer(chars, np.uint8)[rng.choice(len(chars), size=n, p=p / p.sum())])
        if it % 3 == 0 and n > 40:                                  # a well-formed head now and then, so that sample columns are reached
            text = b"1\t2\t.\tA\tC\t.\t.\t.\tGT\t" + text
        n_samples = int(rng.choice([0, 1, 2, 5, 8, 9, 40]))
        strict = it & 1
        _check(eng, text, n_samples, strict)
        if it % 5 == 0:
            _check(eng, text, n_samples, strict, max_lines=max(1, text.count(b"\n") // 2))


def test_bgzf_crc_check_on_the_gpu_against_zlib():
    # hpgv_bgzf_verify_dev: CRC-32 of every decoded block against its BGZF trailer (a damaged stream can inflate to
    # ISIZE bytes of the wrong text).  Lengths around every boundary of the kernel (the 0 - 3 bytes before the first aligned
    # dword, whole 256-byte rows, the rest), text offsets of every alignment, blocks already refused by the decoder left alone
    import zlib
    rng = np.random.default_rng(5)
    lens = [0, 1, 2, 3, 4, 5, 7, 255, 256, 257, 258, 259, 260, 511, 512, 513, 1023, 1024, 4099, 65279, 65280, 65281, 65535, 65536]
    lens += [int(x) for x in rng.integers(0, 65537, 40)]
    texts = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in lens]
    n = len(texts)
    out_len = np.array(lens, np.uint32)
    gaps = rng.integers(0, 4, n)                                        # texts start at every alignment
    out_off = np.zeros(n, np.uint64)
    at = 1
    for k in range(n):
        at += int(gaps[k])
        out_off[k] = at
        at += lens[k]
    total = at + 16
    text = np.zeros(total, np.uint8)
    for k in range(n):
        text[int(out_off[k]): int(out_off[k]) + lens[k]] = np.frombuffer(texts[k], np.uint8)
    # the "compressed" side only needs each block's trailer behind its payload: payload = k % 5 filler bytes
    comp = bytearray()
    in_off, in_len = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    for k in range(n):
        in_off[k], in_len[k] = len(comp), k % 5
        comp += b"\xAA" * (k % 5) + int(zlib.crc32(texts[k])).to_bytes(4, "little") + lens[k].to_bytes(4, "little")
    comp += b"\0" * 16
    wrong_text = [5, 9, 20, 23, n - 1]                                   # one flipped bit in the text
    for k in wrong_text:
        if lens[k]:
            text[int(out_off[k]) + lens[k] // 2] ^= 0x10
    wrong_text = [k for k in wrong_text if lens[k]]
    refused = [3, 30]                                                    # a status the decoder left: not touched, not checked
    status = np.zeros(n, np.int32)
    status[refused] = 4
    e = hpgv.Engine(0)
    d_comp, d_text = e.alloc(len(comp)), e.alloc(total)
    d_io, d_il, d_oo, d_ol, d_st = e.alloc(8 * n), e.alloc(4 * n), e.alloc(8 * n), e.alloc(4 * n), e.alloc(4 * n)
    for d, a in ((d_comp, np.frombuffer(bytes(comp), np.uint8)), (d_text, text), (d_io, in_off), (d_il, in_len), (d_oo, out_off),
                 (d_ol, out_len), (d_st, status)):
        e.h2d(d, a)
    e.bgzf_verify(d_comp, d_io, d_il, d_oo, d_ol, n, d_text, d_st)
    e.sync()
    got = e.d2h(d_st, (n,), np.int32)
    exp = status.copy()
    exp[[k for k in wrong_text if k not in refused]] = 9                 # HPGV_BLOCK_BAD_CRC
    assert np.array_equal(got, exp), (np.nonzero(got != exp)[0], got[got != exp], [lens[k] for k in np.nonzero(got != exp)[0]])
    e.close()


def test_one_sweep_tokenizer_on_a_large_text_equals_two_sweeps():
    """Thousands of segments in flight (look-backs over several rounds, segments that end inside lines, lines longer than a
    segment): the one-sweep tokenizer's matrix, line offsets, field offsets and statuses equal the two-sweep form's, call after
    call (the records are zeroed per call)."""
    rng = np.random.default_rng(99)
    res = {}
    for n_samples, n_lines in ((20000, 3000), (300, 120000), (70000, 400)):
        codes = np.array(["0/0", "0/1", "1/1", "./.", "1|0", "0/2"])
        bodies = ["\t".join(codes[rng.choice(6, size=n_samples, p=[0.5, 0.25, 0.15, 0.02, 0.05, 0.03])]) for _ in range(16)]
        text = "".join("%s\t%d\trs%d\tA\tG,T\t.\tPASS\t%s\tGT\t%s\n" % (["1", "X"][i % 2], 100 + i, i, "X" * int(rng.integers(1, 200)), bodies[i % 16])
                       for i in range(n_lines))
        for tiles in shipped("tokenizer_tiles", (2, 1, 0)):
            e = hpgv.Engine(0)
            e.set_option("tokenizer_tiles", tiles)
            out = [e.tokenize(text, n_samples, True, None) for _ in range(2 if tiles else 1)]
            e.close()
            res[tiles] = out
        for call in range(2):
            b = res[1][call]
            assert b["n_lines"] == n_lines
            if 2 in res:                                             # (the one-sweep form: an ablation build's)
                a = res[2][call]
                assert a["n_lines"] == n_lines
                for k in ("gt", "is_x", "status", "line_off", "field_off"):
                    assert np.array_equal(a[k], b[k]), (n_samples, k)
        # the line-by-line form shares no parsing code with the tile-parallel ones (tok_parse_tile): the same matrix from it as well
        if 0 in res:
            c = res[0][0]
            assert c["n_lines"] == n_lines
            for k in ("gt", "is_x", "status", "line_off", "field_off"):
                assert np.array_equal(res[1][0][k], c[k]), (n_samples, k)
        # and the matrix is what the text says: every line's body is one of the 16, whose codes the oracle gives
        want = np.stack([orc.tokenize(("1\t1\t.\tA\tG,T\t.\t.\t.\tGT\t" + bd + "\n"), n_samples, True)["gt"][0] for bd in bodies])
        assert np.array_equal(res[1][0]["gt"], want[np.arange(n_lines) % 16])


@pytest.mark.parametrize("n_samples", [17, 1000, 4100, 8200, 16390, 33000])
def test_assoc_text_rows_kernel_equals_the_per_row_kernel(n_samples):
    """hpgv_assoc_text counts with k_assoc_rows (threads own columns across a band of rows; one to four chunks per thread, 64 to 512
    threads) -- against k_batch (HPGV_ASSOC_ROWS=0: one workgroup per row; also what wider cohorts fall back to) and the oracle:
    chromosome X rows, samples that are neither affected nor unaffected, half-missing and multi-allelic calls."""
    import os
    from helpers import check_assoc, oracle_assoc
    rng = np.random.default_rng(n_samples)
    lines = [_line(rng, n_samples, "GT", ["1", "X", "7", "X"][i % 4], 6) for i in range(37)]
    text = "\n".join(lines) + "\n"
    tok = orc.tokenize(text, n_samples, True)
    cond = rng.choice([0, 1, 2], size=n_samples, p=[0.45, 0.45, 0.1]).astype(np.uint8)
    out = {}
    for rows in ("1", "0"):
        os.environ["HPGV_ASSOC_ROWS"] = rows
        try:
            e = hpgv.Engine(0)
            e.set_cohort(cond)
            out[rows] = e.assoc_text(hpgv.TASK_CHISQ, text)
            e.close()
        finally:
            del os.environ["HPGV_ASSOC_ROWS"]
    for k in ("A1", "A2", "U1", "U2", "odds", "chisq", "p", "status"):
        assert np.array_equal(out["1"][k], out["0"][k], equal_nan=True), k
    check_assoc(out["1"], oracle_assoc(orc.TASK_CHISQ, tok["gt"], cond, tok["is_x"], None), hpgv.TASK_CHISQ)


def test_assoc_text_rows_kernel_on_random_widths():
    # cohort widths drawn at random (HPGV_SOAK_SHAPES of them: 5 in the suite, hundreds in a soak run) through hpgv_assoc_text's two
    # counting kernels and the oracle
    rng = np.random.default_rng(int(os.environ.get("HPGV_FUZZ_SEED", "78")))
    for _ in range(int(os.environ.get("HPGV_SOAK_SHAPES", "5"))):
        n_samples = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 5000)), int(rng.integers(5000, 34000)), 16 * int(rng.integers(1, 2100))]))
        test_assoc_text_rows_kernel_equals_the_per_row_kernel(n_samples)


@pytest.mark.parametrize("n_samples,block_bytes,text_shift", [(200, [65280], 0), (2504, [65280], 0), (37, [65280, 1000, 500, 700, 3000, 65280, 10], 0),
                                                              (900, [4096, 65280, 2047, 2048, 2049], 0), (300, [65280, 5000], 4), (2504, [65280], 12)])
def test_windows_of_decoded_text_tokenized_from_the_decoders_tile_records(n_samples, block_bytes, text_shift):
    """The bgzip decoder's CRC kernel leaves the tokenizer's tile records of the text it checks (hpgv_bgzf_verify_tiles_dev); windows of
    that text -- starting at any line, ending at any line -- are then tokenized WITHOUT the counting sweep (hpgv_text_alias_tiles).
    Same outputs, bit for bit, as the ordinary two-sweep call on the same window, and as the oracle: blocks of the size bgzip writes
    (seams inside tiles), blocks shorter than a tile (several seams in one tile: counted again), a block the decoder refused (its
    text patched afterwards: counted again)."""
    import zlib
    rng = np.random.default_rng(n_samples)
    fmts = ["GT", "GT:DP", "DP:GT", "GT"]
    lines = [_line(rng, n_samples, fmts[i % 4], ["1", "X", "7"][i % 3], int(rng.integers(3, 300))) for i in range(max(40, 600_000 // (4 * n_samples + 60)))]
    text = ("\n".join(lines) + "\n").encode()
    raws, pos, k = [], 0, 0
    while pos < len(text):
        n = block_bytes[k % len(block_bytes)]; k += 1
        raws.append(text[pos: pos + n]); pos += n
    comps = []
    for r in raws:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comps.append(co.compress(r) + co.flush() + zlib.crc32(r).to_bytes(4, "little") + len(r).to_bytes(4, "little"))      # payload + the BGZF trailer
    refused = len(raws) // 3
    comps[refused] = bytes([comps[refused][0] | 0x06]) + comps[refused][1:]          # block type 3: the decoder refuses it
    n = len(raws)
    in_len = np.array([len(c) - 8 for c in comps], np.uint32); out_len = np.array([len(r) for r in raws], np.uint32)
    in_off = np.concatenate([[0], np.cumsum([len(c) for c in comps[:-1]], dtype=np.uint64)]).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(out_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    cbytes = np.frombuffer(b"".join(comps) + b"\0" * 16, np.uint8)
    total = len(text)
    e = hpgv.Engine(0)
    # (text_shift: a text that does not begin on a 16-byte boundary -- the records then come from a sweep of their own, not out of
    # the CRC loop's loads)
    d_comp, d_text_alloc = e.alloc(len(cbytes)), e.alloc(total + 64 + 16)
    d_text = C.c_void_p(d_text_alloc.value + text_shift)
    d_io, d_il, d_oo, d_ol, d_st = e.alloc(8 * n), e.alloc(4 * n), e.alloc(8 * n), e.alloc(4 * n), e.alloc(4 * n)
    for d, a in ((d_comp, cbytes), (d_io, in_off), (d_il, in_len), (d_oo, out_off), (d_ol, out_len)):
        e.h2d(d, a)
    tiles_bytes = int(e.L.hpgv_text_tiles_bytes(total))
    n_tiles = tiles_bytes // 32 - 1
    d_tiles = e.alloc(tiles_bytes)
    e.h2d(d_tiles, np.zeros(tiles_bytes, np.uint8))
    e.inflate_blocks(d_comp, d_io, d_il, d_oo, d_ol, n, d_text, d_st)
    e.bgzf_verify_tiles(d_comp, d_io, d_il, d_oo, d_ol, n, d_text, d_st, d_tiles, n_tiles)
    e.sync()
    status = e.d2h(d_st, (n,), np.int32)
    assert status[refused] != 0 and (np.delete(status, refused) == 0).all()
    # the host decodes the refused block and patches the text, as the file runner does
    e.h2d(C.c_void_p(d_text.value + int(out_off[refused])), np.frombuffer(raws[refused], np.uint8))
    assert e.d2h(d_text, (total,), np.uint8).tobytes() == text
    starts = [0] + [i + 1 for i, ch in enumerate(text) if ch == 10]
    ends = starts[1:]
    picks = [(0, len(ends) - 1), (1, len(ends) - 1), (0, 0), (len(ends) // 2, len(ends) // 2)]
    picks += [tuple(sorted(int(x) for x in rng.integers(0, len(ends), 2))) for _ in range(12)]
    max_lines = len(ends) + 2
    pitch = (n_samples + 15) // 16 * 16
    d_nl, d_lo, d_fo = e.alloc(16), e.alloc(8 * (max_lines + 2)), e.alloc(40 * max_lines + 16)
    d_gt, d_x, d_stt = e.alloc(max_lines * pitch + 16), e.alloc(max_lines + 16), e.alloc(4 * max_lines + 16)
    host_key = C.create_string_buffer(16)                            # the alias table is keyed by a host address: any will do
    L = e.L
    for first, last in picks:
        a, b = starts[first], ends[last]
        win = C.c_void_p(d_text.value + a)
        out = {}
        for mode in ("tiles", "plain"):
            if mode == "tiles":
                assert L.hpgv_text_alias_tiles(e.h, host_key, win, d_text, d_tiles, n_tiles) == 0
            else:
                assert L.hpgv_text_alias(e.h, host_key, None) == 0
            for d, nb in ((d_gt, max_lines * pitch), (d_x, max_lines), (d_stt, 4 * max_lines), (d_fo, 40 * max_lines), (d_lo, 8 * (max_lines + 2))):
                e.h2d(d, np.full(nb, 0xA5, np.uint8))
            e._chk(L.hpgv_tokenize_dev(e.h, win, b - a, n_samples, 1, max_lines, d_nl, d_lo, d_fo, d_gt, pitch, d_x, d_stt, None))
            e.sync()
            nl = int(e.d2h(d_nl, (1,), np.int32)[0])
            out[mode] = dict(n=nl, gt=e.d2h(d_gt, (nl, pitch), np.uint8)[:, :n_samples], x=e.d2h(d_x, (nl,), np.uint8), st=e.d2h(d_stt, (nl,), np.int32),
                             lo=e.d2h(d_lo, (nl + 1,), np.uint64), fo=e.d2h(d_fo, (nl, 10), np.uint32))
        assert L.hpgv_text_alias(e.h, host_key, None) == 0
        assert out["tiles"]["n"] == out["plain"]["n"] == last - first + 1, (first, last)
        for kk in ("gt", "x", "st", "lo", "fo"):
            assert np.array_equal(out["tiles"][kk], out["plain"][kk]), (first, last, kk)
        tok = orc.tokenize(text[a:b].decode(), n_samples, True)
        assert np.array_equal(out["tiles"]["gt"], tok["gt"]) and np.array_equal(out["tiles"]["st"], tok["status"])
        assert out["tiles"]["lo"].tolist() == [s - a for s in starts[first: last + 2]]
    e.close()
