"""Several devices behind one context (hpgv_create_multi) and behind the host library (HPGV_DEVICES /
hpgv_host_init_devices).  A one-GPU box cannot show a speed-up, but it can show that the dealing is correct: a group of
two contexts on device 0 must give the results of one context, from concurrent worker threads and from the file runners,
byte for byte."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from helpers import check_assoc, hpgv, make_families, oracle_assoc, random_codes
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_context_deals_concurrent_batches():
    n_samples = 3001
    rng = np.random.default_rng(11)
    cond = rng.integers(0, 3, n_samples).astype(np.uint8)
    g = hpgv.Engine([0, 0])
    assert g.L.hpgv_group_size(g.h) == 2 and g.L.hpgv_member_device(g.h, 1) == 0
    g.set_cohort(cond)                                        # goes to both members
    lf = orc.logfact(n_samples * 10)
    g.set_logfact(lf)
    fam = make_families(rng, n_samples, 700, max_children=3)
    g.set_families(n_samples, *fam)
    batches = [random_codes(rng, 150 + 13 * k, n_samples, quirks=True, strict=True) for k in range(12)]
    out = [None] * len(batches)

    def work(k):
        task = hpgv.TASK_CHISQ if k % 2 else hpgv.TASK_FISHER
        out[k] = (task, g.assoc(task, batches[k]), g.tdt(batches[k]))
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(batches))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k, (task, res, tdt) in enumerate(out):
        check_assoc(res, oracle_assoc(task, batches[k], cond, None, lf), task)
        t1, t2 = orc.tdt_counts(batches[k], *fam)
        assert np.array_equal(tdt["t1"], t1) and np.array_equal(tdt["t2"], t2)
    # device-resident calls of a group refer to member 0; the other member is reachable on its own
    m1 = g.L.hpgv_group_member(g.h, 1)
    assert m1 and g.L.hpgv_group_member(g.h, 2) is None
    nA = C.c_int()
    assert g.L.hpgv_assoc_layout(m1, C.byref(nA), None, None) == 0 and nA.value == int((cond == 1).sum())
    g.close()


_RUNNER = r"""
import ctypes as C, sys, importlib
sys.path.insert(0, %(root)r)
b = importlib.import_module("hpg-variant_amd._build")
L = C.CDLL(b.HOSTLIB)
L.hpgv_run_assoc.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_size_t, C.POINTER(C.c_long)]
L.hpgv_run_tdt.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
L.hpgv_host_last_error.restype = C.c_char_p
vcf, ped, out = [a.encode() for a in sys.argv[1:4]]
n = C.c_long(0)
rc = L.hpgv_run_assoc(vcf, ped, out + b".chisq", 1, 1 << 16, C.byref(n))
assert rc == 0, L.hpgv_host_last_error()
rc = L.hpgv_run_assoc(vcf, ped, out + b".fisher", 2, 1 << 16, C.byref(n))
assert rc == 0, L.hpgv_host_last_error()
rc = L.hpgv_run_tdt(vcf, ped, out + b".tdt", 1 << 16, C.byref(n))
assert rc == 0, L.hpgv_host_last_error()
L.hpgv_run_stats.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
rc = L.hpgv_run_stats(vcf, ped, out, 1 << 16, C.byref(n))
assert rc == 0, L.hpgv_host_last_error()
print(L.hpgv_host_device_count(), n.value)
L.hpgv_host_shutdown()
"""


def test_runners_with_hpgv_devices_are_byte_identical(tmp_path):
    """hpgv_run_assoc / hpgv_run_tdt / hpgv_run_stats with HPGV_DEVICES=0,0,0 (three contexts, six engine threads) write the same
    files as with one device -- and the one-pass kernels (default) the same files as the kernel chains (HPGV_BATCH_FUSED=0)."""
    from test_file_runner_gpu import _vcf_from_batch
    from test_host_mirror_gpu import _write_inputs
    rng = np.random.default_rng(21)
    people, names, rows = _write_inputs(tmp_path, rng, 60, 30, 4000)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    ped = str(tmp_path / "ped.txt")
    script = tmp_path / "run.py"
    script.write_text(_RUNNER % {"root": ROOT})
    outs = {}
    exts = (".chisq", ".fisher", ".tdt", ".stats-variants", ".stats-samples", ".stats-summary")
    for tag, devs in (("one", None), ("three", "0,0,0"), ("chains", None)):
        env = {k: v for k, v in os.environ.items() if k not in ("HPGV_DEVICES", "HPGV_BATCH_FUSED")}
        if devs:
            env["HPGV_DEVICES"] = devs
        if tag == "chains":
            env["HPGV_BATCH_FUSED"] = "0"
        r = subprocess.run([sys.executable, str(script), vcf, ped, str(tmp_path / tag)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        n_dev, n_rec = r.stdout.split()
        assert int(n_dev) == (3 if devs else 1) and int(n_rec) == len(rows)
        outs[tag] = [open(str(tmp_path / tag) + ext, "rb").read() for ext in exts]
        import glob
        outs[tag] += [open(f, "rb").read() for f in sorted(glob.glob(str(tmp_path / tag) + ".phenotype-*"))]
    assert len(outs["one"]) > len(exts)                        # the per-phenotype files are there too
    for k in range(len(outs["one"])):
        assert outs["one"][k] == outs["three"][k], k
        assert outs["one"][k] == outs["chains"][k], k
    assert all(len(x) > 100 for x in outs["one"])


def test_bad_device_list_is_refused(tmp_path):
    script = tmp_path / "run.py"
    script.write_text(_RUNNER % {"root": ROOT})
    env = dict(os.environ, HPGV_DEVICES="0,banana")
    r = subprocess.run([sys.executable, str(script), "x.vcf", "x.ped", str(tmp_path / "o")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "HPGV_DEVICES" in r.stderr


@pytest.mark.parametrize("devs,block", [("0,0", 0x700), ("0,0,0", 0x700), ("0,0,0,0", 0x3000), ("0,0", 0xff00)])
def test_bgzip_file_staged_in_parts_one_per_device_is_byte_identical(tmp_path, devs, block):
    """the bgzip DEVICE path on every device of the group.  The file is cut at block starts into one part
    per member; every part is uploaded to, decoded on and tokenized on its own device; the line that straddles two parts is
    joined on the host and goes through as a batch of its own.  HPGV_DEVICES=0,0 / 0,0,0 / 0,0,0,0 on the one GPU of the box:
    same files as one device, byte for byte (assoc, tdt and the stats tool's files), and the trace says the parts were used."""
    from test_file_runner_gpu import _vcf_from_batch
    from test_host_logic_cpu import _bgzf
    from test_host_mirror_gpu import _write_inputs
    rng = np.random.default_rng(len(devs) + block)
    people, names, rows = _write_inputs(tmp_path, rng, 60, 30, 6000)
    vcf = _vcf_from_batch(tmp_path, names, rows)
    packed = str(tmp_path / "in.vcf.gz")
    data = open(vcf, "rb").read()
    open(packed, "wb").write(_bgzf(data, block))
    ped = str(tmp_path / "ped.txt")
    script = tmp_path / "run.py"
    script.write_text(_RUNNER % {"root": ROOT})
    exts = (".chisq", ".fisher", ".tdt", ".stats-variants", ".stats-samples", ".stats-summary")
    outs = {}
    for tag, d in (("one", None), ("parts", devs)):
        env = {k: v for k, v in os.environ.items() if k not in ("HPGV_DEVICES",)}
        env["HPGV_RUN_TRACE"] = "1"
        env["HPGV_BGZF_PART_MIN_KB"] = "64"
        if d:
            env["HPGV_DEVICES"] = d
        r = subprocess.run([sys.executable, str(script), packed, ped, str(tmp_path / tag)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        n_dev, n_rec = r.stdout.split()
        assert int(n_rec) == len(rows)
        n_parts = min(len(d.split(",")), os.path.getsize(packed) // (64 << 10)) if d else 0
        assert ("stage: %d parts, one per device" % n_parts in r.stderr) == bool(d and n_parts >= 2), r.stderr[-3000:]
        outs[tag] = [open(str(tmp_path / tag) + ext, "rb").read() for ext in exts]
    for k in range(len(exts)):
        assert outs["one"][k] == outs["parts"][k], exts[k]
    assert all(len(x) > 100 for x in outs["one"])


def test_bgzip_parts_with_random_block_sizes(tmp_path_factory):
    # block sizes and part counts drawn at random (HPGV_SOAK_SHAPES of them: one in the suite, dozens in a soak run): the parts' seams
    # fall anywhere in a line, a block, a window
    rng = np.random.default_rng(int(os.environ.get("HPGV_FUZZ_SEED", "79")))
    for _ in range(int(os.environ.get("HPGV_SOAK_SHAPES", "1"))):
        devs = ",".join(["0"] * int(rng.integers(2, 5)))
        block = int(rng.choice([int(rng.integers(0x200, 0x1000)), int(rng.integers(0x1000, 0x8000)), int(rng.integers(0x8000, 0xff01))]))
        test_bgzip_file_staged_in_parts_one_per_device_is_byte_identical(tmp_path_factory.mktemp("parts"), devs, block)
