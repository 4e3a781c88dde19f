"""CPU suite: the C-ABI libraries load and export every symbol the headers
declare (no compute calls: there is no GPU here), the product fails loudly
without a device, and the oracle is not reachable from the product."""
import ctypes as C
import os
import re
import subprocess

import pytest

from helpers import hpgv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\((?!\*)", text))
    return sorted(names - {"defined", "sizeof", "void", "int", "char"})


@pytest.fixture(scope="module", autouse=True)
def built():
    hpgv.build()


def test_libhpgv_exports_every_declared_symbol():
    L = hpgv.load()
    names = [n for n in _declared("hpgv.h") if n.startswith("hpgv_")]
    assert set(names) == set(hpgv.SYMBOLS), set(names) ^ set(hpgv.SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_libhpgv_host_exports_reference_api():
    from importlib import import_module
    b = import_module("hpg-variant_amd._build")
    L = C.CDLL(b.HOSTLIB)
    names = [n for n in _declared("hpgv_host.h") if not n.startswith("pthread")]
    for n in names:
        assert hasattr(L, n), n
    # the three per-batch entry points of the reference are there by name
    for n in ("assoc_test", "tdt_test", "get_variants_stats", "init_logarithm_array"):
        assert n in names


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hpgv.HpgvError) as e:
        hpgv.Engine(0)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_link_or_import_the_oracle():
    pkg = os.path.join(ROOT, "hpg-variant_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', src), f
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, flags=re.M), f
                assert "orc_" not in re.sub(r"//.*|/\*.*?\*/", "", src, flags=re.S), f
    from importlib import import_module
    b = import_module("hpg-variant_amd._build")
    for lib in (b.LIB, b.HOSTLIB):
        out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
        assert "oracle" not in out


def test_status_codes_and_enums_match_header():
    text = open(os.path.join(ROOT, "include", "hpgv.h")).read()
    assert "HPGV_TASK_CHISQ = 1" in text and hpgv.TASK_CHISQ == 1
    assert "HPGV_TASK_FISHER = 2" in text and hpgv.TASK_FISHER == 2
    assert "HPGV_COND_AFFECTED = 1" in text and hpgv.COND_AFFECTED == 1
    assert "HPGV_SEX_MALE = 0" in text and hpgv.SEX_MALE == 0
