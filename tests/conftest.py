import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip silently; the
    # tests themselves raise when the HIP library cannot reach a device.
    pass


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Builds libhpgv.so / libhpgv_host.so / the oracle when they are missing or stale (no-op otherwise).
    Building is not a fallback: without a GPU the engine still refuses to run."""
    import importlib
    hpgv = importlib.import_module("hpg-variant_amd")
    hpgv.build()
    from oracle import pyoracle
    pyoracle.build()


@pytest.fixture(scope="session")
def goldens():
    import json
    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "stat_goldens.json")) as f:
        stat = json.load(f)
    with open(os.path.join(here, "reference_kats.json")) as f:
        kats = json.load(f)
    return {"stat": stat, "kats": kats}
