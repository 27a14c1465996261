import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_sessionfinish(session, exitstatus):
    import parity_ledger
    path = parity_ledger.flush()
    if path:
        print(f"\nparity ledger: {len(parity_ledger.rows())} rows -> {path}")


def sha(a) -> str:
    if isinstance(a, str):
        a = a.encode("utf-8")
    elif isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a).tobytes()
    return hashlib.sha256(a).hexdigest()


def npz_str(arr) -> str:
    return bytes(np.asarray(arr, dtype=np.uint8)).decode()


@pytest.fixture(scope="session")
def golden_json():
    def load(name):
        with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
            return json.load(f)
    return load


@pytest.fixture(scope="session")
def golden_npz():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def big_tile():
    """The 5000x5000 synthetic tile (seed 1000) shared by the big-size tests."""
    from oracle import prng
    return prng.synthetic_tile(1000, 5000, 5000)
