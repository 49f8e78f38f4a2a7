"""pytest configuration: markers + shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol export.
`-m gpu` runs on an MI355X: parity of the HIP path (through the C-ABI) against the oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ROBOTS = ["ur5", "iiwa14", "panda", "xarm6"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


PKG_DATA = os.path.join(ROOT, "manipulapy_amd", "data")  # model tables + URDFs of the benchmark robots ship with the package


def golden_path(name):
    if name.startswith("model_") and name.endswith(".npz"):
        return os.path.join(PKG_DATA, name)
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def tables():
    from oracle import ref_numpy as ref

    return {r: ref.load_tables(golden_path(f"model_{r}.npz")) for r in ROBOTS}


@pytest.fixture(scope="session")
def dyn_golden():
    return {r: dict(np.load(golden_path(f"dynamics_{r}.npz"))) for r in ROBOTS}
