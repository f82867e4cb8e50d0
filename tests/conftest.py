"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI load/export checks (no device).
`-m gpu`       : parity tests proper -- every call goes through the C ABI into the HIP
                 library on a real MI355X and is compared with the oracle / goldens.
"""

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs through the HIP library)")


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(GOLDEN_DIR, "cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_data():
    return np.load(os.path.join(GOLDEN_DIR, "golden.npz"))


@pytest.fixture(autouse=True)
def _sample_rate():
    import pygmu2_amd as pg
    pg.set_sample_rate(44100)
    yield
