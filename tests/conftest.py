import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    """max |a-b| / max(|b|) — the 'within 1e-5 rel' yardstick of BASELINE.json north_star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / denom


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_bprmf_step")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_ml100k_curve")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_sampler")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_lightgcn")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_sasrec_emb")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_eval")


@pytest.fixture(scope="session")
def g10():
    return load_golden("g10_optimizers")
