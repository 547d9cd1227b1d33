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
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def toy():
    return load_golden("toy_all_models.npz")


@pytest.fixture(scope="session")
def c1():
    return load_golden("c1_bpr_ml100k_shape.npz")


@pytest.fixture(scope="session")
def unknowns():
    return load_golden("unknowns_mf_bpr.npz")
