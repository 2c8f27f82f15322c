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


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def pair():
    ref = np.load(os.path.join(GOLDEN, "cloud17.npy"))
    src = np.load(os.path.join(GOLDEN, "cloud15.npy"))
    gold = np.load(os.path.join(GOLDEN, "pair_15_17.npz"))
    return ref, src, gold


@pytest.fixture(scope="session")
def synth32():
    return np.load(os.path.join(GOLDEN, "synth32.npz"))


@pytest.fixture(scope="session")
def lattice():
    return np.load(os.path.join(GOLDEN, "lattice_ties.npz"))


@pytest.fixture()
def ctx():
    import stereovision_amd as sv
    c = sv.Context(0)
    yield c
    c.close()
