import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a test marked gpu on a box without a GPU is an error of the invocation, not a skip:
    # the driver selects with -m gpu / -m "not gpu"
    pass


def load_golden(name):
    path = os.path.join(ROOT, "tests", "golden", name)
    z = np.load(path)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "iuf" else z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden_toy():
    return load_golden("toy.npz")


@pytest.fixture(scope="session")
def golden_mag():
    return load_golden("mag01_slice.npz")


@pytest.fixture(scope="session")
def golden_mag_full():
    """The whole shipped ogbn_mag_0.1 topology + reference outputs (tests/golden/make_golden.py::main_full)."""
    return load_golden("mag01_full.npz")
