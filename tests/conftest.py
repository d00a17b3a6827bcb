import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    for k in ("edge_index", "edge_type"):           # the larger fixtures store the COO lists as int32; the reference's are int64
        if k in g and g[k].dtype != np.int64:
            g[k] = g[k].astype(np.int64)
    return g
