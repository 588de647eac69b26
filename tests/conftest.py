import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "hm-opencl_amd"), ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_py
    oracle_py.build(ref=os.path.isdir("/root/reference/source"))
    return oracle_py


@pytest.fixture(scope="session")
def slots():
    import numpy as np
    t = np.load(os.path.join(GOLDEN, "slots.npz"))["table"]
    return t[np.argsort(t[:, 0])]
