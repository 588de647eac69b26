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


@pytest.fixture(scope="session", autouse=True)
def _torch_before_engine(request):
    """`-m gpu` runs: bring up torch's HIP runtime BEFORE the first hmme context exists.  Both live in this process (torch for device
    tensors / streams in some tests, libhmme.so for everything else); with the engine first, a later torch.cuda initialisation was
    seen to fail with "No HIP GPUs are available" on the GPU box (round 2, a -k subset run) although the engine kept working."""
    if request.config.getoption("markexpr", "").strip() == "gpu":
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    yield
