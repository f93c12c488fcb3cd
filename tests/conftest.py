import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every drain of the suite cross-checks the list header the capture kernel published against the device-side counters
os.environ.setdefault("AMPS_RECC_CHECK_HEADER", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gr_amps_amd import capi
    capi.load()
    return torch.device("cuda:0")
