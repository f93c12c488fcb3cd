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


@pytest.fixture(params=[512, 768], ids=["D512", "D768"])
def decim(request):
    """input samples per filter-bank frame: 512 = the default form (60 ksps per channel, 3 samples per symbol), 768 = the 4/3 x
    oversampled one (40 ksps, 2 samples per symbol)"""
    return request.param


def wb_cfg(decim, first, P=8, **kw):
    """the `wideband=` dictionary of capi.Recc for a decimation, and the samples per symbol that go with it"""
    d = {"channels": 1024, "decim": decim, "taps_per_branch": P, "first_channel": first}
    d.update(kw)
    return d, 1536 // decim
