"""-m gpu: a slice of the randomised differential campaign (tests/fuzzlib.py; scripts/fuzz_parity.py runs it at length)."""
import pytest

import fuzzlib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,resident", [(11, False), (12, False), (13, True)])
def test_random_streams_match_the_cpu_model(gpu, seed, resident):
    bad = []
    for case in range(40):
        ok, info = fuzzlib.run_case(case, seed, resident)
        if not ok:
            bad.append((case, info))
    assert not bad, bad
