"""-m gpu: error behaviour of the C ABI on the IQ seam -- what a caller sees when a list overflows, and that the handle keeps
working afterwards (the record lists' counters and status words are cleared by the kernels of later pushes, not by memsets)."""
import errno

import numpy as np
import pytest

import oracle
from gr_amps_amd import capi, synth

pytestmark = pytest.mark.gpu
SPS = 10
FULL = 41 + 7 + 7 * 240


def _stream(nbursts, seed, gap_syms=3600):
    rng = np.random.default_rng(seed)
    bursts, off = [], 2500
    for _ in range(nbursts):
        _, _, _, _, words = synth.random_message(rng)
        bursts.append((off, synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)))
        off += gap_syms * SPS
    n = off + 6000
    return synth.fsk_modulate(n, bursts, sps=SPS, fs=20e3 * SPS, snr_db=30.0, rng=rng)[None, :]


def _pad(iq, n_channels):
    """the one real channel plus idle ones: 64 channels and more put a handle on the fused form of the capture stage (the resolve
    kernel's own workgroup decodes), fewer on the queue form (recc_resolve.hip.h) -- the overflow paths differ"""
    x = np.zeros((n_channels, iq.shape[1]), iq.dtype)
    x[:1] = iq
    return x


@pytest.mark.parametrize("C", [1, 65])
def test_record_list_overflow_is_reported_and_the_handle_recovers(gpu, C):
    """five bursts into a handle built for two: drain says -ENOSPC (the reference has no such limit: its message port queues),
    and the stream goes on -- the next drains are clean and complete, on both record lists"""
    iq5, iq2 = _stream(5, 1), _stream(2, 2)
    parts = [iq5, iq2, iq2, iq2]
    want = oracle.fused_push_all(np.concatenate(parts, axis=1), sps=SPS)
    assert len(want) == 11
    iq5, iq2 = _pad(iq5, C), _pad(iq2, C)
    with capi.Recc(n_channels=C, sps=SPS, max_samples=max(iq5.shape[1], iq2.shape[1]), max_bursts=2) as r:
        r.push_iq(iq5)
        with pytest.raises(capi.AmpsError) as e:
            r.drain()
        assert e.value.code == -errno.ENOSPC
        for i in range(3):                       # the stream goes on; both lists get reused
            r.push_iq(iq2)
            got = r.drain()
            assert got.tobytes() == want[5 + 2 * i:7 + 2 * i].tobytes()
            assert len(r.drain()) == 0


@pytest.mark.parametrize("C", [1, 65])
def test_overflow_in_a_split_drain_does_not_leak_into_the_next_list(gpu, C):
    iq5, iq2 = _pad(_stream(5, 3), C), _pad(_stream(2, 4), C)
    with capi.Recc(n_channels=C, sps=SPS, max_samples=max(iq5.shape[1], iq2.shape[1]), max_bursts=2) as r:
        r.push_iq(iq5)
        r.drain_begin()
        with pytest.raises(capi.AmpsError) as e:
            r.drain_end()
        assert e.value.code == -errno.ENOSPC
        r.reset()
        r.push_iq(iq2)
        r.drain_begin()
        assert len(r.drain_end()) == 2
        r.drain_begin()                          # nothing pushed since: empty, twice in a row
        assert len(r.drain_end()) == 0
        assert len(r.drain()) == 0


def test_argument_errors(gpu):
    iq = _stream(1, 5)
    n = iq.shape[1]
    with capi.Recc(n_channels=1, sps=SPS, max_samples=n // 2, max_bursts=4) as r:
        with pytest.raises(capi.AmpsError) as e:
            r.push_iq(iq)                         # larger than max_samples_per_push
        assert e.value.code == -errno.E2BIG
        with pytest.raises(capi.AmpsError) as e:
            r.push_iq(iq[:, :1000], nsamp=2000)   # more samples than the row holds
        assert e.value.code == -errno.EINVAL
        with pytest.raises(capi.AmpsError) as e:
            r.drain_end()                         # no split drain open
        assert e.value.code == -errno.EINVAL
        r.drain_begin()
        with pytest.raises(capi.AmpsError) as e:
            r.drain_begin()                       # one at a time
        assert e.value.code == -errno.EBUSY
        with pytest.raises(capi.AmpsError) as e:
            r.drain()
        assert e.value.code == -errno.EBUSY
        assert len(r.drain_end()) == 0
    with pytest.raises(capi.AmpsError) as e:
        capi.Recc(n_channels=1, sps=7, max_samples=1024, max_bursts=4)     # no kernel for 7 samples per symbol
    assert e.value.code == -errno.EINVAL
