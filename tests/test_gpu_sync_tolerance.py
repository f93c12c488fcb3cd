"""Tolerant sync: cfg.sync_tolerance = k accepts a trigger with up to k of its 74 Manchester symbols
wrong.  It diverges from the reference's exact memmem (lib/recc_impl.cc:118) and is off by default; the oracle here is
the CPU model of the fused seam with the same k (oracle/fused_model.c), compared bit-exact, plus the transmitted truth."""
import numpy as np
import pytest

import oracle
from gr_amps_amd import capi, synth, synth_wideband as sw

pytestmark = pytest.mark.gpu


def _block_with_damaged_preambles(seed, flips_per_burst, sps=10, snr_db=30.0):
    """One channel; burst i has flips_per_burst[i] bits of the dotting / word-sync inverted at the transmitter (each
    wrong bit = two wrong Manchester symbols at the receiver)."""
    rng = np.random.default_rng(seed)
    burst_len = (synth.BURST_PREFIX_BITS + 7 + 7 * 240) * 2 * sps
    spacing = burst_len + (74 + 400) * sps
    bursts, truth, off = [], [], 3000
    for nflip in flips_per_burst:
        kind, min10, esn, dialed, words = synth.random_message(rng)
        bits = synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)
        # the trigger covers the last 26 dotting bits and the 11 sync bits = burst bits [4, 41)
        for p in rng.choice(np.arange(6, 39), size=nflip, replace=False):
            bits[int(p)] ^= 1
        bursts.append((off, bits))
        truth.append((min10, nflip))
        off += spacing + int(rng.integers(0, 500))
    iq = synth.fsk_modulate(off + 2000, bursts, sps=sps, fs=20e3 * sps, snr_db=snr_db, rng=rng)
    return iq, truth


@pytest.mark.parametrize("sps", [10, 4])
def test_tolerant_sync_finds_damaged_preambles_and_matches_its_cpu_model(gpu, sps):
    flips = [0, 1, 2, 0, 1, 3, 2, 0]
    iq, truth = _block_with_damaged_preambles(900 + sps, flips, sps=sps)
    found = {}
    for k in (0, 2, 4):
        with capi.Recc(n_channels=1, sps=sps, max_samples=iq.size, max_bursts=64, sync_tolerance=k) as r:
            for part in np.array_split(iq, 3):
                r.push_iq(part[None, :])
            got = r.drain()
        want = oracle.fused_push_all(iq[None, :], sps=sps, tolerance=k)
        assert got.tobytes() == want.tobytes(), k
        found[k] = {g["min"].decode() for g in got}
    # a wrong bit costs two symbols (plus, rarely, a neighbour through the boxcar): k = 2f finds every burst with <= f flips
    for k in (0, 2, 4):
        must = {m for m, f in truth if 2 * f <= k}
        assert must <= found[k], (k, must - found[k])
    assert found[0] < found[2] <= found[4]
    assert {m for m, f in truth if f == 3}.isdisjoint(found[0])


def test_tolerance_zero_and_nonzero_agree_on_clean_signals(gpu):
    C, N = 6, 3 * 40000
    chans = [synth.make_channel_block(N, 2, seed=950 + c)[0] for c in range(C)]
    iq = np.stack(chans)
    outs = []
    for k in (0, 1, 5, 8):
        with capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=64, sync_tolerance=k) as r:
            r.push_iq(iq)
            outs.append(r.drain())
    assert len(outs[0]) == 2 * C
    for o in outs[1:]:       # same bursts and words; a tolerant trigger may match one phase more on one side (centre +-1 sample)
        assert len(o) == len(outs[0])
        for a, b in zip(o, outs[0]):
            assert a["channel"] == b["channel"] and abs(int(a["position"]) - int(b["position"])) <= 1
            assert np.array_equal(a["word_raw"], b["word_raw"]) and np.array_equal(a["word_dec"], b["word_dec"]) and a["min"] == b["min"]


def test_tolerant_sync_on_the_wideband_seam(gpu, decim):
    """bit-domain correlator (behind the fused channelizer) with tolerance: same records as the two-kernel form, and as
    the exact correlator on undamaged bursts"""
    first, C, D = 96, 832, decim
    n = int(0.25 * sw.FS_WIDE) // 1536 * 1536    # a burst lasts 0.173 s
    bursts = [(first + 3, 120000), (first + 400, 90000), (first + 830, 200000)]
    x, truth = sw.make_wideband(n, bursts, seed=11)
    outs = []
    for k, unfused in ((0, False), (3, False), (3, True)):
        with capi.Recc(n_channels=C, sps=1536 // D, max_samples=n // D + 72, max_bursts=64, unfused_wideband=unfused, sync_tolerance=k,
                       wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}) as r:
            r.push_wideband(x)
            r.push_wideband(np.zeros(64 * D, np.complex64))
            outs.append(r.drain())
    assert len(outs[0]) == len(bursts)
    assert outs[1].tobytes() == outs[2].tobytes()
    assert [g["min"] for g in outs[1]] == [g["min"] for g in outs[0]]
    assert all(np.array_equal(a["word_raw"], b["word_raw"]) for a, b in zip(outs[0], outs[1]))


def test_tolerance_out_of_range_is_rejected(gpu):
    with pytest.raises(capi.AmpsError):
        capi.Recc(n_channels=1, sps=10, max_samples=4096, max_bursts=4, sync_tolerance=9)


@pytest.mark.parametrize("sps,lead_syms", [(10, 0), (10, 3), (10, 6), (4, 5), (3, 4)])
def test_tolerant_trigger_that_begins_in_front_of_the_stream(gpu, sps, lead_syms):
    """with sync_tolerance > 0 a trigger may begin BEFORE the first sample (its first symbols are the tolerated wrong
    ones); block 0 of the capture's timing tracking then looks in front of the stream, where the CPU model reads 1 (gbit()).  The
    device used to read stream bit 0 unshifted there.  The stream here starts `lead_syms` symbols INTO the trigger of its first burst."""
    rng = np.random.default_rng(77 + sps + lead_syms)
    kind, min10, esn, dialed, words = synth.random_message(rng)
    bits = synth.burst_bits(words, dcc=2, rng=rng)
    n = (len(bits) * 2 + 400) * sps
    iq = synth.fsk_modulate(n + 200 * sps, [(200 * sps, bits)], sps=sps, fs=20e3 * sps, snr_db=30.0, rng=rng)
    # the trigger's first symbol = burst symbol 8 (burst bits [4, 41)): cut the stream `lead_syms` symbols behind it
    cut = 200 * sps + (8 + lead_syms) * sps
    x = np.ascontiguousarray(iq[cut:cut + (n // 64) * 64])
    for k in (0, 8):
        with capi.Recc(n_channels=1, sps=sps, max_samples=x.size, max_bursts=8, sync_tolerance=k) as r:
            r.push_iq(x[None, :])
            got = r.drain()
        want = oracle.fused_push_all(x[None, :], sps=sps, tolerance=k)
        assert got.tobytes() == want.tobytes(), (k, len(got), len(want))
        if k == 8 and lead_syms in (3, 4, 5, 6):                 # found although its first symbols were never received, and decoded
            assert len(got) == 1 and got[0]["min"].decode() == min10
