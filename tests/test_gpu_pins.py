"""-m gpu pins: the device BCH decoder against an independent brute-force reference on EVERY syndrome
and every weight-3 pattern, and every constant the reference embeds in its sources pushed through the GPU seams."""
import itertools
import json
import os

import numpy as np
import pytest

import bchref
import oracle
from gr_amps_amd import capi, synth, synth_wideband as sw

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "survey_kats.json")))


def _words(ints):
    return np.array([bchref.bits(w) for w in ints], np.uint8)


def test_gpu_bch63_equals_brute_force_on_all_syndromes_and_weight3_patterns(gpu):
    lead = bchref.coset_leaders()
    rx = list(range(4096)) + [(1 << i) | (1 << j) | (1 << k) for i, j, k in itertools.combinations(range(63), 3)]
    with capi.Recc(n_channels=1, max_bursts=4) as r:
        msg, valid, nerr = r.bch_decode(_words(rx))              # k = 51: the full-length (63,51) code, no shortening positions
    n_ok = n_quirk = n_fail = 0
    for w, m, v, ne in zip(rx, msg, valid, nerr):
        rem = bchref.polymod(w)
        if rem in lead:
            wt, e = lead[rem]
            assert v == 1 and ne == wt and bchref.from_bits(list(m) + [0] * 12) == ((w ^ e) >> 12) << 12
            n_ok += 1
        elif bchref.evaluate(w, 1) == 0 and bchref.is_cube(bchref.evaluate(w, 3)):
            assert v == 1 and ne == 3                            # IT++'s Lambda = 1 + S3 x^3 case: three roots -> accepted
            ok, out, nf = oracle.bch63_decode(np.array(bchref.bits(w), np.uint8))
            assert ok and nf == 3 and list(out[:51]) == list(m)
            n_quirk += 1
        else:
            assert v == 0
            n_fail += 1
    assert n_ok >= 2017 and n_quirk >= 21 and n_fail >= 2058 and n_ok + n_quirk + n_fail == len(rx)


def test_gpu_encoder_on_the_reference_in_source_words(gpu):
    c = KATS["in_source_constants"]
    words = [c["control_filler_word"], c["overhead_word_1"]]       # lib/focc_impl.cc:294, apps/testalloc.cc:39
    msg = np.array([[int(b) for b in w] for w in words], np.uint8)
    with capi.Recc(n_channels=1, max_bursts=4) as r:
        cw = r.bch_encode(msg)                                      # (40,28): focc_impl::focc_bch, lib/focc_impl.cc:156-176
        dec, valid, nerr = r.bch_decode(cw)
        # one and two channel errors are corrected; a correction that lands in the 23 shortening positions is refused
        bad = cw.copy(); bad[0, 3] ^= 1; bad[1, 7] ^= 1; bad[1, 30] ^= 1
        dec2, valid2, nerr2 = r.bch_decode(bad)
    for i, w in enumerate(words):
        m = int(w, 2) << 12
        assert bchref.from_bits([0] * 23 + list(cw[i])) == m | bchref.polymod(m)      # systematic remainder of x^12 m(x) mod g(x)
        assert list(cw[i][:28]) == [int(b) for b in w]
    assert "".join(map(str, cw[0][28:])) == c["control_filler_parity_not_reference_pinned"]
    assert valid.all() and (nerr == 0).all() and np.array_equal(dec, msg)
    assert valid2.all() and list(nerr2) == [1, 2] and np.array_equal(dec2, msg)


def test_in_source_trigger_through_the_three_seams(gpu):
    """lib/recc_impl.cc:76's bit string, Manchester coded by the rule of :54-59, is found by the symbol seam at the symbol it
    ends on, and a burst built around it is decoded by the IQ seam and by the wideband seam"""
    bits = [int(b) for b in KATS["trigger_bits"]]
    trig = np.array([s for b in bits for s in ((0, 1) if b else (1, 0))], np.uint8)
    rng = np.random.default_rng(5)
    kind, min10, esn, dialed, words = synth.random_message(rng)
    full = synth.burst_bits(words, dcc=1, rng=rng)
    assert list(full[4:41]) == bits                                  # the generator's preamble = 4 more dotting bits + the in-source string
    payload = synth.manchester(full[41:])
    assert payload.size == 3374
    idle = (rng.integers(0, 2, 5000) & 1).astype(np.uint8)
    idle[1::2] = idle[0::2]                                          # pairs (0,0)/(1,1) only: the trigger cannot occur in the idle stream
    stream = np.concatenate([idle, trig, payload, np.zeros(10, np.uint8)])
    with capi.Recc(n_channels=1, max_bursts=4) as r:
        b, ch = r.push_symbols(stream[None, :])
        assert len(b) == 1 and np.array_equal(b[0], payload)
        rec = r.decode_bursts(b)
    assert rec[0]["min"].decode() == min10 and rec[0]["valid"][0] and capi.MSG_CLASSES[rec[0]["msg_class"]] == kind
    # IQ seam: the same burst as CPFSK at 10 samples per symbol
    iq = synth.fsk_modulate(60000, [(7000, full)], sps=10, fs=200e3, snr_db=30.0, rng=rng)
    with capi.Recc(n_channels=1, sps=10, max_samples=65536, max_bursts=4) as r:
        r.push_iq(iq[None, :])
        got = r.drain()
    assert len(got) == 1 and got[0]["min"].decode() == min10 and np.array_equal(got[0]["word_raw"], rec[0]["word_raw"])
    assert int(got[0]["position"]) == 7000 + (4 + 37) * 2 * 10 - 1   # decision instant of the trigger's last symbol
    # wideband seam: one channel of the band
    n = int(0.3 * sw.FS_WIDE) // 512 * 512
    x, truth = sw.make_wideband(n, [(96 + 100, 150000)], seed=9)
    with capi.Recc(n_channels=832, sps=3, max_samples=n // 512 + 72, max_bursts=8,
                   wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 96}) as r:
        r.push_wideband(x)
        r.push_wideband(np.zeros(64 * 512, np.complex64))
        w = r.drain()
    (k, off), (kind2, min2, _, _, _) = list(truth.items())[0]
    assert len(w) == 1 and int(w[0]["channel"]) == 100 and w[0]["min"].decode() == min2 and w[0]["valid"].all()
