"""not gpu: the CPU model of the fused seams (oracle/fused_model.c: trigger runs, hold-off, the wait for a burst's tail, the capture with
its timing tracking) against a SECOND statement of the same rules written from the prose (tests/trackref.py, vectorised numpy with
rational arithmetic for the tracking decision), and -- through tests/refdecode.py, the second restatement of bursts_message -- every
record field of every captured burst.  The slicer bits come from the model itself (all four specs; the slicer has its own pins:
tests/test_cpu_exact_slicer.py, test_cpu_oracle.py), the inputs from tests/fuzzlib.py: random sample rates, SNR 8-30 dB, truncated
bursts, damaged preambles, symbol-clock offsets to 1500 ppm, carrier offsets to 3 kHz, tolerant sync, tracked and fixed timing, and
random push schedules (the model streams; the second statement sees the finished bit stream)."""
import numpy as np

import fuzzlib
import oracle
import refdecode
import trackref
from test_second_restatement import _check


def test_trigger_is_the_last_37_preamble_bits():
    assert trackref.trigger_symbols().tolist() == list(oracle.trigger())


def _run(case, seed, sps=None):
    rng, info, iq = fuzzlib.build_case(case, seed, sps)
    sps, tol, track = info["sps"], info["tol"], not info["fixed"]
    nb = 0
    for c in range(iq.shape[0]):
        f = oracle.Fused(c, sps, tol, False, info["slicer"], track)
        recs, off, N = [], 0, iq.shape[1]
        while off < N:
            b = int(min(N - off, rng.integers(1, max(2, N // 2))))
            recs.append(f.push(iq[c, off:off + b], cap=64))
            off += b
        recs = np.concatenate(recs)
        g = f.taps()[2]
        assert len(g) == (N // 64) * 64
        caps = trackref.captures(g, sps, tol, track)
        assert [int(r["position"]) for r in recs] == [nc for nc, _ in caps], (case, c, info)
        for r, (nc, sym) in zip(recs, caps):
            assert int(r["channel"]) == c
            _check(r, refdecode.decode(sym), (case, c, nc))
        nb += len(caps)
    return nb, info


def test_model_equals_the_second_statement_on_random_streams():
    total, seen = 0, set()
    for case in range(160):
        nb, info = _run(case, 4242)
        total += nb
        seen.add((info["slicer"], info["fixed"], info["tol"] > 0, info["ppm"] != 0, info["cfo"] != 0))
    assert total >= 200                                            # bursts captured and compared field by field
    assert {s for s, *_ in seen} == {0, 1, 2, 3}                   # every slicer spec's bits went through both statements
    assert any(fx for _, fx, *_ in seen) and any(not fx for _, fx, *_ in seen)
    assert any(t for *_, t, _, _ in seen) and any(p for *_, p, _ in seen) and any(cf for *_, cf in seen)


def test_two_samples_per_symbol_rule_in_both_statements():
    """the wideband seam at D = 768 delivers two samples per symbol, where a block picks its own delay by its Manchester violations
    (DESIGN.md 4.4b): the same random streams, forced to that rate, through both statements"""
    total, moved = 0, 0
    for case in range(60):
        nb, info = _run(case, 777, sps=2)
        total += nb
    assert total >= 70


def test_two_samples_per_symbol_follows_a_clock_offset():
    """a burst 500 ppm slow slides 3.5 samples through a capture at two samples per symbol.  The stream is made the way the wideband
    seam sees it -- modulated at 480 ksps, through the filter bank's prototype (decimated to that rate), every twelfth sample -- so the
    slide is smooth, not one whole sample at a time.  Tracked, every transmitted word arrives; at one fixed phase the late words do
    not; and both statements agree on every symbol."""
    from gr_amps_amd import synth
    from oracle import channelizer as cz
    rng = np.random.default_rng(11)
    _, min10, _, _, words = synth.random_message(rng)
    bits = synth.burst_bits(words, dcc=1, rng=rng)
    x = synth.fsk_modulate(24 * (1000 + 2 * len(bits) + 1500), [(24 * 700 + 5, bits)], sps=24, fs=480e3, snr_db=14, rng=rng,
                           dtype=np.complex128, sym_ppm=-500.0)
    h = cz.design_taps(8, cutoff_hz=cz.cutoff_for_decim(768))[::64] * 64.0
    iq = np.convolve(x, h)[:x.size][3::12].astype(np.complex64)
    f = oracle.Fused(0, 2)
    recs = f.push(iq, cap=8)
    g = f.taps()[2]
    (nc, sym), = trackref.captures(g, 2)
    assert len(recs) == 1 and int(recs[0]["position"]) == nc and recs[0]["valid"].all()
    want = refdecode.decode(sym)
    _check(recs[0], want, 0)
    assert want["min"] == min10
    sent = [list(w) for w in words]
    assert [list(recs[0]["word_dec"][w]) for w in range(len(sent))] == sent
    fixed = oracle.decode_bursts(trackref.capture(g, nc, 2, track=False))[0]
    nw = len(sent)
    assert [list(fixed["word_dec"][w]) for w in range(nw)] != sent or not fixed["valid"][:nw].all()


def test_tracking_moves_and_both_statements_move_alike():
    """a burst 700 ppm fast at 3 samples per symbol slides 7 samples (2.4 symbols) through the capture: the two statements must agree
    on every symbol, the tracked capture carries the transmitted words to the last one, and the same instants without tracking do not
    (the BCH verdict is no witness: a stream that slipped a whole bit is a cyclic shift of code words, valid and wrong)"""
    from gr_amps_amd import synth
    rng = np.random.default_rng(5)
    _, min10, _, _, words = synth.random_message(rng)
    bits = synth.burst_bits(words, dcc=2, rng=rng)
    iq = synth.fsk_modulate(3000 + len(bits) * 6 + 4000, [(1000, bits)], sps=3, fs=60e3, snr_db=25, rng=rng, sym_ppm=700.0)
    f = oracle.Fused(0, 3)
    recs = f.push(iq, cap=8)
    g = f.taps()[2]
    (nc, sym), = trackref.captures(g, 3)
    assert len(recs) == 1 and int(recs[0]["position"]) == nc and recs[0]["valid"].all()
    want = refdecode.decode(sym)
    _check(recs[0], want, 0)
    assert want["min"] == min10 and want["manch_bad"][-1] <= 2
    sent = [int(b) for b in bits[-7 * 240:]]                       # the seven words x five repeats as transmitted
    last = slice(14 + 480 * 6, 14 + 480 * 7)
    assert refdecode.manchester(sym[last], 240)[0] == sent[-240:]
    fixed = trackref.capture(g, nc, 3, track=False)
    assert (fixed != sym).sum() > 200 and refdecode.manchester(fixed[last], 240)[0] != sent[-240:]
