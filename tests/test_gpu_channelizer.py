"""-m gpu: the polyphase channelizer seam (configs 2-3): wideband fc32 @ 30.72 Msps -> 1024-branch filter
bank -> fused RECC path at 3 samples/symbol (D = 512) or 2 (D = 768), checked against the numpy filter-bank model (float
tolerance), against the CPU model of the fused seam (bit-exact on the channelizer's own output) and against the
transmitted words.  Every test runs at both decimations (conftest.py: `decim`)."""
import numpy as np
import pytest

import oracle
from oracle import channelizer as cz
from gr_amps_amd import capi, synth_wideband as sw
from conftest import wb_cfg

pytestmark = pytest.mark.gpu


def _handle(D, C, first, max_frames, P=8, max_bursts=256, **kw):
    wb, sps = wb_cfg(D, first, P)
    return capi.Recc(n_channels=C, sps=sps, max_samples=max_frames, max_bursts=max_bursts, wideband=wb, **kw)


def test_channelizer_matches_numpy_filter_bank(gpu, decim, P=8):
    D = decim
    rng = np.random.default_rng(1)
    n = 200 * D
    t = np.arange(n)
    x = 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    for k, a in ((3, 1.0), (100, 0.5), (511, 0.7), (900, 0.3)):          # tones at +5 kHz offset in four channels
        x += a * np.exp(2j * np.pi * (sw.bin_freq(k) + 5e3) * t / sw.FS_WIDE)
    x = x.astype(np.complex64)
    with _handle(D, 1024, 0, n // D + 8, P) as r:
        got = r.debug_channelize(x)
    want = cz.channelize(x, P=P, D=D)
    assert got.shape == want.shape == (1024, n // D)
    scale = np.abs(want).max()
    err = np.abs(got - want).max() / scale
    assert err < 2e-5, err
    # the tone in channel 100 comes out at +5 kHz with continuous phase (60 ksps at D = 512, 40 ksps at D = 768)
    y = got[100, 40:]
    ph = np.angle(y[1:] * np.conj(y[:-1]))
    rate = sw.FS_WIDE / D
    assert np.allclose(ph, 2 * np.pi * 5e3 / rate, atol=3e-2) and abs(ph.mean() - 2 * np.pi * 5e3 / rate) < 2e-3


def test_channelizer_streaming_equals_one_shot(gpu, decim):
    D = decim
    rng = np.random.default_rng(2)
    n = 96 * D + 77
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    with _handle(D, 832, 96, 200) as r:
        one = r.debug_channelize(x)
    with _handle(D, 832, 96, 200) as r:
        parts, off = [], 0
        for m in (1, 511, 512, 513, 5000, 12345, n):
            m = min(m, n - off)
            if m <= 0:
                break
            parts.append(r.debug_channelize(x[off:off + m]))
            off += m
        many = np.concatenate(parts, axis=1)
    assert one.shape == (832, n // D) and many.shape == one.shape
    assert np.array_equal(one.view(np.uint32), many.view(np.uint32))     # frame arithmetic does not depend on the chunking


def test_channelizer_carry_written_by_the_kernel_equals_the_copy_kernel(gpu, decim):
    """Launches of 64 workgroups and more write the next launch's carry themselves (a slice per workgroup); shorter ones leave it to
    chz_carry_kernel.  A stream pushed in pieces that alternate between the two forms -- with ragged leftovers in the carry --
    comes out bit for bit as in one push (which is itself a long launch)."""
    D = decim
    rng = np.random.default_rng(12)
    pieces = [4160 * D + 13, 70 * D - 13, 4223 * D + 501, 3 * D + 11, 4100 * D - 1]   # 65 / 2 / 66 / 1 / 65 workgroups
    n = sum(pieces)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    with _handle(D, 64, 96, 4300) as r:
        one = r.debug_channelize(x[:4300 * D])                 # reference for the first stretch ...
    with _handle(D, 64, 96, n // D + 8) as r:
        whole = r.debug_channelize(x)                          # ... and for everything (194 workgroups)
    assert np.array_equal(one.view(np.uint32), whole[:, :one.shape[1]].view(np.uint32))
    with _handle(D, 64, 96, 4300) as r:
        parts, off = [], 0
        for m in pieces:
            parts.append(r.debug_channelize(x[off:off + m]))
            off += m
        many = np.concatenate(parts, axis=1)
    assert many.shape == whole.shape == (64, n // D)
    assert np.array_equal(many.view(np.uint32), whole.view(np.uint32))


def test_wideband_bursts_decode_to_the_transmitted_words(gpu, decim):
    D = decim
    first, C = 96, 832
    n = int(0.2 * sw.FS_WIDE) // D * D
    bursts = [(first + 4, 200000), (first + 5, 250000), (first + 6, 300000), (first + 700, 100000), (first + 831, 400000), (first + 0, 50000)]
    x, truth = sw.make_wideband(n, bursts, seed=3)
    with _handle(D, C, first, n // D + 8) as r:
        chan = r.debug_channelize(x)
    with _handle(D, C, first, n // D + 8) as r:
        half = (n // 2) // D * D + 100                                   # two ragged pushes
        r.push_wideband(x[:half])
        r.push_wideband(x[half:])
        got = r.drain()
    assert len(got) == len(bursts)
    by_chan = {int(g["channel"]): g for g in got}
    for (k, off), (kind, min10, esn, dialed, words) in truth.items():
        g = by_chan[k - first]
        assert g["min"].decode() == min10 and capi.MSG_CLASSES[g["msg_class"]] == kind and g["valid"].all()
        for w, bits in enumerate(words):
            assert list(g["word_raw"][w][:36]) == list(bits)
    # the RECC kernels are bit-exact against the CPU model when both see the channelizer's own output
    active = sorted(by_chan)
    want = oracle.fused_push_all(chan[active], sps=1536 // D)
    want["channel"] = np.array(active, np.uint32)[want["channel"]]
    assert got.tobytes() == want.tobytes()
    # and the float64 filter-bank model leads to the same words
    ref_chan = cz.channelize(x, P=8, D=D, first_bin=first, n_channels=C)[active].astype(np.complex64)
    ref = oracle.fused_push_all(ref_chan, sps=1536 // D)
    assert [r_["min"] for r_ in ref] == [g["min"] for g in got]
    assert all(np.array_equal(a["word_raw"], b["word_raw"]) for a, b in zip(ref, got))


def test_fused_and_two_kernel_wideband_forms_agree(gpu, decim):
    """amps_recc_push_wideband fuses discriminator + boxcar + slicer behind the FFT (only slicer bits reach HBM);
    AMPS_RECC_FLAG_UNFUSED_WIDEBAND keeps the channel-major intermediate.  Same arithmetic -> same records, for
    one-shot and ragged pushes."""
    D = decim
    first, C = 96, 832
    n = int(0.25 * sw.FS_WIDE) // D * D
    bursts = [(first + 10 * i + (i % 3), 60000 + 211111 * i) for i in range(8)]
    x, truth = sw.make_wideband(n, bursts, seed=5)
    outs = []
    for unfused, chunks in ((False, [n]), (True, [n]), (False, [100000, 1, 4000000, n]), (True, [777777, n])):
        with _handle(D, C, first, n // D + 72, max_bursts=64, unfused_wideband=unfused) as r:
            off, recs = 0, []
            for m in chunks:
                m = min(m, n - off)
                if m <= 0:
                    break
                r.push_wideband(x[off:off + m])
                recs.append(r.drain())
                off += m
            # flush: the fused form holds back up to 63 frames, the two-kernel form up to 1; push silence
            r.push_wideband(np.zeros(64 * D, np.complex64))
            recs.append(r.drain())
            outs.append(np.concatenate(recs))
    assert len(outs[0]) == len(bursts)
    for o in outs[1:]:
        assert o.tobytes() == outs[0].tobytes()
    mins = sorted(v[1] for v in truth.values())
    assert sorted(g["min"].decode() for g in outs[0]) == mins


def test_config2_64_channels_behind_the_channelizer(gpu, decim):
    """BASELINE configs[2]: 64 RECC channels (a sub-band of the 1024 bins) behind the polyphase channelizer.  Bursts in
    the first, last and interior channels of the group and in channels just outside it (which must not be reported)."""
    D = decim
    first, C = 200, 64
    n = int(0.25 * sw.FS_WIDE) // D * D
    inside = [(first + 0, 90000), (first + 63, 140000), (first + 17, 60000), (first + 18, 200000), (first + 40, 30000)]
    outside = [(first - 1, 100000), (first + 64, 120000)]
    x, truth = sw.make_wideband(n, inside + outside, seed=21)
    with _handle(D, C, first, n // D + 72, max_bursts=64) as r:
        chan = r.debug_channelize(x)
    assert chan.shape[0] == C
    with _handle(D, C, first, n // D + 72, max_bursts=64) as r:
        for part in np.array_split(x, 3):
            r.push_wideband(part)
        r.push_wideband(np.zeros(64 * D, np.complex64))
        got = r.drain()
    assert sorted(int(g["channel"]) for g in got) == sorted(k - first for k, _ in inside)
    by_chan = {int(g["channel"]): g for g in got}
    for (k, off), (kind, min10, esn, dialed, words) in truth.items():
        if not (first <= k < first + C):
            continue
        g = by_chan[k - first]
        assert g["min"].decode() == min10 and g["valid"].all()
        for w, bits in enumerate(words):
            assert list(g["word_raw"][w][:36]) == list(bits)
    want = oracle.fused_push_all(chan, sps=1536 // D)
    assert [(int(w["channel"]), w["min"]) for w in want] == [(int(g["channel"]), g["min"]) for g in got]
    assert all(np.array_equal(a["word_raw"], b["word_raw"]) and np.array_equal(a["word_dec"], b["word_dec"]) for a, b in zip(want, got))


def test_wideband_stream_origin(gpu, decim):
    """the same on the wideband seam (origin counts channel samples, i.e. channelizer frames)"""
    D = decim
    first, C = 96, 832
    n = int(0.25 * sw.FS_WIDE) // D * D
    bursts = [(first + 7, 100000), (first + 500, 150000)]
    x, truth = sw.make_wideband(n, bursts, seed=31)
    outs = []
    for origin in (0, (1 << 42) + 64 * 999):
        with _handle(D, C, first, n // D + 72) as r:
            if origin:
                r.set_origin(origin)
            r.push_wideband(x[: n // 3])
            r.push_wideband(x[n // 3:])
            r.push_wideband(np.zeros(64 * D, np.complex64))
            g = r.drain()
        g["position"] -= np.uint64(origin)
        outs.append(g)
    assert len(outs[0]) == len(bursts) and outs[0].tobytes() == outs[1].tobytes()
