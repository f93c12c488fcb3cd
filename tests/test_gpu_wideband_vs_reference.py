"""-m gpu: the wideband seam against the restated reference chain, channel by channel (parity holes 2 and 3).

The reference has no channelizer: it tunes a receiver 160 kHz below ONE channel, takes 400 ksps and runs freq_xlating_fir_filter_ccc
(299 taps) -> quadrature_demod_cf -> clock_recovery_mm_ff -> binary_slicer_fb -> recc -> recc_decode on it
(grc/recctest.grc:889-937, 458, 846-874, 807).  Here every tested channel of a 30.72 Msps block is cut out exactly that way --
the band [f_k - 360 kHz, f_k + 40 kHz) by an ideal float64 FFT-domain extraction, which puts the channel at +160 kHz of a
400 ksps stream -- and pushed through oracle.chain_iq400; the same block goes through amps_recc_push_wideband.  Words are
compared wherever the reference chain decodes the burst (its Mueller & Mueller loop has to lock inside the four dotting bits
the precursor has to spare: it misses a few per cent of the bursts at any SNR, scripts/slicer_sensitivity.py)."""
import numpy as np
import pytest

import oracle
from gr_amps_amd import capi, synth, synth_wideband as sw

pytestmark = pytest.mark.gpu
FS, FIRST, CW = sw.FS_WIDE, 96, 832
BLEN = 3456 * 1536                                   # samples of one seizure burst at 30.72 Msps


def _block(torch, dev, n, bursts, snr_db, seed):
    """bursts: [(channel, offset, level_dB)] -> (complex64 [n] on the device, {index: (MIN, words)}); the noise gives 0 dB-level
    bursts `snr_db` of C/N in 30 kHz"""
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sigma = 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0) * np.sqrt(FS / 30e3)
    x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g, dtype=torch.float32) * float(sigma))
    truth = {}
    for i, (c, off, lvl) in enumerate(bursts):
        k = (FIRST + c) % 1024
        _, min10, _, _, words = synth.random_message(rng)
        sym = synth.manchester(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)).astype(np.float32) * 2 - 1
        f = torch.from_numpy(sym).to(dev).repeat_interleave(1536) * (2 * np.pi * 8e3 / FS)
        fc = 2 * np.pi * sw.bin_freq(k) / FS
        ph = torch.cumsum(f.double() + fc, 0) + float(rng.uniform(0, 2 * np.pi)) + fc * off
        x[off:off + BLEN] += float(10.0 ** (lvl / 20.0)) * torch.polar(torch.ones_like(ph, dtype=torch.float32), ph.remainder(2 * np.pi).float())
        truth[i] = (min10, [list(w) for w in words])
    return x, truth


def _cut400(torch, X, n, c):
    """channel c of the block with spectrum X as the reference's receiver sees it: 400 ksps, the channel at +160 kHz"""
    nout = n * 5 // 384
    cbin = int(round((sw.bin_freq((FIRST + c) % 1024) - 160e3) / FS * n))
    idx = (torch.arange(-nout // 2, nout // 2, device=X.device) + cbin) % n
    return (torch.fft.ifft(torch.fft.ifftshift(X[idx])) * (nout / n)).to(torch.complex64).cpu().numpy()


def _gpu_records(x, n, D):
    with capi.Recc(n_channels=CW, sps=1536 // D, max_samples=n // D + 72, max_bursts=1024,
                   wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": FIRST}) as r:
        r.push_wideband(x)
        import torch
        r.push_wideband(torch.zeros(64 * D, dtype=torch.complex64, device=x.device))
        recs = r.drain()
    by = {}
    for g in recs:
        by.setdefault(int(g["channel"]), []).append(g)
    return by


def _good(recs, min10, words):
    sent = [bytes(np.asarray(w, np.uint8)) for w in words]
    return any(g["min"].decode() == min10 and g["valid"][:len(sent)].all() and [bytes(g["word_dec"][w]) for w in range(len(sent))] == sent for g in recs)


def test_wideband_records_equal_the_reference_chain_channel_by_channel(gpu, decim):
    """twelve channels across the band (both edges, the centre, neighbours of the DC bin), one burst each at 30 dB: every burst
    the restated flow graph decodes from its own 400 ksps cut comes out of the wideband seam with the same fields and words"""
    import torch
    D = decim
    n = int(0.45 * FS) // 1536 * 1536
    assert n % 384 == 0                                               # the 400 ksps cut (x 5 / 384) has a whole number of samples
    rng = np.random.default_rng(5)
    chans = [0, 1, 40, 200, 415, 416, 417, 600, 700, 829, 830, 831]
    bursts = [(c, int(rng.integers(40000, n - BLEN - 40000)), 0.0) for c in chans]
    x, truth = _block(torch, gpu, n, bursts, 30.0, seed=41)
    got = _gpu_records(x, n, D)
    X = torch.fft.fft(x.to(torch.complex128))
    nref = 0
    for i, (c, off, _) in enumerate(bursts):
        min10, words = truth[i]
        assert _good(got.get(c, []), min10, words), ("wideband seam lost channel", c)
        assert len(got[c]) == 1
        ref = oracle.chain_iq400(_cut400(torch, X, n, c), 160e3, chunk=4096)
        if not len(ref):
            continue                                                  # the reference's loop did not lock on this burst
        nref += 1
        a, b = ref[0], got[c][0]
        for f in ("dcc", "valid", "first_valid_rep", "word_raw", "word_dec", "a_MIN1", "b_MIN2", "msg_class", "min", "dialed", "esn", "n_called_words"):
            assert np.array_equal(a[f], b[f]), (c, f)
    assert nref >= 9, nref                                            # measured: 12 of 12 here; ~98 % of the bursts at 20 dB and above


@pytest.mark.parametrize("spacing,levels,must", [(1, (0.0, 10.0, 20.0), 2), (2, (20.0, 30.0, 40.0), 2)])
def test_overlapping_neighbour_bursts(gpu, spacing, levels, must, decim):
    """A weak burst (20 dB C/N) with a time-overlapping burst in the adjacent (30 kHz) or alternate (60 kHz) channel at +0 ... +40 dB:
    the reference's selectivity is its 299-tap channel filter (grc/recctest.grc:115-155), the wideband seam's is the prototype
    of the filter bank (Kaiser beta 8, 8 taps per branch, -6 dB at 13 kHz at D = 512 and at 15 kHz at D = 768).  The seam must decode the weak burst wherever the
    restated reference chain does, and in any case at the first `must` levels (adjacent +0 / +10 dB, alternate +20 / +30 dB)."""
    import torch
    D = decim
    n = int(0.45 * FS) // 1536 * 1536
    rng = np.random.default_rng(100 + spacing)
    bursts, weak = [], []
    for j, lvl in enumerate(levels):
        for rep in range(3):                                          # three weak channels per level, neighbour above or below
            c = 60 + 90 * j + 25 * rep
            off = int(rng.integers(40000, n - BLEN - 700000))
            side = 1 if rep % 2 == 0 else -1
            weak.append((len(bursts), c, lvl))
            bursts.append((c, off, 0.0))
            bursts.append((c + side * spacing, off + int(rng.integers(-600000, 600000)), lvl))   # overlaps >= 88 % of the weak burst
    x, truth = _block(torch, gpu, n, bursts, 20.0, seed=200 + spacing)
    got = _gpu_records(x, n, D)
    X = torch.fft.fft(x.to(torch.complex128))
    table = {}
    for i, c, lvl in weak:
        min10, words = truth[i]
        g_ok = _good(got.get(c, []), min10, words)
        r_ok = _good(oracle.chain_iq400(_cut400(torch, X, n, c), 160e3, chunk=4096), min10, words)
        table.setdefault(lvl, []).append((g_ok, r_ok))
        assert g_ok or not r_ok, ("the reference chain decodes a burst the wideband seam loses", spacing, lvl, c)
    for lvl in levels[:must]:
        assert all(g for g, _ in table[lvl]), (spacing, lvl, table[lvl])
    # the strong neighbours themselves always decode
    for i, (c, off, lvl) in enumerate(bursts):
        if lvl > 0.0:
            assert _good(got.get(c, []), *truth[i]), ("strong burst lost", c, lvl)
