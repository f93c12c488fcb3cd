"""TEST INFRASTRUCTURE: a wideband block with impaired mobiles, and the restated reference chain's verdict on it -- the same on every
box (with the 400 ksps cuts made by rocFFT on the GPU, whether the restated Mueller & Mueller loop locked on a burst
at a 2 kHz carrier offset changed from box to box, and the comparison rested on one to four bursts of sixteen).

The block is synthesised on the GPU with ELEMENTWISE float64 operations only (no scan, no FFT: a CPFSK phase is piecewise linear, so a
sample is its symbol's start phasor -- a 3456-point cumulative sum done in numpy -- times a table entry exp(j w k)), the noise by
torch's counter-based generator; the reference side runs on the CPU in float64 numpy (pocketfft).

The reference has no channelizer: it tunes a receiver 160 kHz below ONE channel, takes 400 ksps and runs freq_xlating_fir_filter_ccc
(299 taps) -> quadrature_demod_cf -> clock_recovery_mm_ff -> binary_slicer_fb -> recc -> recc_decode (grc/recctest.grc:889-937, 458,
846-874, 807); `cut400` is what that receiver would deliver for channel c of the block -- the band [f_c - 360 kHz, f_c + 40 kHz) by an
ideal FFT-domain extraction."""
import numpy as np

import oracle
from gr_amps_amd import synth, synth_wideband as sw

FS = sw.FS_WIDE


def symbol_starts(nsym, ppm):
    """first sample and length of every symbol of synth.symbol_waveform(sym, 1536, ppm): sample m belongs to symbol floor(m * rate)"""
    rate = (1.0 + ppm * 1e-6) / 1536.0
    nsamp = int(np.floor(nsym / rate))
    s = np.arange(nsym, dtype=np.float64)
    m = np.ceil(s / rate).astype(np.int64)
    for _ in range(3):                                                      # float64 fix-up: the first m with floor(m * rate) >= s
        m = np.where(np.floor((m - 1) * rate) >= s, m - 1, m)
        m = np.where(np.floor(m * rate) < s, m + 1, m)
    m[0] = 0
    length = np.diff(np.concatenate([m, [nsamp]]))
    assert (length > 0).all() and (np.floor(m * rate) == s).all() and (np.floor((m[1:] - 1) * rate) == s[:-1]).all()
    return m, length, nsamp


def make_block(torch, dev, n, chans, first, ppm, cfo, snr_db, seed):
    """one seizure burst in each of `chans` (band channel numbers; FFT bin = first + c), every mobile `ppm` off in its bit clock and
    `cfo` Hz off in its carrier (both signs alternate from channel to channel), white noise for `snr_db` of C/N in 30 kHz.
    Returns (complex64 [n] on the device, {c: (MIN, words)})."""
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sigma = 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0) * np.sqrt(FS / 30e3)
    x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g, dtype=torch.float32) * float(sigma))
    planted = {}
    for i, c in enumerate(chans):
        k = (first + c) % 1024
        sgn = 1.0 if i % 2 == 0 else -1.0
        _, min10, _, _, words = synth.random_message(rng)
        sym = synth.manchester(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)).astype(np.int64)
        fc = sw.bin_freq(k) + sgn * cfo
        start, length, nsamp = symbol_starts(len(sym), sgn * ppm)
        off = int(rng.integers(40000, n - nsamp - 40000))
        w = np.where(sym > 0, fc + 8e3, fc - 8e3) * (2 * np.pi / FS)        # rad per sample, per symbol
        phi = float(rng.uniform(0, 2 * np.pi)) + 2 * np.pi * fc / FS * off + np.concatenate([[0.0], np.cumsum(w * length)[:-1]])
        tab = np.exp(1j * np.outer(np.array([fc - 8e3, fc + 8e3]) * (2 * np.pi / FS), np.arange(int(length.max()) + 1)))
        ln = torch.from_numpy(length).to(dev)
        kk = torch.arange(nsamp, device=dev) - torch.repeat_interleave(torch.from_numpy(start).to(dev), ln)
        row = torch.repeat_interleave(torch.from_numpy(sym).to(dev), ln)
        burst = torch.repeat_interleave(torch.from_numpy(np.exp(1j * phi)).to(dev), ln) * torch.from_numpy(tab).to(dev)[row, kk]
        x[off:off + nsamp] += burst.to(torch.complex64)
        planted[c] = (min10, [list(wd) for wd in words])
    return x, planted


def cut400(X, n, c, first):
    """channel c of the block with spectrum X (numpy.fft.fft of the complex128 block) as the reference's receiver sees it"""
    nout = n * 5 // 384
    cbin = int(round((sw.bin_freq((first + c) % 1024) - 160e3) / FS * n))
    idx = (np.arange(-nout // 2, nout // 2) + cbin) % n
    return (np.fft.ifft(np.fft.ifftshift(X[idx])) * (nout / n)).astype(np.complex64)


def good(recs, min10, words):
    sent = [bytes(np.asarray(w, np.uint8)) for w in words]
    return any(g["min"].decode() == min10 and g["valid"][:len(sent)].all() and [bytes(g["word_dec"][w]) for w in range(len(sent))] == sent for g in recs)


def reference_verdicts(x, chans, first, planted):
    """x: the block as a host complex64 array.  {c: the restated chain decodes the planted burst of channel c from its own 400 ksps cut}"""
    n = x.size
    assert n % 384 == 0
    X = np.fft.fft(x.astype(np.complex128))
    return {c: good(oracle.chain_iq400(cut400(X, n, c, first), 160e3, chunk=4096), *planted[c]) for c in chans}
