"""Synthetic wideband AMPS band: many 30 kHz RECC channels in one complex stream at fs = M * 30 kHz
(SURVEY.md 8d configs 2-3).  Used by tests and bench.py; numpy for small cases, torch on the GPU for
bench-sized blocks."""
import numpy as np

from . import synth

FS_WIDE = 30.72e6
M = 1024


def bin_freq(k, fs=FS_WIDE, m=M):
    return k * fs / m if k < m // 2 else (k - m) * fs / m


def make_wideband(nsamp, bursts, seed, snr_db=30.0, fs=FS_WIDE, noise_bw=60e3, sym_ppm=0.0, cfo_hz=0.0):
    """bursts: list of (fft_bin, sample_offset).  Every burst is a random message.  AWGN is scaled so the
    SNR is `snr_db` inside one channel's 60 kHz output bandwidth.  Returns (complex64 [nsamp], truth dict
    keyed by (bin, offset)).  sym_ppm / cfo_hz: symbol-clock and carrier offset of every mobile (synth.fsk_modulate)."""
    rng = np.random.default_rng(seed)
    sps_w = int(round(fs / 20e3))
    sigma = 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0) * np.sqrt(fs / noise_bw)
    x = (rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)) * sigma
    n = np.arange(nsamp)
    truth = {}
    for k, off in bursts:
        kind, min10, esn, dialed, words = synth.random_message(rng)
        bits = synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)
        sym = synth.manchester(bits).astype(np.float64) * 2.0 - 1.0
        f = synth.symbol_waveform(sym, sps_w, sym_ppm) * 8e3 + cfo_hz
        m = min(f.size, nsamp - off)
        if m <= 0:
            continue
        ph = 2.0 * np.pi * np.cumsum(f[:m]) / fs + rng.uniform(0, 2 * np.pi)
        x[off:off + m] += np.exp(1j * (ph + 2.0 * np.pi * bin_freq(k, fs) * n[off:off + m] / fs))
        truth[(k, off)] = (kind, min10, esn, dialed, words)
    return x.astype(np.complex64), truth
