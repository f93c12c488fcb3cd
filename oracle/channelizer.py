"""numpy (float64) model of the polyphase channelizer seam -- TEST INFRASTRUCTURE ONLY.

Weighted-overlap-add filter bank, the textbook form of what gr_amps_amd/csrc/recc_channelizer.hip.h
computes: frame m covers samples [ (m+1)D - L, (m+1)D ), is weighted by the prototype h, folded modulo M
with an absolute phase reference and transformed by an M-point DFT.  Samples before the stream are zero.
It replaces M instances of the reference's freq_xlating_fir_filter_ccc (grc/recctest.grc:889-937).
"""
import numpy as np


def cutoff_for_decim(D):
    """The prototype's -6 dB point: 13 kHz behind the 2x oversampled bank (D = 512, 60 ksps per channel), 15 kHz at D = 768 (40 ksps,
    two samples per Manchester symbol), where the slicer has no third sampling phase to spare for a carrier offset."""
    return {512: 13.0e3, 768: 15.0e3}[int(D)]


def design_taps(P, M=1024, cutoff_hz=13.0e3, chan_hz=30.0e3, beta=8.0):
    """Kaiser(beta) windowed sinc with unit DC gain: the prototype specified in DESIGN.md."""
    L = P * M
    fc = cutoff_hz / (M * chan_hz)
    i = np.arange(L, dtype=np.float64)
    m = i - 0.5 * (L - 1)
    h = 2.0 * fc * np.sinc(2.0 * fc * m) * np.kaiser(L, beta)
    return h / h.sum()


def channelize(x, P=8, M=1024, D=512, first_bin=0, n_channels=None, taps=None):
    """x: complex wideband stream from sample 0.  Returns complex128 [C][nframes], nframes = len(x) // D."""
    h = design_taps(P, M, cutoff_for_decim(D)) if taps is None else np.asarray(taps, np.float64)
    L = h.size
    x = np.asarray(x, np.complex128)
    nfr = x.size // D
    xp = np.concatenate([np.zeros(L - D, np.complex128), x[:nfr * D]])
    C = M if n_channels is None else n_channels
    out = np.empty((nfr, M), np.complex128)
    for m in range(nfr):
        seg = xp[m * D:m * D + L] * h
        n0 = (m + 1) * D - L
        u = np.roll(seg.reshape(-1, M).sum(0), n0 % M)
        out[m] = np.fft.fft(u)
    bins = (first_bin + np.arange(C)) % M
    return np.ascontiguousarray(out[:, bins].T)
