"""The oracle's restatements of the GNU Radio stages G1 / G2 (oracle/ref_chain.c) against THIRD-PARTY implementations of the same
published formulas -- scipy.signal / numpy, written by neither the reference's authors nor this repository's.  Not reference vectors (only
a GNU Radio build gives those: scripts/pin_with_reference.sh), but independent of the hand that wrote the oracle:

  firdes.low_pass(gain, fs, fc, width, WIN_BLACKMAN)   windowed-sinc design, ntaps = int(74 fs / (22 width)) made odd, DC gain = gain
                                                       == gain * scipy.signal.firwin(ntaps, fc, window="blackman", fs=fs)
  freq_xlating_fir_filter_ccc(decim, taps, fc, fs)     == mix by exp(-j 2 pi fc n / fs), FIR, keep every decim-th sample (the block's
                                                       documented equivalent: it filters with heterodyned taps and de-rotates the output)
  quadrature_demod_cf(gain)                            == gain * angle(x[n] conj(x[n-1])), with fast_atan2f's documented error
(grc/recctest.grc:115-155, 889-937, 458: the parameters the flow graph uses.)"""
import numpy as np
import pytest

import oracle
from gr_amps_amd import synth

signal = pytest.importorskip("scipy.signal")


@pytest.mark.parametrize("gain,fs,fc,width", [(3.0, 400e3, 10e3, 4.5e3), (1.0, 200e3, 10e3, 5e3), (2.0, 400e3, 25e3, 9e3)])
def test_firdes_low_pass_is_scipys_windowed_sinc(gain, fs, fc, width):
    t = oracle.firdes_low_pass(gain, fs, fc, width)
    ntaps = int(74.0 * fs / (22.0 * width))
    ntaps += 1 - (ntaps & 1)
    assert t.size == ntaps
    w = gain * signal.firwin(ntaps, fc, window="blackman", fs=fs)
    assert np.abs(t - w).max() < 5e-8 * gain
    assert abs(float(t.sum()) - gain) < 1e-5 * gain


def test_freq_xlating_fir_is_mix_filter_decimate():
    iq, _ = synth.make_channel_block(1 << 15, 1, seed=5, sps=20)
    n = np.arange(iq.size)
    x = (iq * np.exp(2j * np.pi * 0.4 * n)).astype(np.complex64)            # a channel at +160 kHz of 400 ksps (grc/recctest.grc:591)
    taps = oracle.firdes_low_pass(3.0, 400e3, 10e3, 4.5e3)
    y = oracle.freq_xlating_fir(x, taps, 160e3, 400e3, 2)
    mixed = x.astype(np.complex128) * np.exp(-2j * np.pi * (160e3 / 400e3) * n)
    want = signal.lfilter(taps.astype(np.float64), 1.0, mixed)[::2]
    m = min(y.size, want.size)
    assert m >= iq.size // 2 - 1
    assert np.abs(y[:m] - want[:m]).max() < 2e-6 * np.abs(want).max() + 1e-6   # float32 accumulation + the rotator renormalised every 512 outputs


def test_quadrature_demod_is_the_phase_step():
    iq, _ = synth.make_channel_block(1 << 15, 2, seed=6)
    d = oracle.quadrature_demod(iq)
    want = np.angle(iq[1:].astype(np.complex128) * np.conj(iq[:-1].astype(np.complex128)))
    assert np.abs(d[1:] - want).max() < 1e-5                                   # fast_atan2f: 255-entry table + linear interpolation


@pytest.mark.parametrize("D", [512, 768])
def test_filter_bank_model_is_mix_filter_decimate_per_channel(D):
    """The numpy model the GPU filter bank is held to (oracle/channelizer.py: weighted overlap-add + FFT, what chz12_kernel computes) IS, bin
    by bin, the per-channel chain it replaces with another prototype: mix bin k to DC, FIR with the prototype, keep every D-th sample
    -- freq_xlating_fir_filter_ccc's job (grc/recctest.grc:889-937) -- stated with scipy.signal.lfilter; and its prototype is scipy's
    Kaiser-windowed sinc (-6 dB at 13 kHz behind the D = 512 bank, at 15 kHz behind the D = 768 one).  With an absolute phase reference
    the identity does not care whether D divides M (at D = 768 the fold's branch rotation has period four frames)."""
    from oracle import channelizer as cz
    rng = np.random.default_rng(1)
    M, P = 1024, 8
    n = D * 40
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    Y = cz.channelize(x, P, M, D)
    h = cz.design_taps(P, M, cz.cutoff_for_decim(D))
    w = signal.firwin(h.size, cz.cutoff_for_decim(D), window=("kaiser", 8.0), fs=M * 30e3)
    assert np.abs(h - w / w.sum()).max() < 1e-15
    nn = np.arange(n)
    for k in (0, 5, 96, 511, 512, 927, 1023):
        mixed = x * np.exp(-2j * np.pi * k * nn / M)
        want = signal.lfilter(h[::-1], 1.0, mixed)[D - 1::D][:Y.shape[1]]       # frame m ends with sample (m + 1) D - 1
        assert np.abs(Y[k] - want).max() < 1e-10 * np.abs(want).max(), k          # float64 rounding of the mixer phase over 30 000 samples
