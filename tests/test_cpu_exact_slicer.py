"""Slicer spec D ("exact-sign discriminator", include/amps_recc_numerics.h) of the CPU model against an independent float64
statement of what it claims to compute: the sign of the boxcar sum of libm atan2 phase steps.  No GPU.

The claim: g[n] = (sum_{k=n-sps+1..n} arg(x[k] conj(x[k-1])) >= 0), evaluated from four multiply-adds and sign bits.  In exact
arithmetic that is an identity; in binary32 the two conj-products round, so the bits may differ where the sum is within rounding
of zero (or of a multiple of 2 pi for the winding number) -- the test allows differences only at |S| <= 1e-4 rad."""
import numpy as np
import pytest

import oracle

TOL_RAD = 1.0e-4


def _fsk(n, sps, snr_db, seed, carrier=True):
    rng = np.random.default_rng(seed)
    sym = rng.integers(0, 2, n // sps + 1) * 2 - 1
    f = np.repeat(sym, sps)[:n] * 8e3
    ph = 2 * np.pi * np.cumsum(f) / (20e3 * sps)
    sig = np.exp(1j * ph) if carrier else np.zeros(n, complex)
    sigma = 10 ** (-snr_db / 20) / np.sqrt(2)
    return (sig + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


def _boxcar_of_libm_atan2(x, sps):
    """float64: S[n] = sum of the last sps principal-value phase steps (samples before the stream are zero -> step 0)"""
    x = x.astype(np.complex128)
    t = x * np.conj(np.concatenate([[0], x[:-1]]))
    d = np.where(t == 0, 0.0, np.angle(t))
    cs = np.concatenate([[0.0], np.cumsum(d)])
    idx = np.arange(len(x))
    return cs[idx + 1] - cs[np.maximum(idx - sps + 1, 0)]


@pytest.mark.parametrize("sps", [3, 4, 5, 6, 8, 10, 12])
@pytest.mark.parametrize("snr,carrier", [(30.0, True), (10.0, True), (0.0, True), (0.0, False)])
def test_exact_slicer_is_the_sign_of_the_libm_boxcar(sps, snr, carrier):
    n = 64 * 1500
    x = _fsk(n, sps, snr, seed=100 * sps + int(snr), carrier=carrier)
    f = oracle.Fused(0, sps, slicer=3)
    f.push(x)
    _, _, g = f.taps()
    assert len(g) == n
    S = _boxcar_of_libm_atan2(x, sps)
    want = (S >= 0).astype(np.uint8)
    assert g[:sps].all()                                   # no partner yet: ones by definition
    diff = np.nonzero(g[sps:] != want[sps:])[0] + sps
    assert len(diff) <= 2 and (np.abs(S[diff]) <= TOL_RAD).all(), (len(diff), np.abs(S[diff]).max() if len(diff) else 0)
    # the winding number is exercised: in noise the sum leaves (-pi, pi] often, where spec B (the telescoped form) is wrong
    if snr <= 0.0:
        assert (np.abs(S) > np.pi).mean() > 0.02
        fb = oracle.Fused(0, sps, slicer=1)
        fb.push(x)
        assert (fb.taps()[2][sps:] != want[sps:]).mean() > 0.01


@pytest.mark.parametrize("sps", [3, 10])
def test_exact_slicer_equals_spec_a_away_from_zero(sps):
    """against the model's own spec A (binary32 arctangent polynomial, 4e-6 rad per step): same bit wherever |S_A| > 1e-4"""
    x = _fsk(64 * 2000, sps, 8.0, seed=7)
    fa, fd = oracle.Fused(0, sps, slicer=0), oracle.Fused(0, sps, slicer=3)
    fa.push(x)
    fd.push(x)
    _, Sa, ga = fa.taps()
    gd = fd.taps()[2]
    diff = np.nonzero(ga[sps:] != gd[sps:])[0] + sps
    assert (np.abs(Sa[diff]) <= TOL_RAD).all()


@pytest.mark.parametrize("blocks", [[64], [1, 63, 777, 4096, 10000], [2047, 2049]])
def test_exact_slicer_model_is_push_invariant(blocks):
    from gr_amps_amd import synth
    x, truth = synth.make_channel_block(90000, 2, seed=42, snr_db=15.0)
    one = oracle.Fused(0, 10, slicer=3)
    ref = one.push(x)
    assert len(ref) == len(truth) and all(r["min"].decode() == t[2] for r, t in zip(ref, truth))
    f = oracle.Fused(0, 10, slicer=3)
    got, off, k = [], 0, 0
    while off < len(x):
        m = min(blocks[k % len(blocks)], len(x) - off)
        got.append(f.push(x[off:off + m]))
        off += m
        k += 1
    got = np.concatenate(got)
    assert got.tobytes() == ref.tobytes()
    assert np.array_equal(f.taps()[2], one.taps()[2][:len(f.taps()[2])])
