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


# ---- the kernels' own bit logic, evaluated on the host through the C ABI (amps_recc_debug_exact_slice): no device needed
def _spec_d_bits(sx, st, sc, sps):
    """per-sample statement of include/amps_recc_numerics.h spec D on sign-bit arrays (index = time); positions < sps are don't-care"""
    n = len(sx)
    wp = np.zeros(n, int)
    wm = np.zeros(n, int)
    g = np.zeros(n, np.uint8)
    for i in range(n):
        s1 = sx[i - 1] if i >= 1 else 0
        wp[i] = (not sx[i]) and s1 and st[i]
        wm[i] = sx[i] and (not s1) and (not st[i])
        if i >= sps:
            ss = sx[i - sps]
            K = int((not sx[i]) and ss and sc[i]) - int(sx[i] and (not ss) and (not sc[i])) + int(wm[i - sps + 1:i + 1].sum()) - int(wp[i - sps + 1:i + 1].sum())
            g[i] = K > 0 or (K == 0 and not sc[i])
    return g, wp, wm


@pytest.mark.parametrize("sps", [3, 4, 5, 6, 8, 10, 12])
def test_streaming_kernels_window_logic_on_the_host(sps):
    import ctypes as C
    from gr_amps_amd import capi
    L = capi.load()
    L.amps_recc_debug_exact_slice.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(sps)
    for trial in range(300):
        # random sign words, with a bias towards long runs (real signals) in half of the trials
        if trial % 2:
            bits = [np.repeat(rng.integers(0, 2, 8), 4)[:32] ^ (rng.random(32) < 0.1) for _ in range(3)]
        else:
            bits = [rng.integers(0, 2, 32) for _ in range(3)]
        sx, st, sc = (np.asarray(b, int) for b in bits)
        words = (C.c_uint32 * 3)(*[int(sum(int(b[i]) << i for i in range(32))) for b in (sx, st, sc)])   # oldest sample at bit 0
        out = (C.c_uint32 * 3)()
        assert L.amps_recc_debug_exact_slice(0, sps, words, out) == 0
        g, _, _ = _spec_d_bits(sx, st, sc, sps)
        got = np.array([(out[0] >> i) & 1 for i in range(32)], np.uint8)
        assert np.array_equal(got[sps + 1:], g[sps + 1:]), (trial, sps)
    assert L.amps_recc_debug_exact_slice(0, 7, words, out) != 0          # unsupported samples per symbol


def test_filter_bank_word_logic_on_the_host():
    """the filter bank's form: newest frame at bit 0, the previous 32 frames in a second word, wrap words carried from call to call"""
    import ctypes as C
    from gr_amps_amd import capi
    L = capi.load()
    L.amps_recc_debug_exact_slice.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(33)
    n = 32 * 40
    sx, st, sc = (rng.integers(0, 2, n) for _ in range(3))
    g, wp, wm = _spec_d_bits(sx, st, sc, 3)

    def word(b, w):       # frames [32 w, 32 w + 32), newest at bit 0
        return int(sum(int(b[32 * w + 31 - i]) << i for i in range(32))) if w >= 0 else 0
    prev = [0, 0, 0]      # SX, wp, wm of the previous 32 frames
    for w in range(n // 32):
        inp = (C.c_uint32 * 6)(word(sx, w), word(st, w), word(sc, w), prev[0], prev[1], prev[2])
        out = (C.c_uint32 * 3)()
        assert L.amps_recc_debug_exact_slice(1, 3, inp, out) == 0
        lo = 3 if w == 0 else 0
        got = np.array([(out[0] >> (31 - i)) & 1 for i in range(32)], np.uint8)
        assert np.array_equal(got[lo:], g[32 * w + lo:32 * w + 32]), w
        assert out[1] == word(wp, w) and out[2] == word(wm, w)
        prev = [word(sx, w), out[1], out[2]]
