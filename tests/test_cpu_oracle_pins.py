"""Pins for the parts of the oracle no reference vector reaches ("pin the oracle harder"): every claim the
restatement makes about third-party arithmetic is checked against an INDEPENDENT statement of the published algorithm --
brute force for IT++'s BCH(63,2) bounded-distance decoder, closed-form invariants for GNU Radio's firdes / fast_atan2f /
MMSE table.  What stays unpinned is listed in oracle/amps_oracle.h."""
import itertools

import numpy as np

import bchref
import oracle


def _oracle_decode(word):
    ok, out, nf = oracle.bch63_decode(np.array(bchref.bits(word), np.uint8))
    return ok, bchref.from_bits(out), nf


def test_generator_divides_x63_minus_1_and_is_m1_times_m3():
    assert oracle.bch_generator() == bchref.G
    assert bchref.polymod((1 << 63) | 1) == 0                         # cyclic code of length 63
    for p in (1, 2, 3, 4, 6):                                          # alpha, alpha^2, alpha^3, alpha^4, alpha^6 are roots of g
        assert bchref.evaluate(bchref.G, p) == 0
    assert bchref.evaluate(bchref.G, 5) != 0


def test_encoder_is_systematic_polynomial_division():
    rng = np.random.default_rng(0)
    for _ in range(200):
        msg = rng.integers(0, 2, 51)
        cw = bchref.from_bits(list(oracle.bch_encode(msg)))
        m = bchref.from_bits(list(msg) + [0] * 12)
        assert cw == m | bchref.polymod(m)                              # message, then the remainder of x^12 m(x) mod g(x)
        assert bchref.polymod(cw) == 0


def test_decoder_equals_brute_force_on_all_4096_syndromes():
    """One received word per syndrome (the remainder itself): bounded-distance decoding succeeds exactly on the 2017
    syndromes of the patterns of weight <= 2 and returns the nearest code word; on the other 2079 the IT++ iteration
    reports failure -- EXCEPT its documented quirk: S1 = 0 and S3 a cube gives Lambda = 1 + S3 x^3 with three roots, which
    IT++ accepts (#roots == deg Lambda) and "corrects" three positions."""
    lead = bchref.coset_leaders()
    assert len(lead) == 1 + 63 + 1953
    n_ok = n_fail = n_quirk = 0
    for r in range(4096):
        ok, out, nf = _oracle_decode(r)
        if r in lead:
            w, e = lead[r]
            assert ok and nf == w and out == r ^ e and bchref.polymod(out) == 0
            n_ok += 1
        else:
            s1, s3 = bchref.evaluate(r, 1), bchref.evaluate(r, 3)
            if s1 == 0 and bchref.is_cube(s3):
                assert ok and nf == 3 and bchref.polymod(out) == 0 and bin(out ^ r).count("1") == 3
                n_quirk += 1
            else:
                assert not ok
                n_fail += 1
    assert (n_ok, n_quirk, n_fail) == (2017, 21, 2058)


def test_every_weight3_pattern_follows_the_itpp_rule():
    """all C(63,3) = 39711 patterns on the zero code word: decoded to another code word at distance 2 (nf == 2, result != 0),
    to the sent word through the S1 = 0 cube case (nf == 3), or rejected -- never anything else"""
    lead = bchref.coset_leaders()
    counts = {"other_codeword": 0, "cube": 0, "reject": 0}
    for i, j, k in itertools.combinations(range(63), 3):
        e = (1 << i) | (1 << j) | (1 << k)
        ok, out, nf = _oracle_decode(e)
        r = bchref.polymod(e)
        if r in lead:
            w, e2 = lead[r]
            assert ok and nf == w == 2 and out == e ^ e2 and out != 0 and bchref.polymod(out) == 0
            counts["other_codeword"] += 1
        elif bchref.evaluate(e, 1) == 0 and bchref.is_cube(bchref.evaluate(e, 3)):
            # alpha^i + alpha^j + alpha^k = 0 => S3 = alpha^(i+j+k): a cube iff 3 | i+j+k.  Lambda = 1 + S3 x^3 then has its three
            # roots and IT++ flips THOSE positions (the cube roots of 1/S3, in general not i, j, k): a code word 3 away from rx
            assert (i + j + k) % 3 == 0
            assert ok and nf == 3 and bchref.polymod(out) == 0 and bin(out ^ e).count("1") == 3
            counts["cube"] += 1
            counts["cube_back_to_sent"] = counts.get("cube_back_to_sent", 0) + int(out == 0)
        else:
            assert not ok
            counts["reject"] += 1
    assert counts["other_codeword"] + counts["cube"] + counts["reject"] == 39711
    assert counts["other_codeword"] > 0 and counts["reject"] > 0 and counts["cube"] > 0 and counts["cube_back_to_sent"] > 0


# ------------------------------------------------------------------ GNU Radio 3.7 invariants (published behaviour of the blocks)
def test_firdes_low_pass_published_invariants():
    """gr::filter::firdes::low_pass(gain, fs, fc, width, WIN_BLACKMAN): ntaps = int(74 fs / (22 width)) made odd; taps are the
    windowed sinc scaled so that their SUM is the gain (DC gain); linear phase; -6 dB at the cutoff; Blackman stop band."""
    for gain, fs, fc, width in ((3, 400e3, 10e3, 4.5e3), (1, 200e3, 8e3, 2e3), (2.5, 48e3, 3e3, 1e3)):
        taps = oracle.firdes_low_pass(gain, fs, fc, width).astype(np.float64)
        n = int(74.0 * fs / (22.0 * width))
        assert taps.size == (n | 1)
        assert abs(taps.sum() - gain) < 2e-5 * gain
        assert np.allclose(taps, taps[::-1], atol=1e-7 * gain)
        w = np.exp(-2j * np.pi * np.outer(np.array([0.0, fc, fc + width, fs / 4]), np.arange(taps.size)) / fs)
        h = np.abs(w @ taps) / gain
        assert abs(h[0] - 1) < 1e-4 and abs(20 * np.log10(h[1]) + 6.02) < 0.3      # half amplitude at the cutoff
        assert 20 * np.log10(h[2]) < -65 and 20 * np.log10(h[3]) < -70             # Blackman: ~74 dB beyond the transition band


def test_fast_atan2f_published_invariants():
    """gr::fast_atan2f: 255-entry table of atan on [0,1] with linear interpolation, octant unfolding; special cases return
    exact constants; worst-case error of a 255-interval linear interpolation of atan is h^2/8 max|atan''| = 1.25e-6"""
    f = oracle.fast_atan2f
    assert f(0.0, 0.0) == 0.0
    assert f(0.0, 1.0) == 0.0 and abs(f(0.0, -1.0) - np.pi) < 1e-6
    assert abs(f(1.0, 0.0) - np.pi / 2) < 1e-6 and abs(f(-1.0, 0.0) + np.pi / 2) < 1e-6
    assert abs(f(1.0, 1.0) - np.pi / 4) < 2e-6 and abs(f(-1.0, -1.0) + 3 * np.pi / 4) < 2e-6
    th = np.linspace(-np.pi, np.pi, 20001)[1:-1]
    for r in (1e-3, 1.0, 37.5):
        err = max(abs(f(float(np.float32(r * np.sin(t))), float(np.float32(r * np.cos(t)))) - t) for t in th)
        assert err < 1e-5, (r, err)
    # odd in y, and the octant folds are reflections
    for y, x in ((0.3, 0.9), (0.9, 0.3), (0.5, -0.2), (2.0, -7.0)):
        assert abs(f(y, x) + f(-y, x)) < 1e-7
        assert abs(f(y, x) + f(x, y) - np.pi / 2) < 4e-6 if x > 0 and y > 0 else True


def test_mmse_table_published_invariants():
    """gr::filter::mmse_fir_interpolator_ff: 129 rows x 8 taps; row 0 and row 128 are the unit impulses on the two centre taps;
    row s and row 128-s are mirror images; every row has unit DC gain to ~1e-3 (it is a least-squares, not a Lagrange, fit);
    a band-limited signal is interpolated to ~1e-3"""
    t = oracle.mmse_taps().astype(np.float64)
    assert t.shape == (129, 8)
    assert np.allclose(t[0], np.eye(8)[3], atol=1e-6) or np.allclose(t[0], np.eye(8)[4], atol=1e-6)
    assert np.allclose(t[128], np.eye(8)[4], atol=1e-6) or np.allclose(t[128], np.eye(8)[3], atol=1e-6)
    assert np.allclose(t, t[::-1, ::-1], atol=2e-6)
    assert np.abs(t.sum(axis=1) - 1).max() < 2e-3
    assert np.abs(np.diff(t, axis=0)).max() < 0.02                     # 128 steps per sample: neighbouring rows are close
    k = np.arange(8)
    for f0 in (0.02, 0.1, 0.2):
        for s in range(0, 129, 8):
            x = np.cos(2 * np.pi * f0 * k + 0.7)
            a, b = (3, 4) if t[0][3] > 0.5 else (4, 3)
            ref = np.cos(2 * np.pi * f0 * (a + (b - a) * s / 128.0) + 0.7)
            assert abs(t[s] @ x - ref) < 4e-3


def test_in_source_constants_are_consistent_with_each_other():
    """lib/recc_impl.cc:76 ends in lib/focc_impl.cc:189's word sync; :186's dotting is its alternating head"""
    trig = "1010101010101010101010101011100010010"
    assert trig.endswith("11100010010") and trig[:26] == "10" * 13
    t = oracle.trigger()
    assert "".join(str(int(b)) for b in t[1::2]) == trig            # second symbol of each pair carries the bit ('1' -> (0,1))
    assert np.array_equal(t[0::2], 1 - t[1::2])
