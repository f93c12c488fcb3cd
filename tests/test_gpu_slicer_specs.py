"""-m gpu parity tests of the atan-free slicer specs of include/amps_recc_numerics.h:
  spec B (AMPS_RECC_FLAG_SLICER_PRODUCT): the sign of Im(x[n] conj(x[n-sps])) instead of discriminator + boxcar;
  spec C (AMPS_RECC_FLAG_SLICER_SINE):    spec A's boxcar over Im(x[n] conj(x[n-1])), no arctangent;
  spec D (AMPS_RECC_FLAG_SLICER_EXACT):   the sign of spec A's boxcar sum from sign bits and the winding number.
Every check goes through the C ABI and compares with the CPU model (oracle/fused_model.c, orc_fused_set_slicer)
bit for bit; the words must also equal those of spec A (the arctangent discriminator, the default of rounds 1-3) on the same bursts."""
import numpy as np
import pytest

import oracle
from gr_amps_amd import capi, synth, synth_wideband as sw

pytestmark = pytest.mark.gpu
D = 512
WORD_FIELDS = ("channel", "dcc", "valid", "first_valid_rep", "word_raw", "word_dec", "msg_class", "a_MIN1", "b_MIN2",
               "esn", "dialed", "min", "manch_bad")


def _channels(C, N, seed0, nb=1, snr=30.0, sps=10):
    iq, truth = [], []
    for c in range(C):
        x, t = synth.make_channel_block(N, nb, seed=seed0 + c, snr_db=snr, sps=sps)
        iq.append(x)
        truth.append(t)
    return np.stack(iq), truth


SPECS = [("product", 1), ("sine", 2), ("exact", 3)]


@pytest.mark.parametrize("spec,sid", SPECS)
@pytest.mark.parametrize("sps", [3, 4, 5, 6, 8, 10, 12])
def test_product_slicer_iq_matches_cpu_model_every_sps(gpu, sps, spec, sid):
    C, N = 3, 3456 * sps * 2 + 5000
    iq, truth = _channels(C, N, 900 + sps, nb=2, sps=sps)
    with capi.Recc(n_channels=C, sps=sps, max_samples=N, max_bursts=64, slicer=spec) as r:
        r.push_iq(iq)
        got = r.drain()
    want = oracle.fused_push_all(iq, sps=sps, slicer=sid)
    assert len(want) == sum(len(t) for t in truth)
    assert got.tobytes() == want.tobytes()
    # the transmitted words came back
    k = 0
    for c in range(C):
        for (off, kind, min10, esn, dialed, words) in truth[c]:
            assert got[k]["min"].decode() == min10 and got[k]["valid"].all()
            k += 1


@pytest.mark.parametrize("spec,sid", SPECS)
def test_product_slicer_bits_and_statistic_bit_exact(gpu, spec, sid):
    iq, _ = _channels(1, 50000, 950, nb=1)
    x = iq[0]
    with capi.Recc(n_channels=1, sps=10, max_samples=65536, max_bursts=8, slicer=spec) as r:
        d, s, g = r.debug_demod(x)
    f = oracle.Fused(0, 10, slicer=sid)
    f.push(x)
    md, ms, mg = f.taps()
    assert len(g) == len(mg)
    assert np.array_equal(g, mg)
    assert np.array_equal(s.view(np.uint32), ms.view(np.uint32))
    assert np.array_equal(d.view(np.uint32), md.view(np.uint32))
    # inside the burst (carrier on) spec B slices exactly like spec A
    fa = oracle.Fused(0, 10)
    fa.push(x)
    _, _, ga = fa.taps()
    assert (g != ga).mean() < 0.25          # they only differ in carrier-off noise (phase wraps)


@pytest.mark.parametrize("spec,sid", SPECS)
@pytest.mark.parametrize("blocks", [[64], [1, 63, 777, 4096, 10000], [2047, 2049], [40000, 1, 1, 30000]])
def test_product_slicer_ragged_pushes(gpu, blocks, spec, sid):
    C, N = 3, 90000
    iq, truth = _channels(C, N, 970, nb=2)
    models = [oracle.Fused(c, 10, slicer=sid) for c in range(C)]
    with capi.Recc(n_channels=C, sps=10, max_samples=65536, max_bursts=64, slicer=spec) as r:
        off, k, total = 0, 0, 0
        while off < N:
            m = min(blocks[k % len(blocks)], N - off)
            r.push_iq(np.ascontiguousarray(iq[:, off:off + m]))
            got = r.drain()
            want = np.concatenate([models[c].push(iq[c, off:off + m]) for c in range(C)])
            assert got.tobytes() == want.tobytes(), f"push at {off} (+{m})"
            total += len(got)
            off += m
            k += 1
            if len(blocks) == 1 and k > 40:
                break
    if len(blocks) > 1:
        assert total == sum(len(t) for t in truth)


@pytest.mark.parametrize("spec,snr", [("product", 30.0), ("product", 18.0), ("sine", 30.0), ("sine", 18.0), ("sine", 12.0),
                                      ("exact", 30.0), ("exact", 18.0), ("exact", 12.0)])
def test_product_slicer_words_equal_spec_a(gpu, snr, spec):
    """same bursts through both numeric specs: identical words, fields and validity (the run centre may move by a sample)"""
    C, N = 8, 4 * 40000
    iq, truth = _channels(C, N, 1000, nb=4, snr=snr)
    out = {}
    for sp in ("atan", spec):
        with capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=256, slicer=sp) as r:
            r.push_iq(iq)
            out[sp] = r.drain()
    a, b = out["atan"], out[spec]
    assert len(a) == len(b) == sum(len(t) for t in truth)
    fields = WORD_FIELDS if snr >= 30.0 else ("channel", "valid", "word_dec", "msg_class", "min")   # raw bits may differ in noise
    for f in fields:
        assert np.array_equal(a[f], b[f]), f
    assert np.abs(a["position"].astype(np.int64) - b["position"].astype(np.int64)).max() <= 1


@pytest.mark.parametrize("spec,sid", SPECS)
def test_product_slicer_wideband_fused_unfused_and_cpu_model(gpu, spec, sid, decim):
    """wideband seam under spec B: fused kernel == two-kernel form == CPU model on the channelizer's own output"""
    first, C, D = 96, 832, decim
    sps = 1536 // D
    n = int(0.25 * sw.FS_WIDE) // 1536 * 1536
    planted = [(first + 3, 100000), (first + 417, 150000), (first + 830, 60000), (first + 831, 200000)]
    x, truth = sw.make_wideband(n, planted, seed=31)
    wb = {"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}
    with capi.Recc(n_channels=C, sps=sps, max_samples=n // D + 72, max_bursts=64, wideband=wb) as r:
        chan = r.debug_channelize(x)
    outs = []
    for unfused, chunks in ((False, [n]), (True, [n]), (False, [100000, 1, 4000000, n])):
        with capi.Recc(n_channels=C, sps=sps, max_samples=n // D + 72, max_bursts=64, wideband=wb, unfused_wideband=unfused,
                       slicer=spec) as r:
            off = 0
            for m in chunks:
                m = min(m, n - off)
                if m <= 0:
                    break
                r.push_wideband(x[off:off + m])
                off += m
            r.push_wideband(np.zeros(64 * D, np.complex64))
            outs.append(r.drain())
    assert sorted(int(g["channel"]) for g in outs[0]) == sorted(k - first for k, _ in planted)
    assert outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes()
    active = sorted(int(g["channel"]) for g in outs[0])
    want = oracle.fused_push_all(chan[active], sps=sps, slicer=sid)
    want["channel"] = np.array(active, np.uint32)[want["channel"]]
    assert outs[0].tobytes() == want.tobytes()
    for (k, off), (kind, min10, esn, dialed, words) in truth.items():
        g = outs[0][active.index(k - first)]
        assert g["min"].decode() == min10 and g["valid"].all()
    # and the words equal those of spec A
    with capi.Recc(n_channels=C, sps=sps, max_samples=n // D + 72, max_bursts=64, wideband=wb) as r:
        r.push_wideband(x)
        r.push_wideband(np.zeros(64 * D, np.complex64))
        a = r.drain()
    for f in WORD_FIELDS:
        assert np.array_equal(a[f], outs[0][f]), f


def test_round2_golden_fixture_on_the_device(gpu):
    """the committed round-2 vectors (tests/golden/recc_golden_r02.npz): records of specs B and C on the round-1 IQ block, records
    and slicer bit streams (sha256 of the hard-decision tap) of all three specs on the 10 dB block"""
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    g1 = np.load(os.path.join(here, "golden", "recc_golden.npz"))
    g2 = np.load(os.path.join(here, "golden", "recc_golden_r02.npz"))
    x = (g1["iq_i16"].astype(np.float32) / 8192.0).view(np.complex64)
    xn = (g2["noisy_i8"].astype(np.float32) / 48.0).view(np.complex64).reshape(2, -1)
    for name in ("atan", "product", "sine"):
        if name != "atan":
            with capi.Recc(n_channels=1, sps=10, max_samples=65536, max_bursts=16, slicer=name) as r:
                r.push_iq(x[None, :])
                assert r.drain().view(np.uint8).tobytes() == g2["iq_records_" + name].tobytes()
        # the round-2 records were taken at one fixed phase per capture; the tracking default is pinned by recc_golden_r04.npz
        with capi.Recc(n_channels=2, sps=10, max_samples=xn.shape[1], max_bursts=16, slicer=name, fixed_timing=True) as r:
            r.push_iq(xn)
            assert r.drain().view(np.uint8).tobytes() == g2["noisy_records_" + name].tobytes()
        for c in range(2):
            with capi.Recc(n_channels=1, sps=10, max_samples=xn.shape[1], max_bursts=16, slicer=name) as r:
                bits = r.debug_demod(xn[c])[2]
            assert hashlib.sha256(bits.tobytes()).hexdigest() == str(g2["noisy_bits_sha_" + name][c]), (name, c)



@pytest.mark.parametrize("sps,snr", [(3, 30.0), (3, 6.0), (10, 30.0), (10, 3.0), (12, 0.0)])
def test_exact_slicer_is_the_sign_of_the_libm_boxcar_on_the_device(gpu, sps, snr):
    """spec D's claim, checked on the device against an independent float64 statement: its bit is (sum of the last sps libm atan2
    phase steps >= 0) wherever that sum is more than 1e-4 rad away from zero -- no arctangent is evaluated on the device"""
    iq, _ = _channels(1, 3456 * sps + 20000, 990 + sps, nb=1, snr=snr, sps=sps)
    x = iq[0]
    with capi.Recc(n_channels=1, sps=sps, max_samples=65536, max_bursts=8, slicer="exact") as r:
        g = r.debug_demod(x[:65536])[2]
    n = len(g)
    xc = x[:n].astype(np.complex128)
    t = xc * np.conj(np.concatenate([[0], xc[:-1]]))
    d = np.where(t == 0, 0.0, np.angle(t))
    cs = np.concatenate([[0.0], np.cumsum(d)])
    idx = np.arange(n)
    S = cs[idx + 1] - cs[np.maximum(idx - sps + 1, 0)]
    diff = np.nonzero(g[sps:] != (S[sps:] >= 0))[0] + sps
    assert len(diff) <= 2 and (np.abs(S[diff]) <= 1.0e-4).all(), (len(diff), np.abs(S[diff]).max() if len(diff) else 0.0)
    assert (np.abs(S) > np.pi).any() or snr > 10.0          # the noisy cases leave (-pi, pi]: the winding number is exercised
