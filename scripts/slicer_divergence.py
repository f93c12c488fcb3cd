"""Word-level divergence of the atan-free slicer specs (B: product detector, C: sine discriminator) from spec A (the default of rounds 1-3; spec D since) over
SNR, on the IQ seam (10 samples per symbol, SNR in the 200 kHz sample bandwidth) and on the wideband seam (SNR in a channel's 60 kHz).
For every SNR: bursts transmitted / found (trigger) / decoded with the transmitted MIN and all sent words valid, per spec, and the
number of bursts whose decoded words differ from spec A's.  usage (GPU box): python scripts/slicer_divergence.py [bursts_per_point]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi, synth, synth_wideband as sw  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 96
SPECS = ("atan", "sine", "product")


def score(recs, truth_by_key, key):
    """truth_by_key: {(channel, MIN): (MIN, sent words)}; a record counts once, under its channel and decoded MIN"""
    found = len(recs)
    good = 0
    words = {}
    for g in recs:
        k = key(g)
        t = truth_by_key.get(k)
        if t is None:
            continue
        min10, sent = t
        nsent = len(sent)
        ok = g["min"].decode() == min10 and g["valid"][:nsent].all() and all(list(g["word_dec"][w]) == list(sent[w]) for w in range(nsent))
        good += int(ok)
        words[k] = g["word_dec"][:nsent].tobytes()
    return found, good, words


print("seam snr_dB sent | " + " | ".join("%s found/good" % s for s in SPECS) + " | words != spec A (sine, product)")
for snr in (30, 24, 18, 15, 12, 10, 8):
    C, per = 8, NB // 8
    N = per * 40000 + 8000
    iq, truth = [], {}
    for c in range(C):
        x, t = synth.make_channel_block(N, per, seed=7000 + 100 * snr + c, snr_db=float(snr))
        iq.append(x)
        for i, (off, kind, min10, esn, dialed, wds) in enumerate(t):
            truth[(c, i)] = (min10, wds, off)
    iq = np.stack(iq)
    by_pos = {(c, m): (m, w) for (c, i), (m, w, off) in truth.items()}
    res = {}
    for sp in SPECS:
        with capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=1024, slicer=sp) as r:
            r.push_iq(iq)
            recs = r.drain()
        res[sp] = score(recs, by_pos, lambda g: (int(g["channel"]), g["min"].decode()))
    diff = [sum(1 for k, v in res["atan"][2].items() if res[sp][2].get(k) != v) for sp in ("sine", "product")]
    print("iq   %5d %4d | " % (snr, len(truth)) + " | ".join("%5d/%-5d" % res[sp][:2] for sp in SPECS) + " | %d, %d" % tuple(diff), flush=True)

first, Cw, D = 96, 832, 512
for snr in (30, 24, 18, 15, 12, 10, 8):
    n = int(0.45 * sw.FS_WIDE) // D * D
    rng = np.random.default_rng(snr)
    chans = rng.choice(Cw, size=min(NB, 64), replace=False)
    planted = [((first + int(c)) % 1024, int(rng.integers(20000, n - 3456 * 1536 - 20000))) for c in chans]
    x, truth = sw.make_wideband(n, planted, seed=8000 + snr, snr_db=float(snr))
    tb = {((k - first) % 1024, m): (m, w) for (k, off), (kind, m, esn, dialed, w) in truth.items()}
    res = {}
    for sp in SPECS:
        with capi.Recc(n_channels=Cw, sps=3, max_samples=n // D + 72, max_bursts=1024, slicer=sp,
                       wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": first}) as r:
            r.push_wideband(x)
            r.push_wideband(np.zeros(64 * D, np.complex64))
            recs = r.drain()
        res[sp] = score(recs, tb, lambda g: (int(g["channel"]), g["min"].decode()))
    diff = [sum(1 for k, v in res["atan"][2].items() if res[sp][2].get(k) != v) for sp in ("sine", "product")]
    print("wide %5d %4d | " % (snr, len(tb)) + " | ".join("%5d/%-5d" % res[sp][:2] for sp in SPECS) + " | %d, %d" % tuple(diff), flush=True)
