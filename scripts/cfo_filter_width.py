"""Carrier offset on the IQ seam: how much of the loss is the flow graph's 10 kHz channel filter, how much the slicer's threshold?
(VERDICT r05 item 8 / DESIGN.md 9.2.)  The same 400 ksps blocks as scripts/impairment_sweep.py (channel at +160 kHz, one burst per
block, white noise, C/N in 30 kHz) go through the TRANSLATE seam of the library (amps_recc_set_xlate + amps_recc_push_raw: the flow
graph's own filter stage on the GPU) with the filter's cut-off / transition width as the parameter; the first column is the flow
graph's filter (10 kHz / 4.5 kHz, 299 taps), which is what the IQ seam of the sweep sits behind.
usage (GPU box): python scripts/cfo_filter_width.py [bursts_per_point]"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 500
N_IQ = 40000
FILTERS = [(10e3, 4.5e3), (12e3, 4.5e3), (14e3, 4.5e3), (16e3, 4.5e3), (14e3, 9e3)]      # (cut-off, transition width) in Hz at 400 ksps
POINTS = [(12, 0), (12, 1000), (12, -1000), (12, 2000), (12, -2000), (12, 3000), (12, -3000), (10, 0), (10, 2000), (30, 2000), (30, 4000)]


def job(args):
    seed, snr, cfo = args
    from gr_amps_amd import synth
    x, t = synth.make_channel_block(2 * N_IQ, 1, seed=seed, sps=20, snr_db=float(snr) - 10.0 * np.log10(400.0 / 30.0), first=4000, cfo_hz=float(cfo))
    x = (x * np.exp(2j * np.pi * 0.4 * np.arange(x.size))).astype(np.complex64)
    off, kind, min10, esn, dialed, words = t[0]
    return x, min10, [list(w) for w in words]


def score(recs, min10, sent):
    sentb = [bytes(np.asarray(w, np.uint8)) for w in sent]
    for r in recs:
        if r["min"].decode() == min10 and all(bool(r["valid"][w]) and bytes(r["word_dec"][w]) == sentb[w] for w in range(len(sent))):
            return 1
    return 0


def main():
    pool = mp.get_context("fork").Pool(min(96, os.cpu_count() or 8))
    from gr_amps_amd import capi
    print("burst loss on the translate seam (library default slicer, tracked capture), %d bursts per point; columns = channel filter cut-off / width" % NB)
    print("C/N  cfo_Hz | " + " | ".join("%4.0fk/%3.1fk" % (c / 1e3, w / 1e3) for c, w in FILTERS), flush=True)
    for snr, cfo in POINTS:
        res = pool.map(job, [(770000 + 1000 * snr + i, snr, cfo) for i in range(NB)], chunksize=8)
        raw = np.stack([r[0] for r in res])
        out = []
        for cut, width in FILTERS:
            with capi.Recc(n_channels=NB, sps=10, max_samples=N_IQ + 64, max_bursts=4 * NB) as r:
                r.set_xlate(rate_hz=400e3, center_hz=160e3, decim=2, cutoff_hz=cut, width_hz=width)
                r.push_raw(raw)
                r.push_raw(np.zeros((NB, 2048), np.complex64))
                recs = r.drain()
            by = {}
            for g in recs:
                by.setdefault(int(g["channel"]), []).append(g)
            good = sum(score(by.get(c, []), res[c][1], res[c][2]) for c in range(NB))
            out.append(1.0 - good / NB)
        print("%3d %6d | " % (snr, cfo) + " | ".join("   %.4f " % v for v in out), flush=True)
    pool.close()


if __name__ == "__main__":
    main()
