"""Symbol-clock and carrier offsets of the mobile: burst loss of the fused seams (tracking capture = default, and
AMPS_RECC_FLAG_FIXED_TIMING = rounds 1-3) beside the restated reference chain, whose Mueller & Mueller loop tracks +-0.5 %
(grc/recctest.grc:846-874).  VERDICT r03 item 2.

  IQ seam   : one burst per block as the flow graph's source delivers it (400 ksps, channel at +160 kHz, white noise), through the
              flow graph's channel filter (oracle.freq_xlating_fir, 299 taps, decim 2); the 200 ksps stream goes to
              amps_recc_push_iq (library default slicer) with and without tracking, and to oracle.chain_iq200.
  wideband  : 0.45 s blocks at 30.72 Msps, a burst in every second channel; amps_recc_push_wideband with and without tracking;
              reference column = every planted channel cut out at 400 ksps (+160 kHz) through oracle.chain_iq400.

C/N is stated in 30 kHz.  A burst is GOOD when a record on its channel carries the transmitted MIN and every transmitted word
valid and equal to what was sent.  Per-word columns: fraction of bursts whose words 0-1 / last two transmitted words were lost
(a clock offset hurts the late words first).  usage (GPU box): python scripts/impairment_sweep.py [bursts_per_point]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 500
SNRS = (12, 30)
PPMS = (0, 50, -50, 100, -100, 200, -200, 500, -500)
CFOS = (1000, -1000, 2000, -2000, 4000, -4000)
POINTS = [(p, 0) for p in PPMS] + [(0, c) for c in CFOS] + [(100, 2000), (-100, -2000)]
N_IQ = 40000
FS = 30.72e6
_TAPS = None


def _score(recs, min10, sent):
    """(good, first two words ok, last two words ok) of the records attributed to one planted burst"""
    sentb = [bytes(np.asarray(w, np.uint8)) for w in sent]
    good = head = tail = 0
    for r in recs:
        ok = [bool(r["valid"][w]) and bytes(r["word_dec"][w]) == sentb[w] for w in range(len(sent))]
        if r["min"].decode() == min10 and all(ok):
            good = 1
        head |= int(all(ok[:2]))
        tail |= int(all(ok[-2:]))
    return good, head, tail


def iq_job(args):
    global _TAPS
    seed, snr, ppm, cfo = args
    from gr_amps_amd import synth
    import oracle
    if _TAPS is None:
        _TAPS = oracle.firdes_low_pass(3.0, 400e3, 10e3, 4.5e3)
    x, t = synth.make_channel_block(2 * N_IQ, 1, seed=seed, sps=20, snr_db=float(snr) - 10.0 * np.log10(400.0 / 30.0), first=4000,
                                    sym_ppm=float(ppm), cfo_hz=float(cfo))
    x = (x * np.exp(2j * np.pi * 0.4 * np.arange(x.size))).astype(np.complex64)
    y = oracle.freq_xlating_fir(x, _TAPS, 160e3, 400e3, 2)[:N_IQ].astype(np.complex64)
    off, kind, min10, esn, dialed, words = t[0]
    return y, min10, [list(w) for w in words], _score(oracle.chain_iq200(y, channel=0), min10, words)


def ref400_job(args):
    seg, min10, words = args
    import oracle
    return _score(oracle.chain_iq400(seg, 160e3, chunk=4096), min10, words)


def main():
    pool = mp.get_context("fork").Pool(min(96, os.cpu_count() or 8))     # forked BEFORE the GPU is touched
    import torch
    from gr_amps_amd import capi, synth, synth_wideband as sw
    dev = torch.device("cuda:0")
    t0 = time.time()
    spec = capi.SLICER_NAMES[capi.load().amps_recc_default_slicer()]
    print("library default slicer: %s; %d bursts per point; loss = 1 - good/sent, then (words 0-1 lost, last two words lost)" % (spec, NB))
    print("seam C/N ppm cfo_Hz sent | tracked (default) | fixed timing | restated reference chain", flush=True)
    fmt = lambda v, n: "%.4f (%.3f, %.3f)" % (1 - v[0] / n, 1 - v[1] / n, 1 - v[2] / n)
    for snr in SNRS:
        for ppm, cfo in POINTS:
            res = pool.map(iq_job, [(770000 + 1000 * snr + i, snr, ppm, cfo) for i in range(NB)], chunksize=8)
            iq = np.stack([r[0] for r in res])
            truth = [(r[1], r[2]) for r in res]
            cols = {"ref": np.array([r[3] for r in res]).sum(0)}
            for name, fixed in (("tracked", False), ("fixed", True)):
                with capi.Recc(n_channels=NB, sps=10, max_samples=N_IQ, max_bursts=4 * NB, fixed_timing=fixed) as r:
                    r.push_iq(iq)
                    recs = r.drain()
                by = {}
                for g in recs:
                    by.setdefault(int(g["channel"]), []).append(g)
                cols[name] = np.array([_score(by.get(c, []), *truth[c]) for c in range(NB)]).sum(0)
            print("iq   %3d %5d %6d %5d | %s | %s | %s" % (snr, ppm, cfo, NB, fmt(cols["tracked"], NB), fmt(cols["fixed"], NB), fmt(cols["ref"], NB)), flush=True)
    # ---------------- wideband seam
    first, Cw, D = 96, 832, int(os.environ.get("IMP_DECIM", "768"))      # the filter bank's decimation (768 = library default since round 6)
    print("wideband seam at D = %d" % D, flush=True)
    n = int(0.45 * FS) // D * D
    nout = n * 5 // 384
    assert n % 384 == 0
    nblk = max(1, (NB + 415) // 416)
    for snr in SNRS:
        for ppm, cfo in POINTS:
            tot = {k: np.zeros(3, np.int64) for k in ("tracked", "fixed", "ref")}
            sent_total = 0
            for b in range(nblk):
                rng = np.random.default_rng(88000 + 100 * snr + b)
                g = torch.Generator(device=dev)
                g.manual_seed(6000 + 100 * snr + b)
                sigma = 10.0 ** (-snr / 20.0) / np.sqrt(2.0) * np.sqrt(FS / 30e3)
                x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g, dtype=torch.float32) * float(sigma))
                planted = {}
                sps_w = 1536
                for c in range(0, Cw, 2):
                    k = (first + c) % 1024
                    _, min10, _, _, words = synth.random_message(rng)
                    sym = synth.manchester(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)).astype(np.float32) * 2 - 1
                    wave = synth.symbol_waveform(sym, sps_w, float(ppm)).astype(np.float32)
                    blen = wave.size
                    off = int(rng.integers(30000, n - blen - 30000))
                    f = torch.from_numpy(wave).to(dev) * (2 * np.pi * 8e3 / FS)
                    fc = 2 * np.pi * (sw.bin_freq(k) + cfo) / FS
                    ph = torch.cumsum(f.double() + fc, 0) + float(rng.uniform(0, 2 * np.pi)) + fc * off
                    x[off:off + blen] += torch.polar(torch.ones_like(ph, dtype=torch.float32), ph.remainder(2 * np.pi).float())
                    planted[c] = (min10, [list(w) for w in words], off, k, blen)
                sent_total += len(planted)
                for name, fixed in (("tracked", False), ("fixed", True)):
                    with capi.Recc(n_channels=Cw, sps=1536 // D, max_samples=n // D + 72, max_bursts=4096, fixed_timing=fixed,
                                   wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}) as r:
                        r.push_wideband(x)
                        r.push_wideband(torch.zeros(64 * D, dtype=torch.complex64, device=dev))
                        recs = r.drain()
                    by = {}
                    for gr_ in recs:
                        by.setdefault(int(gr_["channel"]), []).append(gr_)
                    tot[name] += np.array([_score(by.get(c, []), planted[c][0], planted[c][1]) for c in planted]).sum(0)
                X = torch.fft.fft(x.to(torch.complex128))
                jobs = []
                for c, (min10, words, off, k, blen) in planted.items():
                    cbin = int(round((sw.bin_freq(k) - 160e3) / FS * n))
                    idx = (torch.arange(-nout // 2, nout // 2, device=dev) + cbin) % n
                    y = torch.fft.ifft(torch.fft.ifftshift(X[idx])) * (nout / n)
                    o4 = off * 5 // 384
                    seg = y[max(0, o4 - 6000):o4 + blen * 5 // 384 + 4000].to(torch.complex64).cpu().numpy()
                    jobs.append((seg, min10, words))
                del X
                tot["ref"] += np.array(pool.map(ref400_job, jobs, chunksize=2)).sum(0)
            print("wide %3d %5d %6d %5d | %s | %s | %s" % (snr, ppm, cfo, sent_total, fmt(tot["tracked"], sent_total), fmt(tot["fixed"], sent_total),
                                                        fmt(tot["ref"], sent_total)), flush=True)
    print("elapsed %.0f s" % (time.time() - t0))
    pool.close()


if __name__ == "__main__":
    main()
