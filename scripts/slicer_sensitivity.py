"""Sensitivity of the four slicer specs (A atan + boxcar, B product detector, C sine discriminator, D exact sign) against the restated
reference chain: burst-loss and wrong-word rate over the carrier-to-noise ratio, on both seams.
C/N is stated in a 30 kHz AMPS channel bandwidth on both seams.

  IQ seam   : one burst per block, synthesised as the flow graph's source delivers it -- 400 ksps, the channel at +160 kHz, white
              noise (grc/recctest.grc:591) -- and passed through the flow graph's own channel filter (oracle.freq_xlating_fir
              with the 299 firdes.low_pass taps, decim 2: grc/recctest.grc:889-937, 115-155).  The resulting 200 ksps stream is
              what BOTH sides get: amps_recc_push_iq under specs A / B / C on the GPU, and oracle.chain_iq200 = quadrature_demod_cf
              -> clock_recovery_mm_ff -> binary_slicer_fb -> recc -> recc_decode (grc/recctest.grc:458, 846-874, 807).
  wideband  : 0.45 s blocks at 30.72 Msps, one burst in every second channel (416 per block) at a random offset, white noise.
              GPU: amps_recc_push_wideband under specs A / B / C.  Reference column: every planted channel is cut out of the same
              block at 400 ksps with the channel at +160 kHz (ideal FFT-domain band extraction, float64 -- what an N210 tuned
              160 kHz below the channel would deliver) and pushed through oracle.chain_iq400 = the flow graph from its 299-tap
              freq_xlating_fir_filter_ccc on.

A burst is GOOD when a record on its channel carries the transmitted MIN and every transmitted word valid and equal to what
was sent; LOST otherwise.  WRONG WORDS = words flagged valid whose 36 bits differ from the transmitted ones (undetected errors),
over the words of all records attributed to planted bursts.  usage (GPU box): python scripts/slicer_sensitivity.py [bursts_per_point]
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
SNRS = list(range(6, 19))
SNRS_HIGH = [20, 24, 30]          # where the restated reference chain's loss floor shows (its M&M loop must lock within the 4 spare dotting bits)
SPECS = ("atan", "product", "sine", "exact")
SKIP_REF = os.environ.get("SENS_NO_REF") == "1"       # the reference column costs most of the time; a quick A / B / C / D run skips it
N_IQ = 40000
FS = 30.72e6


# ------------------------------------------------------------------------------------------------- worker side (CPU only)
def _words_of(rec, nsent):
    return [bytes(rec["word_dec"][w]) for w in range(nsent)], [bool(rec["valid"][w]) for w in range(nsent)], rec["min"].decode()


def _score_recs(recs, min10, sent):
    """(good, words_valid, words_wrong) of the records attributed to ONE planted burst"""
    sentb = [bytes(np.asarray(w, np.uint8)) for w in sent]
    good, nvalid, nwrong = 0, 0, 0
    for r in recs:
        wd, va, m = _words_of(r, len(sent))
        if m == min10 and all(va) and wd == sentb:
            good = 1
        for w in range(len(sent)):
            if va[w]:
                nvalid += 1
                nwrong += int(wd[w] != sentb[w])
    return good, nvalid, nwrong


_TAPS = None


def iq_job(args):
    """one burst as the flow graph's source delivers it (400 ksps, channel at +160 kHz), through the flow graph's channel filter;
    the restated reference chain runs on the filtered 200 ksps stream the GPU seam gets as well"""
    global _TAPS
    seed, snr = args
    from gr_amps_amd import synth
    import oracle
    if _TAPS is None:
        _TAPS = oracle.firdes_low_pass(3.0, 400e3, 10e3, 4.5e3)
    # C/N in 30 kHz -> SNR in the 400 kHz sample bandwidth
    x, t = synth.make_channel_block(2 * N_IQ, 1, seed=seed, sps=20, snr_db=float(snr) - 10.0 * np.log10(400.0 / 30.0), first=4000)
    x = (x * np.exp(2j * np.pi * 0.4 * np.arange(x.size))).astype(np.complex64)
    y = oracle.freq_xlating_fir(x, _TAPS, 160e3, 400e3, 2)[:N_IQ].astype(np.complex64)
    off, kind, min10, esn, dialed, words = t[0]
    recs = oracle.chain_iq200(y, channel=0)
    return y, min10, [list(w) for w in words], _score_recs(recs, min10, words)


def ref400_job(args):
    seg, min10, words = args
    import oracle
    recs = oracle.chain_iq400(seg, 160e3, chunk=4096)
    return _score_recs(recs, min10, words)


# ------------------------------------------------------------------------------------------------- GPU side
def crossing(snrs, loss, level=0.01):
    """SNR (dB, linear interpolation of log10 loss) where the loss rate falls through `level`; None if it never does"""
    for i in range(len(snrs) - 1):
        a, b = loss[i], loss[i + 1]
        if a > level >= b:
            la, lb = np.log10(max(a, 1e-6)), np.log10(max(b, 1e-6))
            return snrs[i] + (la - np.log10(level)) / (la - lb) * (snrs[i + 1] - snrs[i])
    return None


def main():
    pool = mp.get_context("fork").Pool(min(96, os.cpu_count() or 8))     # forked BEFORE the GPU is touched
    import torch
    from gr_amps_amd import capi, synth, synth_wideband as sw
    dev = torch.device("cuda:0")
    t00 = time.time()
    table = {}

    # ---------------- IQ seam
    print("seam C/N_dB(30kHz) sent | loss A / B / C / D / ref | wrong-word rate A / B / C / D / ref (valid words)", flush=True)
    for snr in SNRS + SNRS_HIGH:
        res = pool.map(iq_job, [(910000 + 1000 * snr + i, snr) for i in range(NB)], chunksize=8)
        iq = np.stack([r[0] for r in res])
        truth = [(r[1], r[2]) for r in res]
        ref = np.array([r[3] for r in res]).sum(0)
        cols = {}
        for sp in SPECS:
            with capi.Recc(n_channels=NB, sps=10, max_samples=N_IQ, max_bursts=4 * NB, slicer=sp) as r:
                r.push_iq(iq)
                recs = r.drain()
            by = {}
            for g in recs:
                by.setdefault(int(g["channel"]), []).append(g)
            cols[sp] = np.array([_score_recs(by.get(c, []), *truth[c]) for c in range(NB)]).sum(0)
        cols["ref"] = ref
        table[("iq", snr)] = cols
        print("iq   %5d %5d | " % (snr, NB) + " / ".join("%.4f" % (1 - cols[k][0] / NB) for k in SPECS + ("ref",)) + " | "
              + " / ".join("%.1e (%d)" % (cols[k][2] / max(1, cols[k][1]), cols[k][1]) for k in SPECS + ("ref",)), flush=True)

    # ---------------- wideband seam
    first, Cw, D = 96, 832, 512
    n = int(0.45 * FS) // D * D
    nout = n * 5 // 384                                  # samples of the 400 ksps cut (n is a multiple of 384)
    assert n % 384 == 0
    blen = 3456 * 1536
    nblk = max(1, (NB + 415) // 416)
    for snr in SNRS + SNRS_HIGH:
        tot = {k: np.zeros(3, np.int64) for k in SPECS + ("ref",)}
        sent_total = 0
        for b in range(nblk):
            rng = np.random.default_rng(77000 + 100 * snr + b)
            g = torch.Generator(device=dev)
            g.manual_seed(5000 + 100 * snr + b)
            sigma = 10.0 ** (-snr / 20.0) / np.sqrt(2.0) * np.sqrt(FS / 30e3)      # C/N in 30 kHz
            x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g, dtype=torch.float32) * float(sigma))
            planted = {}
            for c in range(0, Cw, 2):
                k = (first + c) % 1024
                _, min10, _, _, words = synth.random_message(rng)
                sym = synth.manchester(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)).astype(np.float32) * 2 - 1
                off = int(rng.integers(30000, n - blen - 30000))
                f = torch.from_numpy(sym).to(dev).repeat_interleave(1536) * (2 * np.pi * 8e3 / FS)
                fc = 2 * np.pi * sw.bin_freq(k) / FS
                ph = torch.cumsum(f.double() + fc, 0) + float(rng.uniform(0, 2 * np.pi)) + fc * off
                x[off:off + blen] += torch.polar(torch.ones_like(ph, dtype=torch.float32), ph.remainder(2 * np.pi).float())
                planted[c] = (min10, [list(w) for w in words], off, k)
            sent_total += len(planted)
            # device under test
            for sp in SPECS:
                with capi.Recc(n_channels=Cw, sps=3, max_samples=n // D + 72, max_bursts=4096, slicer=sp,
                               wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": first}) as r:
                    r.push_wideband(x)
                    r.push_wideband(torch.zeros(64 * D, dtype=torch.complex64, device=dev))
                    recs = r.drain()
                by = {}
                for gr_ in recs:
                    by.setdefault(int(gr_["channel"]), []).append(gr_)
                tot[sp] += np.array([_score_recs(by.get(c, []), planted[c][0], planted[c][1]) for c in planted]).sum(0)
            # reference column: cut every planted channel out at 400 ksps, channel at +160 kHz, float64
            X = torch.fft.fft(x.to(torch.complex128))
            jobs = []
            for c, (min10, words, off, k) in planted.items():
                cbin = int(round((sw.bin_freq(k) - 160e3) / FS * n))
                idx = (torch.arange(-nout // 2, nout // 2, device=dev) + cbin) % n
                y = torch.fft.ifft(torch.fft.ifftshift(X[idx])) * (nout / n)
                o4 = off * 5 // 384
                seg = y[max(0, o4 - 6000):o4 + blen * 5 // 384 + 4000].to(torch.complex64).cpu().numpy()
                jobs.append((seg, min10, words))
            del X
            if not SKIP_REF:
                tot["ref"] += np.array(pool.map(ref400_job, jobs, chunksize=2)).sum(0)
        table[("wide", snr)] = tot
        table[("wide_sent", snr)] = sent_total
        print("wide %5d %5d | " % (snr, sent_total) + " / ".join("%.4f" % (1 - tot[k][0] / sent_total) for k in SPECS + ("ref",)) + " | "
              + " / ".join("%.1e (%d)" % (tot[k][2] / max(1, tot[k][1]), tot[k][1]) for k in SPECS + ("ref",)), flush=True)

    # ---------------- SNR at 1 % burst loss and the penalties
    print("\nSNR (dB) at 1 % burst loss, log-linear interpolation between the measured points:")
    allsnr = SNRS + SNRS_HIGH
    for seam in ("iq", "wide"):
        cr = {}
        for k in SPECS + ("ref",):
            loss = [1 - table[(seam, s)][k][0] / (NB if seam == "iq" else table[("wide_sent", s)]) for s in allsnr]
            cr[k] = crossing(allsnr, loss)
        fmt = lambda v: "n/a" if v is None else "%.2f" % v
        pen = lambda k: "n/a" if (cr[k] is None or cr["atan"] is None) else "%+.2f" % (cr[k] - cr["atan"])
        print("%-4s A %s | B %s | C %s | D %s | reference chain %s || penalty vs A: B %s dB, C %s dB, D %s dB, reference chain %s dB"
              % (seam, fmt(cr["atan"]), fmt(cr["product"]), fmt(cr["sine"]), fmt(cr["exact"]), fmt(cr["ref"]), pen("product"), pen("sine"), pen("exact"), pen("ref")))
    print("elapsed %.0f s" % (time.time() - t00))
    pool.close()


if __name__ == "__main__":
    main()
