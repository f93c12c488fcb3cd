"""CPU-only go / no-go for a filter-bank decimation D: burst loss of the fused seam's CPU model behind the
prototype filter at D = 512 (3 samples per symbol) and D = 768 (2) for two prototype cutoffs, one channel simulated at fs / 64 = 480 ksps
(the prototype decimated by 64 is alias-free to -80 dB: cutoff 13 kHz, stop band 22.5 kHz), bursts at random offsets on that
grid, white noise, C/N stated in 30 kHz.  TEST INFRASTRUCTURE: the oracle's fused model, nothing of the product.
usage: python scripts/sens_decim_cpu.py [bursts_per_point] [ppm] [cfo_hz]"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 400
PPM = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
CFO = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
FS = 480e3
SPS = 24
FORMS = [(512, 8, 13e3), (768, 12, 13e3), (768, 12, 15e3), (512, 8, 15e3)]      # (D, decimation of the 480 ksps stream, prototype cutoff)


def job(args):
    seed, cn = args
    from gr_amps_amd import synth
    import oracle
    from oracle import channelizer as cz
    rng = np.random.default_rng(seed)
    kind, min10, esn, dialed, words = synth.random_message(rng)
    bits = synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)
    n = 3456 * SPS + 6000 * SPS // 10
    off = 1500 + int(rng.integers(0, 4 * SPS))
    snr = cn - 10.0 * np.log10(FS / 30e3)
    x = synth.fsk_modulate(n, [(off, bits)], sps=SPS, fs=FS, snr_db=snr, rng=rng, dtype=np.complex128, sym_ppm=PPM, cfo_hz=CFO)
    sent = [bytes(np.asarray(w, np.uint8)) for w in words]
    res = []
    ph0 = int(rng.integers(0, 24))
    for D, q, cut in FORMS:
        y = np.convolve(x, cz.design_taps(8, cutoff_hz=cut)[::64] * 64.0)[:n]
        ph = ph0 % q
        z = y[ph::q].astype(np.complex64)
        z = z[:z.size // 64 * 64]
        recs = oracle.fused_push_all(z[None, :], sps=SPS // q)
        good = 0
        for r in recs:
            if r["min"].decode() == min10 and all(bool(r["valid"][w]) and bytes(r["word_dec"][w]) == sent[w] for w in range(len(sent))):
                good = 1
        res.append(good)
    return res


if __name__ == "__main__":
    import oracle
    oracle.build()
    with mp.Pool(8) as pool:
        print(f"ppm {PPM} cfo {CFO}  bursts/point {NB}   C/N (30 kHz) and burst loss at D@cutoff = " + " / ".join(f"{d}@{c / 1e3:.0f}k" for d, _, c in FORMS))
        for cn in (7, 8, 9, 10, 11, 12, 13, 14, 16, 20):
            r = np.array(pool.map(job, [(77000 + 1000 * cn + i, cn) for i in range(NB)], chunksize=4))
            print(cn, " ".join(f"{1.0 - r[:, k].mean():.4f}" for k in range(r.shape[1])), flush=True)
