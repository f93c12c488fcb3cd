"""Burst loss of the wideband seam at the two decimations of the filter bank -- D = 512 (2x oversampled, 3 samples per symbol, prototype
-6 dB at 13 kHz) and D = 768 (4/3 x oversampled, 2 samples per symbol, -6 dB at 15 kHz) -- on the SAME blocks, library default slicer
(spec D), tracked capture: what the 1.4 x throughput of the D = 768 form costs in sensitivity (within 1 dB of the
D = 512 form at the same C/N, +-100 ppm, +-2 kHz).  416 bursts per block (every second channel of the band), white noise, C/N stated in
30 kHz; every mobile off by `ppm` in its bit clock and `cfo` Hz in its carrier, signs alternating from channel to channel.
A burst is GOOD when a record on its channel carries the transmitted MIN and every transmitted word valid and equal to what was sent.
usage (GPU box): python scripts/decim_sensitivity.py [blocks_per_point]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NBLK = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CASES = [(0, 0), (100, 0), (100, 2000), (500, 0)]
SNRS = [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 20, 30]


def crossing(snrs, loss, level=0.01):
    for i in range(len(snrs) - 1):
        a, b = loss[i], loss[i + 1]
        if a > level >= b:
            la, lb = np.log10(max(a, 1e-6)), np.log10(max(b, 1e-6))
            return snrs[i] + (la - np.log10(level)) / (la - lb) * (snrs[i + 1] - snrs[i])
    return None


def main():
    import torch
    import widebandref as W
    from gr_amps_amd import capi
    dev = torch.device("cuda:0")
    first, Cw = 96, 832
    n = int(0.45 * W.FS) // 1536 * 1536
    chans = list(range(0, Cw, 2))
    t0 = time.time()
    print("ppm cfo_Hz C/N_dB bursts | loss D512 / D768 | wrong valid words D512 / D768", flush=True)
    cross = {}
    for ppm, cfo in CASES:
        losses = {512: [], 768: []}
        for snr in SNRS:
            good = {512: 0, 768: 0}
            wrong = {512: [0, 0], 768: [0, 0]}
            sent = 0
            for b in range(NBLK):
                x, planted = W.make_block(torch, dev, n, chans, first, ppm, cfo, float(snr), seed=31000 + 100 * snr + b)
                sent += len(planted)
                for D in (512, 768):
                    with capi.Recc(n_channels=Cw, max_samples=n // D + 72, max_bursts=4096,
                                   wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}) as r:
                        r.push_wideband(x)
                        r.push_wideband(torch.zeros(64 * D, dtype=torch.complex64, device=dev))
                        recs = r.drain()
                    by = {}
                    for g in recs:
                        by.setdefault(int(g["channel"]), []).append(g)
                    for c, (min10, words) in planted.items():
                        rs = by.get(c, [])
                        good[D] += W.good(rs, min10, words)
                        sentb = [bytes(np.asarray(w, np.uint8)) for w in words]
                        for g in rs:
                            for w in range(len(sentb)):
                                if g["valid"][w]:
                                    wrong[D][1] += 1
                                    wrong[D][0] += bytes(g["word_dec"][w]) != sentb[w]
            for D in (512, 768):
                losses[D].append(1.0 - good[D] / sent)
            print("%4d %5d %5d %5d | %.4f / %.4f | %.1e (%d) / %.1e (%d)" % (ppm, cfo, snr, sent, losses[512][-1], losses[768][-1],
                  wrong[512][0] / max(1, wrong[512][1]), wrong[512][1], wrong[768][0] / max(1, wrong[768][1]), wrong[768][1]), flush=True)
        cross[(ppm, cfo)] = (crossing(SNRS, losses[512]), crossing(SNRS, losses[768]))
    print("\nC/N (dB in 30 kHz) at 1 % burst loss, log-linear interpolation:")
    for (ppm, cfo), (a, b) in cross.items():
        f = lambda v: "n/a" if v is None else "%.2f" % v
        print("  %4d ppm %5d Hz: D512 %s  D768 %s  (D768 - D512 = %s dB)" % (ppm, cfo, f(a), f(b), "n/a" if a is None or b is None else "%+.2f" % (b - a)))
    print("elapsed %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
