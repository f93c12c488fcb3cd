"""Burst loss of the wideband seam at D = 768 in reference mode against majority mode (AMPS_RECC_FLAG_MAJORITY: bitwise 3-of-5 vote, one BCH
decode per word, word A parsed from the CORRECTED word) on the SAME blocks, library default slicer, tracked capture.  DESIGN.md 4.2b: at two
samples per symbol a thin tail of bursts whose timing falls on the T/4 boundary is lost between 10 and 16 dB C/N, and since the reference
parses word A from its first repeat UNCORRECTED (lib/recc_decode_impl.cc:112) one raw bit error there costs the MIN although every word is
valid -- "majority mode does not have that weakness" was a statement; this measures it.
A burst is GOOD when a record on its channel carries the transmitted MIN and every transmitted word valid and equal to what was sent.
usage (GPU box): python scripts/majority_sensitivity.py [blocks_per_point]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NBLK = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CASES = [(0, 0), (100, 2000)]
SNRS = [8, 9, 10, 11, 12, 13, 14, 16, 20]


def main():
    import torch
    import widebandref as W
    from gr_amps_amd import capi
    dev = torch.device("cuda:0")
    first, Cw, D = 96, 832, 768
    n = int(0.45 * W.FS) // 1536 * 1536
    chans = list(range(0, Cw, 2))
    t0 = time.time()
    print("ppm cfo_Hz C/N_dB bursts | loss reference mode / majority mode | wrong valid words ref / maj", flush=True)
    for ppm, cfo in CASES:
        for snr in SNRS:
            good, wrong, sent = {False: 0, True: 0}, {False: [0, 0], True: [0, 0]}, 0
            for b in range(NBLK):
                x, planted = W.make_block(torch, dev, n, chans, first, ppm, cfo, float(snr), seed=31000 + 100 * snr + b)   # the blocks of decim_sensitivity.py
                sent += len(planted)
                for maj in (False, True):
                    with capi.Recc(n_channels=Cw, max_samples=n // D + 72, max_bursts=4096, majority=maj,
                                   wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}) as r:
                        r.push_wideband(x)
                        r.push_wideband(torch.zeros(64 * D, dtype=torch.complex64, device=dev))
                        recs = r.drain()
                    by = {}
                    for g in recs:
                        by.setdefault(int(g["channel"]), []).append(g)
                    for c, (min10, words) in planted.items():
                        rs = by.get(c, [])
                        good[maj] += W.good(rs, min10, words)
                        sentb = [bytes(np.asarray(w, np.uint8)) for w in words]
                        for g in rs:
                            for w in range(len(sentb)):
                                if g["valid"][w]:
                                    wrong[maj][1] += 1
                                    wrong[maj][0] += bytes(g["word_dec"][w]) != sentb[w]
            print("%4d %5d %5d %5d | %.4f / %.4f | %d of %d / %d of %d" % (ppm, cfo, snr, sent, 1.0 - good[False] / sent, 1.0 - good[True] / sent,
                  wrong[False][0], wrong[False][1], wrong[True][0], wrong[True][1]), flush=True)
    print("elapsed %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
