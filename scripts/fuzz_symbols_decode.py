"""At-length differential fuzz of the exact drop-in seam: (1) amps_recc_push_symbols vs the recc_impl::work replica for random
streams (triggers anywhere, incl. around the 65536-byte wrap, overlapping and truncated bursts, occasional non-binary bytes) and
random chunk schedules 1..61439; (2) amps_recc_decode_bursts vs the restated bursts_message for bursts with random bit
damage per block (0..6 flips), all-noise bursts, non-binary bytes, reference and majority mode.
usage (GPU box): python scripts/fuzz_symbols_decode.py [rounds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from gr_amps_amd import capi, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
FULL = 41 + 7 + 7 * 240
for rd in range(rounds):
    # ---- (1) symbol seam
    C = int(rng.integers(1, 9))
    n = int(rng.integers(20000, 220000))
    streams = np.zeros((C, n), np.uint8)
    for c in range(C):
        s = rng.integers(0, 2, n).astype(np.uint8)
        off = int(rng.integers(0, 6000))
        while off < n - 200:
            _, _, _, _, words = synth.random_message(rng)
            bits = synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)
            keep = FULL if rng.random() < 0.7 else int(rng.integers(40, FULL))
            m = synth.manchester(bits[:keep])
            m = m[:max(0, n - off)]
            s[off:off + m.size] = m
            off += int(rng.choice([m.size + rng.integers(0, 300), rng.integers(500, 9000), 65536 - rng.integers(0, 4000)]))
        if rng.random() < 0.3:
            s[rng.integers(0, n, 5)] = rng.integers(2, 256, 5).astype(np.uint8)
        streams[c] = s
    refs = [oracle.Recc() for _ in range(C)]
    mode = rng.integers(0, 3)
    with capi.Recc(n_channels=C, max_bursts=max(4, C)) as r:
        off = 0
        while off < n:
            m = int(min(n - off, rng.integers(1, 61440) if mode == 0 else rng.integers(1, 5000) if mode == 1 else rng.choice([1, 73, 74, 75, 4096, 61439])))
            chunk = np.ascontiguousarray(streams[:, off:off + m])
            gb, gc = r.push_symbols(chunk)
            rb = [(c, b) for c in range(C) for b in [refs[c].work(chunk[c])] if b is not None]
            ok = len(gb) == len(rb) and all(int(gc[i]) == rb[i][0] and np.array_equal(gb[i], rb[i][1]) for i in range(len(rb)))
            if not ok:
                bad += 1
                print("SYMBOL MISMATCH round", rd, "offset", off, "chunk", m, len(gb), len(rb), flush=True)
                break
            off += m
    # ---- (2) decode
    nb = int(rng.integers(1, 200))
    bursts = np.zeros((nb, 3374), np.uint8)
    for i in range(nb):
        kind = rng.integers(0, 10)
        if kind == 0:
            b = rng.integers(0, 2, 3374).astype(np.uint8)                      # noise
        else:
            _, _, _, _, words = synth.random_message(rng)
            bits = np.array(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)[41:], np.uint8)
            for w in range(7):
                for rep in range(5):
                    nf = int(rng.choice([0, 0, 0, 1, 1, 2, 2, 3, 4, 6]))
                    pos = 7 + 240 * w + 48 * rep + rng.choice(48, nf, replace=False)
                    bits[pos] ^= 1
            b = synth.manchester(bits)[:3374]
            if kind == 1:
                b[rng.integers(0, 3374, int(rng.integers(1, 40)))] ^= 1       # broken Manchester pairs
            if kind == 2:
                b[rng.integers(0, 3374, 3)] = rng.integers(2, 256, 3).astype(np.uint8)
        bursts[i] = b
    chans = rng.integers(0, 1000, nb).astype(np.uint32)
    for majority in (False, True):
        with capi.Recc(n_channels=1, max_bursts=nb + 4, majority=majority) as r:
            got = r.decode_bursts(bursts, chans)
        want = oracle.decode_bursts(bursts, chans, majority=majority)
        if got.tobytes() != want.tobytes():
            bad += 1
            k = next(i for i in range(nb) if got[i].tobytes() != want[i].tobytes())
            f = [x for x in got.dtype.names if not np.array_equal(got[k][x], want[k][x])]
            print("DECODE MISMATCH round", rd, "majority", majority, "burst", k, "fields", f, flush=True)
print("%d rounds, %d mismatches, %.1f s" % (rounds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
