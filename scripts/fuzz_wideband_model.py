"""Wideband seam against the CPU model at length, at either decimation: random band position, six bursts at a random C/N (9 .. 30 dB)
with a random symbol-clock / carrier offset, a random slicer spec, exact or tolerant sync, tracked or fixed timing, a random push
schedule (host or device blocks, sync / split / no drains) -- the records must equal, byte for byte, those of the CPU model
(oracle.fused_push_all) run on the filter bank's own channel-major output (amps_recc_debug_channelize).
usage (GPU box): python scripts/fuzz_wideband_model.py [first_seed] [n_seeds] [decim]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle
import widebandref as W
from gr_amps_amd import capi

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D = int(sys.argv[3]) if len(sys.argv) > 3 else 768
sps = 1536 // D
oracle.build()
dev = torch.device("cuda:0")
bad, nrec, t0 = 0, 0, time.time()
for seed in range(s0, s0 + nseeds):
    rng = np.random.default_rng(seed)
    first, C = int(rng.integers(0, 1024)), 832
    n = int(0.3 * W.FS) // 1536 * 1536
    chans = sorted(int(c) for c in rng.choice(C, size=6, replace=False))
    snr = float(rng.uniform(9, 30))
    ppm = float(rng.choice([0.0, 0.0, 100.0, 400.0, 1000.0]))
    cfo = float(rng.choice([0.0, 0.0, 1000.0, 2000.0]))
    spec = str(rng.choice(["exact", "exact", "atan", "sine", "product"]))
    sid = {"atan": 0, "product": 1, "sine": 2, "exact": 3}[spec]
    tol = int(rng.choice([0, 0, 2]))
    fixed = bool(rng.integers(0, 4) == 0)
    x, planted = W.make_block(torch, dev, n, chans, first, ppm, cfo, snr, seed=5000 + seed)
    wb = {"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}
    with capi.Recc(n_channels=C, sps=sps, max_samples=n // D + 136, max_bursts=64, wideband=wb) as r:
        chan = r.debug_channelize(torch.cat([x, torch.zeros(64 * D, dtype=torch.complex64, device=dev)]))   # with the silence that flushes the seam below
    want = oracle.fused_push_all(chan[chans], sps=sps, tolerance=tol, slicer=sid, tracking=not fixed)
    want["channel"] = np.array(chans, np.uint32)[want["channel"]]
    cuts = np.sort(rng.integers(1, n, size=int(rng.integers(0, 5))))
    schedule = [int(b - a) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n]) if b > a]
    resident, mode = bool(rng.integers(0, 2)), str(rng.choice(["sync", "split", "none"]))
    xh = x.cpu().numpy()
    with capi.Recc(n_channels=C, sps=sps, max_samples=n // D + 72, max_bursts=64, wideband=wb, slicer=spec, sync_tolerance=tol, fixed_timing=fixed) as r:
        off, recs, open_, keep = 0, [], False, []
        for m in schedule + [64 * D]:
            blk = x[off:off + m] if (resident and off < n) else (xh[off:off + m] if off < n else np.zeros(m, np.complex64))
            off += m
            r.push_wideband(blk)
            if mode == "sync":
                recs.append(r.drain())
            elif mode == "split":
                if open_:
                    recs.append(r.drain_end())
                r.drain_begin()
                open_ = True
        recs.append(r.drain_end() if open_ else r.drain())
        got = np.concatenate(recs)
    got = got[np.lexsort((got["position"], got["channel"]))]
    # the model saw only the planted channels: records of other channels (false triggers on noise) would be a finding of their own
    nrec += len(got)
    if got.tobytes() != want.tobytes():
        bad += 1
        print("MISMATCH seed", seed, dict(D=D, snr=round(snr, 1), ppm=ppm, cfo=cfo, spec=spec, tol=tol, fixed=fixed, schedule=schedule, resident=resident, mode=mode),
              [(int(a["channel"]), int(a["position"])) for a in got], [(int(a["channel"]), int(a["position"])) for a in want], flush=True)
print("decim %d: %d seeds, %d records, %d mismatches, %.1f s" % (D, nseeds, nrec, bad, time.time() - t0))
sys.exit(1 if bad else 0)
