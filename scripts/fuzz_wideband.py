"""Wideband schedule fuzz at length: whatever the push schedule (ragged sizes, host or device blocks, sync / split / no
drains, fused or two-kernel form, exact or tolerant sync) the records equal those of one push with the same tolerance.
usage (GPU box): python scripts/fuzz_wideband.py [first_seed] [n_seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gr_amps_amd import capi, synth_wideband as sw

D = 512
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad, t0 = 0, time.time()
for seed in range(s0, s0 + nseeds):
    rng = np.random.default_rng(seed)
    first, C = int(rng.integers(0, 1024)), 832
    n = int(0.26 * sw.FS_WIDE) // D * D
    chans = rng.choice(C, size=6, replace=False)
    bursts = [((first + int(c)) % 1024, int(rng.integers(20000, 2200000))) for c in chans]
    x, truth = sw.make_wideband(n, bursts, seed=100 + seed)
    wb = {"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": first}

    def run(schedule, unfused, tol, resident, mode):
        with capi.Recc(n_channels=C, sps=3, max_samples=n // D + 72, max_bursts=64, unfused_wideband=unfused,
                       sync_tolerance=tol, wideband=wb) as r:
            off, recs, open_, keep = 0, [], False, []
            for m in schedule + [64 * D]:
                blk = x[off:off + m] if off < n else np.zeros(m, np.complex64)
                off += m
                if resident:
                    blk = torch.from_numpy(np.ascontiguousarray(blk)).to("cuda:0")
                    torch.cuda.synchronize()
                    keep.append(blk)
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
        return got[np.lexsort((got["position"], got["channel"]))]

    ref = {0: run([n], False, 0, False, "sync"), 3: run([n], False, 3, False, "sync")}   # a tolerant trigger may shift a position by one sample
    ok_truth = len(ref[0]) == len(bursts)
    for trial in range(5):
        cuts = np.sort(rng.integers(1, n, size=int(rng.integers(1, 6))))
        schedule = [int(b - a) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n]) if b > a]
        cfg = (bool(rng.integers(0, 2)), int(rng.choice([0, 3])), bool(rng.integers(0, 2)), str(rng.choice(["sync", "split", "none"])))
        got = run(schedule, *cfg)
        if got.tobytes() != ref[cfg[1]].tobytes():
            bad += 1
            print("MISMATCH seed", seed, "trial", trial, "schedule", schedule, "unfused/tol/resident/mode", cfg, "first", first,
                  "n", len(got), [(int(a["channel"]), int(a["position"])) for a in got], [(int(a["channel"]), int(a["position"])) for a in ref[cfg[1]]], flush=True)
    if not ok_truth:
        print("seed", seed, "found", len(ref[0]), "of", len(bursts), "bursts", bursts, flush=True)
print("%d seeds, %d mismatches, %.1f s" % (nseeds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
