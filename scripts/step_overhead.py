"""Where a wideband832 step's wall time goes beyond its kernels: push (launch) time, drain time, with/without HIP-event timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gr_amps_amd import capi
dev = torch.device("cuda", 0)
NW = 1 << 27
batch, nb = bench.make_wideband_batch(torch, dev, NW, 96, 832, 2, seed=1)
for tk in (False, True):
    r = capi.Recc(n_channels=832, sps=3, max_samples=NW // 512 + 8, max_bursts=4096, time_kernels=tk,
                  wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 96})
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        r.push_wideband(batch); r.drain(copy=False)
    K = 50
    tp = td = 0.0
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for _ in range(K):
        t0 = time.perf_counter(); r.push_wideband(batch); t1 = time.perf_counter()
        recs = r.drain(copy=False); t2 = time.perf_counter()
        tp += t1 - t0; td += t2 - t1
    tot = time.perf_counter() - ta
    print("time_kernels=%s: step %.1f us  (push call %.1f us, drain call %.1f us, %d records)" % (tk, tot / K * 1e6, tp / K * 1e6, td / K * 1e6, len(recs)))
    if tk:
        t = r.timing()
        print("   kernels per step: %.1f us" % (sum(t[k] for k in ("ms_front", "ms_resolve", "ms_decode", "ms_carry", "ms_channelizer")) / (K) * 1e3))
    r.close()
