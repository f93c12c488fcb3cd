"""Kernel times of a wideband832 step with exact and tolerant sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gr_amps_amd import capi
dev = torch.device("cuda", 0)
NW = 1 << 27
batch, nb = bench.make_wideband_batch(torch, dev, NW, 96, 832, 2, seed=1)
for tol in (0, 4):
    with capi.Recc(n_channels=832, sps=3, max_samples=NW // 512 + 8, max_bursts=4096, time_kernels=True, sync_tolerance=tol,
                   wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 96}) as r:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            r.push_wideband(batch); r.drain(copy=False)
        r.timing(reset=True)
        for _ in range(10):
            r.push_wideband(batch); n = len(r.drain(copy=False))
        t = r.timing()
    print("sync_tolerance=%d: bit-domain correlator %.4f ms, channelizer %.4f ms, %d bursts" % (tol, t["ms_front"] / 10, t["ms_channelizer"] / 10, n))
