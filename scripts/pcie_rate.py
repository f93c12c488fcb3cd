"""PCIe-inclusive throughput: the same workloads as bench.py but with HOST-resident input (pageable numpy, and page-locked
torch tensors), i.e. the boundary as a GNU Radio block would use it.  usage (GPU box): python scripts/pcie_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_amps_amd import capi
dev = torch.device("cuda", 0)
# wideband832: 2^25 wideband samples per push (256 MiB, 1.1 s of signal)
NW = 1 << 25
x, nb = bench.make_wideband_batch(torch, dev, NW, 96, 832, 2, seed=1)
host = x.cpu()
for name, blk in (("pageable", host.numpy()), ("page-locked", host.pin_memory())):
    with capi.Recc(n_channels=832, sps=3, max_samples=NW // 512 + 72, max_bursts=4096,
                   wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 96}) as r:
        for _ in range(2):
            r.push_wideband(blk); r.drain(copy=False)
        t0 = time.perf_counter(); K = 5
        for _ in range(K):
            r.push_wideband(blk); n = len(r.drain(copy=False))
        el = (time.perf_counter() - t0) / K
    print("wideband832 host %-11s: %.2f ms per 256 MiB push = %.1f GB/s = %.2f Gsym/s (%d bursts)" % (name, el * 1e3, NW * 8 / el / 1e9, 832 * NW / 1536 / el / 1e9, n))
# direct832: 832 x 2^15 samples per push (218 MB)
C, N = 832, 1 << 15
batch, iq_base, expected = bench.make_batch(torch, dev, C, N, 10, seed=1)
host = batch.cpu()
for name, blk in (("pageable", host.numpy()), ("page-locked", host.pin_memory())):
    with capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=8192) as r:
        for _ in range(2):
            r.push_iq(blk); r.drain(copy=False)
        t0 = time.perf_counter(); K = 5
        for _ in range(K):
            r.push_iq(blk); n = len(r.drain(copy=False))
        el = (time.perf_counter() - t0) / K
    print("direct832   host %-11s: %.2f ms per %d MB push = %.1f GB/s = %.2f Gsym/s" % (name, el * 1e3, C * N * 8 // 1000000, C * N * 8 / el / 1e9, C * N / 10 / el / 1e9))
