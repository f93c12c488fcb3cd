import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_amps_amd import capi
dev = torch.device("cuda", 0)
C, N = 832, 1 << 18
batch, iq_base, expected = bench.make_batch(torch, dev, C, N, 10, seed=1)
r = capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=max(4096, 2 * expected), time_kernels=False)
for _ in range(3):
    r.push_iq(batch); r.drain()
tp = td = ts = 0.0
K = 20
for _ in range(K):
    t0 = time.perf_counter(); r.push_iq(batch); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    # library stream is non-blocking wrt torch: sync through the library instead
    recs = r.drain(); t3 = time.perf_counter()
    tp += t1 - t0; ts += t2 - t1; td += t3 - t2
print("push %.1f us  torch-sync %.1f us  drain %.1f us  (n=%d)" % (tp / K * 1e6, ts / K * 1e6, td / K * 1e6, len(recs)))
# drain with no records pending (pure sync + header copy)
t0 = time.perf_counter()
for _ in range(K): r.drain()
print("empty drain %.1f us" % ((time.perf_counter() - t0) / K * 1e6))
