"""Per-wave entry / exit times of the bit-domain correlator (a -DBITS_TIMELINE build): is the kernel as long as its waves, or as
long as its dispatch?  usage (GPU box): AMPS_RECC_LIB=scripts/variants/btl.so python scripts/bits_timeline.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AMPS_RECC_BITS_TIMELINE"] = "/tmp/bits_tl.bin"
import torch

import bench
from gr_amps_amd import capi

dev = torch.device("cuda:0")
NW = 1 << 27
x, planted = bench.make_wideband_batch(torch, dev, NW, 96, 832, 2, seed=3)
with capi.Recc(n_channels=832, sps=3, max_samples=NW // 512 + 8, max_bursts=8192, sync_torch=False,
               wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 96}) as r:
    torch.cuda.synchronize()
    for _ in range(4):
        r.push_wideband(x)
        n = len(r.drain())
raw = np.fromfile("/tmp/bits_tl.bin", dtype=np.uint64)
nw = int(raw[0])
tl = raw[1:1 + 3 * min(nw, 16384)].reshape(-1, 3).astype(np.int64)
print("waves %d, records %d" % (nw, n))
dur = tl[:, 1] - tl[:, 0]
print("wave life (ticks): min %d median %d p90 %d max %d" % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
for xcc in sorted(set(tl[:, 2].tolist())):
    sel = tl[:, 2] == xcc
    t0 = tl[sel, 0].min()
    st = tl[sel, 0] - t0
    en = tl[sel, 1] - t0
    print("XCC %d: %5d waves | entry: median %6d p90 %6d max %6d | exit: median %6d max %6d" % (xcc, sel.sum(), np.median(st), np.percentile(st, 90), st.max(), np.median(en), en.max()))
