"""Where the resolve + capture + decode kernel spends its time (a -DRESOLVE_TIMELINE build of the library): s_memtime stamps of
thread 0 of every workgroup of the last launch.  usage (GPU box):
AMPS_RECC_LIB=scripts/variants/rtl.so python scripts/resolve_timeline.py [every]   (one burst in every `every`-th channel)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AMPS_RECC_RESOLVE_TIMELINE"] = "/tmp/resolve_tl.bin"
import torch

import bench
from gr_amps_amd import capi

every = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
decim = int(os.environ.get("CHZ_DECIM", "768"))
NW = (1 << 27) if decim == 512 else 11 * 256 * 64 * 768
x, planted = bench.make_wideband_batch(torch, dev, NW, 96, 832, every, seed=3)
with capi.Recc(n_channels=832, sps=1536 // decim, max_samples=NW // decim + 72, max_bursts=8192, sync_torch=False,
               wideband={"channels": 1024, "decim": decim, "taps_per_branch": 8, "first_channel": 96}) as r:
    torch.cuda.synchronize()
    for _ in range(4):
        r.push_wideband(x)
        n = len(r.drain())
tl = np.fromfile("/tmp/resolve_tl.bin", dtype=np.uint64).reshape(-1, 24).astype(np.int64)
t0 = tl[:, 0].min()
names = ["entry", "counts scanned", "hits walked", "decoded", "past barrier", "stored", "state written", "done counted"]
print("records %d; s_memtime ticks since the workgroup's own entry (the counters of the eight XCDs are not aligned)" % n)
has = tl[:, 3] > 0
for label, sel in (("workgroups with a burst", has), ("workgroups without", ~has)):
    if not sel.any():
        continue
    print("%s: %d" % (label, sel.sum()))
    for k in range(8):
        ok = tl[sel, k] > 0
        v = (tl[sel, k] - tl[sel, 0])[ok]
        if v.size:
            print("  %-16s min %6d  median %6d  max %6d" % (names[k], v.min(), np.median(v), v.max()))
dn = ["manchester done", "bch", "valid + raw copy", "word_dec copy", "flip + dcc", "pack", "parse", "(store)", "ring words in LDS"]
d = tl[has, 8:17] - tl[has, 0:1]
order = [8, 0, 1, 2, 3, 4, 5, 6]
print("decode stages of thread 0's wave, ticks since the workgroup's entry (median over workgroups):")
for k in order:
    v = d[:, k][tl[has, 8 + k] > 0]
    if v.size:
        print("  %-18s %6d" % (dn[k], np.median(v)))
print("entry stamps by dispatch order (every 64th workgroup):", (tl[::64, 0] - t0).tolist())
print("last stamp of all:", int(tl[:, 7].max() - t0))
