"""Time of the kernels behind the streaming / filter-bank kernel (bit-domain correlator, resolve + capture + decode) against the
number of bursts in a push: base cost (noise only) and the cost per decoded burst.  usage (GPU box): python scripts/ubench_tail.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from gr_amps_amd import capi, synth

dev = torch.device("cuda:0")
REP = 30


def run(r, push, label):
    for _ in range(5):
        push(); r.drain(copy=False)
    r.timing(reset=True)
    n = 0
    for _ in range(REP):
        push(); n += len(r.drain(copy=False))
    t = r.timing()
    print("%-28s records/push %6.1f | front/bits %.4f resolve+decode %.4f decode %.4f carry %.4f chz %.4f ms"
          % (label, n / REP, t["ms_front"] / REP, t["ms_resolve"] / REP, t["ms_decode"] / REP, t["ms_carry"] / REP, t["ms_channelizer"] / REP), flush=True)


# channel-major IQ seam, 832 x 2^18 @ sps 10
C, N = 832, 1 << 18
for nb in (() if "wide" in sys.argv[1:] else (0, 1, 2)):
    xs = [synth.make_channel_block(N, nb, seed=1000 + c, sps=10)[0] for c in range(16)]
    d = torch.from_numpy(np.stack(xs)).to(dev).repeat(52, 1)[:C].contiguous()
    with capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=8192, time_kernels=True, sync_torch=False) as r:
        torch.cuda.synchronize()
        run(r, lambda: r.push_iq(d), "iq832 bursts/channel=%d" % nb)
# wideband seam, 2^27 samples
DEC = int(os.environ.get("CHZ_DECIM", "512"))            # the filter bank's decimation: 512 (3 samples per symbol) or 768 (2)
NW = (1 << 27) if DEC == 512 else 11 * 256 * 64 * 768
for every in (0, 2, 1):
    x, planted = bench.make_wideband_batch(torch, dev, NW, 96, 832 if every else 0, every or 1, seed=3)
    with capi.Recc(n_channels=832, sps=1536 // DEC, max_samples=NW // DEC + 72, max_bursts=8192, time_kernels=True, sync_torch=False,
                   wideband={"channels": 1024, "decim": DEC, "taps_per_branch": 8, "first_channel": 96}) as r:
        torch.cuda.synchronize()
        run(r, lambda: r.push_wideband(x), "wide832 bursts=%d" % len(planted))
    del x
