"""Is the filter-bank kernel's sustained time set by the package power limit?  The same kernel, same process: launched back to back
(the bench's stream: the package settles at its 1400 W limit and the clock with it) and with the GPU left idle between launches
(every launch starts from the boost clock).  ms per 1 GiB push, HIP events around the kernel alone.
usage (GPU box): python scripts/power_wall.py [idle_ms]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi

idle = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
NW = 1 << 27
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.view_as_complex(torch.randn(NW, 2, device="cuda", generator=g) * 0.5)
torch.cuda.synchronize()
wb = {"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 96}
r = capi.Recc(n_channels=832, sps=3, max_samples=NW // 512 + 8, max_bursts=4096, time_kernels=True, wideband=wb)


def run(n, gap_ms):
    r.timing(reset=True)
    for _ in range(n):
        if gap_ms:
            time.sleep(gap_ms * 1e-3)
        r.push_wideband(x)
        r.drain()
    t = r.timing()
    return t["ms_channelizer"] / t["launches_channelizer"]


for rnd in range(3):
    for _ in range(1500):                 # ~0.7 s of back-to-back load first
        r.push_wideband(x)
    r.drain()
    print("back to back (after 1500 launches): %.4f ms" % run(200, 0), flush=True)
    time.sleep(0.5)
    print("GPU idle %.0f ms before every launch: %.4f ms" % (idle, run(40, idle)), flush=True)
r.close()
