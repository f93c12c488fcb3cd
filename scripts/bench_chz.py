"""Microbench of the wideband seam's filter-bank kernel: ms per 1 GiB push for each slicer spec (HIP events).
usage (GPU box): [AMPS_RECC_LIB=variant.so] python scripts/bench_chz.py [reps]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
specs = sys.argv[2].split(",") if len(sys.argv) > 2 else ("atan", "sine", "product", "exact")
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 40
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # > 1: time ONE channel group of that many (a rank of the one-band multi-GPU split)
decim = int(os.environ.get("CHZ_DECIM", "512"))             # 512 (3 samples per symbol) or 768 (2)
NW = 1 << 27
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.view_as_complex(torch.randn(NW, 2, device="cuda", generator=g) * 0.5)
torch.cuda.synchronize()
wb = {"channels": 1024, "decim": decim, "taps_per_branch": 8, "first_channel": 96}
if groups > 1:
    wb.update(groups=groups, group=groups - 1)
for spec in specs:
    r = capi.Recc(n_channels=832, sps=1536 // decim, max_samples=NW // decim + 72, max_bursts=4096, time_kernels=True, wideband=wb, slicer=spec)
    for _ in range(warm):
        r.push_wideband(x)
        r.drain()
    r.timing(reset=True)
    for _ in range(reps):
        r.push_wideband(x)
        r.drain()
    t = r.timing()
    print("%-8s chz %.4f ms  bits %.4f  resolve %.4f  decode %.4f" % (
        spec, t["ms_channelizer"] / t["launches_channelizer"], t["ms_front"] / max(1, t["launches_front"]),
        t["ms_resolve"] / reps, t["ms_decode"] / reps), flush=True)
    r.close()
