"""Does the streaming kernel's time depend on WHERE its input lies?  One process, one box: the same 1.74 GB block re-allocated behind pads
of different sizes (so that its virtual / physical placement changes), the kernel timed on each copy; then the same handle and block
timed three more times without re-allocating.  usage (GPU box): python scripts/box_probe2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi

C, N = 832, 1 << 18
g = torch.Generator(device="cuda")
g.manual_seed(1)
src = torch.view_as_complex(torch.randn(C, N, 2, device="cuda", generator=g) * 0.5)
torch.cuda.synchronize()


def front_ms(r, x, reps=40):
    for _ in range(20):
        r.push_iq(x)
        r.drain()
    r.timing(reset=True)
    for _ in range(reps):
        r.push_iq(x)
        r.drain()
    t = r.timing()
    return t["ms_front"] / t["launches_front"]


r = capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=4096, time_kernels=True, slicer="exact")
pads = []
for k, pad_mib in enumerate((0, 3, 64 + 1, 512 + 7, 1024 + 13, 2048 + 2, 5, 4096 + 9)):
    pads.append(torch.empty(pad_mib << 20 or 1, dtype=torch.uint8, device="cuda"))
    x = src.clone()
    torch.cuda.synchronize()
    print("pad %5d MiB  block at 0x%012x (mod 2 MiB: %7d, mod 1 GiB: %4d MiB)  front %.4f ms" % (
        pad_mib, x.data_ptr(), x.data_ptr() % (2 << 20), (x.data_ptr() % (1 << 30)) >> 20, front_ms(r, x)), flush=True)
    keep = x
for k in range(3):
    print("same block again                                   front %.4f ms" % front_ms(r, keep), flush=True)
r.close()
# a fresh handle on the last block
r = capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=4096, time_kernels=True, slicer="exact")
print("fresh handle, same block                           front %.4f ms" % front_ms(r, keep), flush=True)
r.close()
