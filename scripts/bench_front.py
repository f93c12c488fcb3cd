"""Microbench of the IQ seam's streaming kernel: ms per 832 x 2^18 push for each slicer spec (HIP events)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
C, N = 832, 1 << 18
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.view_as_complex(torch.randn(C, N, 2, device="cuda", generator=g) * 0.5)
torch.cuda.synchronize()
for spec in ("atan", "sine", "product", "exact"):
    r = capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=4096, time_kernels=True, slicer=spec)
    for _ in range(40):
        r.push_iq(x)
        r.drain()
    r.timing(reset=True)
    for _ in range(reps):
        r.push_iq(x)
        r.drain()
    t = r.timing()
    ms = t["ms_front"] / t["launches_front"]
    print("%-8s front %.4f ms  %.0f GB/s" % (spec, ms, 8.0 * C * N / ms / 1e6), flush=True)
    r.close()
