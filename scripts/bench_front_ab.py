"""Interleaved A/B of the IQ seam's streaming kernel: wave-private streams (AMPS_RECC_COOP=0) against the cooperative form
(AMPS_RECC_COOP=4), both handles alive in one process and pushed in turn (clock drift hits every variant alike)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi

spec = sys.argv[1] if len(sys.argv) > 1 else "sine"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,4".split(","))]
C, N = 832, 1 << 18
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.view_as_complex(torch.randn(C, N, 2, device="cuda", generator=g) * 0.5)
torch.cuda.synchronize()
hs = {}
for v in variants:
    os.environ["AMPS_RECC_COOP"] = str(v)
    hs[v] = capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=4096, time_kernels=True, slicer=spec)
for v, r in hs.items():
    for _ in range(10):
        r.push_iq(x)
        r.drain()
    r.timing(reset=True)
for _ in range(rounds):
    for v, r in hs.items():
        for _ in range(8):
            r.push_iq(x)
            r.drain()
for v, r in hs.items():
    t = r.timing()
    ms = t["ms_front"] / t["launches_front"]
    print("%-6s coop %2d  front %.4f ms  %.0f GB/s" % (spec, v, ms, 8.0 * C * N / ms / 1e6), flush=True)
    r.close()
