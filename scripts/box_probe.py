"""What is different between two boxes that run the SAME library's streaming kernel 8-11 % apart?    Per box, in one
process: the streaming kernel of the IQ seam (spec D, 832 x 2^18, HIP events), a plain read of the same 1.74 GB (torch.sum: rocPRIM's
reduce) and a plain copy of it (torch clone), each the best and the mean of 30; and the clocks rocm-smi reports under that load.
usage (GPU box): python scripts/box_probe.py <tag>"""
import json
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi

tag = sys.argv[1] if len(sys.argv) > 1 else "box"
C, N = 832, 1 << 18
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.view_as_complex(torch.randn(C, N, 2, device="cuda", generator=g) * 0.5)
xr = torch.view_as_real(x)
nbytes = xr.numel() * 4
torch.cuda.synchronize()


def timed(fn, reps=30, warm=10):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    return min(ms), sum(ms) / len(ms)


r = capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=4096, time_kernels=True, slicer="exact")
for _ in range(40):
    r.push_iq(x)
    r.drain()
r.timing(reset=True)
for _ in range(60):
    r.push_iq(x)
    r.drain()
t = r.timing()
front = t["ms_front"] / t["launches_front"]
smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
try:
    card = next(iter(json.loads(smi).values()))
    keep = {k: v for k, v in card.items() if any(s in k for s in ("clk", "Power", "Temperature (Sensor memory)", "Temperature (Sensor junction)"))}
except Exception:
    keep = {}
r.close()
rd = timed(lambda: xr.sum())
cp = timed(lambda: xr.clone())
out = {"tag": tag, "front_ms": round(front, 4), "front_TBps": round(nbytes / front / 1e9, 3),
       "read_sum_ms_best_mean": [round(v, 4) for v in rd], "read_TBps_best": round(nbytes / rd[0] / 1e9, 3),
       "copy_ms_best_mean": [round(v, 4) for v in cp], "copy_TBps_best_read_plus_write": round(2 * nbytes / cp[0] / 1e9, 3),
       "front_over_read_best": round(front / rd[0], 3), "smi_idle_after": keep}
print(json.dumps(out))
