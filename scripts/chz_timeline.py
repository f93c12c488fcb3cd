"""Per-role phase timeline of chz12_kernel (a -DCHZ_TIMELINE build of the library): mean s_memtime cycles between the phase
boundaries of workgroup 0's twelve waves.  usage (GPU box): AMPS_RECC_LIB=scripts/variants/tl.so python scripts/chz_timeline.py [spec]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AMPS_RECC_CHZ_TIMELINE"] = "/tmp/chz_tl.bin"
from gr_amps_amd import capi

spec = sys.argv[1] if len(sys.argv) > 1 else "sine"
NW = 1 << 27
decim = int(os.environ.get("CHZ_DECIM", "512"))
x = torch.view_as_complex(torch.randn(NW, 2, device="cuda") * 0.5)
torch.cuda.synchronize()
r = capi.Recc(n_channels=832, sps=1536 // decim, max_samples=NW // decim + 72, max_bursts=4096, slicer=spec,
              wideband={"channels": 1024, "decim": decim, "taps_per_branch": 8, "first_channel": 96})
for _ in range(6):
    r.push_wideband(x)
    r.drain()
tl = np.fromfile("/tmp/chz_tl.bin", dtype=np.uint64).reshape(12, 8).astype(np.float64)
names = {0: "fold", 1: "pass2", 2: "p3+slicer"}
print("decim %d spec %s, tag %s" % (decim, spec, os.environ.get("AMPS_RECC_LIB", "").split("/")[-1]))
for w in range(12):
    role = 2 - (w >> 2)
    n = max(tl[w, 5], 1.0)
    a = tl[w, :5] / n        # a[k] = mean cycles from the previous stamp to stamp k
    tot = a.sum()
    if role == 0:
        print("wave %2d %-9s step %5.0f | start %4.0f  wait-loads %5.0f  fold %5.0f  issue-loads %5.0f  at-barrier %5.0f" % (w, names[role], tot, a[0], a[1], a[2], a[3], a[4]))
    elif role == 1:
        print("wave %2d %-9s step %5.0f | start %4.0f  pass2 %5.0f  (pass3 %5.0f)  at-barrier %5.0f" % (w, names[role], tot, a[0], a[1], a[3], a[4]))
    else:
        print("wave %2d %-9s step %5.0f | start %4.0f  slicer %5.0f  pass3 %5.0f  at-barrier %5.0f" % (w, names[role], tot, a[0], a[1], a[3], a[4]))
