"""resolve + capture + decode per push with AMPS_RECC_FLAG_KEEP_BURSTS (what the host blocks set): channel-major seam, 832 x 2^18, 2 bursts per channel.
usage (GPU box): [AMPS_RECC_LIB=variant.so] python scripts/ubench_keep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi, synth

C, N, REP = 832, 1 << 18, 30
xs = [synth.make_channel_block(N, 2, seed=1000 + c, sps=10)[0] for c in range(16)]
d = torch.from_numpy(np.stack(xs)).to("cuda").repeat(52, 1)[:C].contiguous()
torch.cuda.synchronize()
for keep in (False, True):
    with capi.Recc(n_channels=C, sps=10, max_samples=N, max_bursts=8192, time_kernels=True, sync_torch=False, keep_bursts=keep) as r:
        for _ in range(5):
            r.push_iq(d)
            (r.drain_bursts() if keep else r.drain(copy=False))
        r.timing(reset=True)
        n = 0
        import time
        t0 = time.perf_counter()
        for _ in range(REP):
            r.push_iq(d)
            n += len((r.drain_bursts()[0] if keep else r.drain(copy=False)))
        wall = (time.perf_counter() - t0) / REP * 1e3
        t = r.timing()
        print("keep_bursts=%-5s records/push %6.1f resolve+capture+decode %.4f ms   push + drain, host clock %.3f ms" % (keep, n / REP, t["ms_resolve"] / REP, wall), flush=True)
