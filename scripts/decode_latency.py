"""Latency vs throughput of the burst decode kernel (amps_recc_decode_bursts): ms per launch for 1 .. 16384 bursts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gr_amps_amd import capi, synth
rng = np.random.default_rng(1)
_, _, _, _, words = synth.random_message(rng)
burst = synth.manchester(synth.burst_bits(words, rng=rng))[74:74 + 3374].astype(np.uint8)
for n in (1, 64, 512, 1664, 16384):
    b = np.tile(burst, (n, 1))
    import torch
    d = torch.from_numpy(b).to("cuda:0")
    with capi.Recc(n_channels=1, sps=10, max_samples=0, max_bursts=max(n, 16), time_kernels=True) as r:
        for _ in range(3):
            r.decode_bursts(d)
        r.timing(reset=True)
        for _ in range(10):
            out = r.decode_bursts(d)
        t = r.timing()
    print(n, "bursts: %.4f ms per launch" % (t["ms_decode"] / 10), "valid", int(out["valid"][:, 0].sum()))
