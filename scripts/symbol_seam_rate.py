"""Throughput of the exact drop-in seam (amps_recc_push_symbols = recc_impl::work on every channel) with device-resident symbols."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_amps_amd import capi, synth
rng = np.random.default_rng(0)
for C, n in ((1, 61439), (832, 4096), (832, 61439)):
    base = []
    for c in range(min(C, 8)):
        _, _, _, _, words = synth.random_message(rng)
        base.append(synth.symbol_stream(n, [(int(rng.integers(100, max(200, n - 4000))), synth.burst_bits(words, rng=rng))] if n > 8000 else [], rng))
    s = np.tile(np.stack(base), ((C + 7) // 8, 1))[:C]
    d = torch.from_numpy(s).to("cuda:0")
    with capi.Recc(n_channels=C, max_bursts=2 * C + 16, time_kernels=True) as r:
        for _ in range(3):
            r.push_symbols(d)
        r.timing(reset=True)
        t0 = time.perf_counter(); K = 10
        for _ in range(K):
            b, ch = r.push_symbols(d)
        el = (time.perf_counter() - t0) / K
        t = r.timing()
    print("%4d channels x %5d symbols per work(): %.3f ms per call (kernel %.3f ms) = %.1f Msym/s, %d bursts per call" %
          (C, n, el * 1e3, t["ms_symbols"] / K, C * n / el / 1e6, len(b)))
