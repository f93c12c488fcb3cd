#!/usr/bin/env python
"""Kernel time of the translate seam (xlate_fir_kernel) and of the tolerant-sync front kernel -- utility paths, not bench.py lines.
usage (GPU box): python scripts/bench_xlate.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
for C, n in ((1, 1 << 26), (64, 1 << 21)):
    x = torch.randn(C, n, 2, device=dev).mul_(0.5)
    x = torch.view_as_complex(x).contiguous()
    with capi.Recc(n_channels=C, sps=10, max_samples=n // 2, max_bursts=1024, time_kernels=True) as r:
        r.set_xlate(rate_hz=400e3, center_hz=160e3, decim=2)
        for _ in range(3):
            r.push_raw(x)
            r.drain()
        r.timing(reset=True)
        for _ in range(10):
            r.push_raw(x)
            r.drain()
        t = r.timing()
    ms = t["ms_xlate"] / 10
    print(f"xlate {C} ch x {n} samples @400k: {ms:.4f} ms/launch, {C * n * 8 / ms / 1e6:.1f} GB/s input, "
          f"{C * n / 2 * 299 * 4 / ms / 1e9:.1f} TFLOP/s (299 real taps x complex)")

iq = np.stack([synth.make_channel_block(1 << 18, 2, seed=c)[0] for c in range(16)])
d = torch.from_numpy(iq).to(dev).repeat(52, 1).contiguous()
for k in (0, 4):
    with capi.Recc(n_channels=832, sps=10, max_samples=1 << 18, max_bursts=8192, time_kernels=True, sync_tolerance=k) as r:
        for _ in range(3):
            r.push_iq(d)
            nrec = len(r.drain(copy=False))
        r.timing(reset=True)
        for _ in range(10):
            r.push_iq(d)
            r.drain(copy=False)
        t = r.timing()
    print(f"front kernel 832 x 2^18, sync_tolerance={k}: {t['ms_front'] / 10:.4f} ms/launch, {nrec} bursts/step")
