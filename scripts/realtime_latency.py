"""Real-time use of one handle ("a 1-channel real-time-latency run"): host-resident IQ arrives in 20 ms blocks
(4000 samples per channel at 200 ksps); latency = push (H2D staging + kernels) + drain, per block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gr_amps_amd import capi, synth
for C in (1, 64, 832):
    n = 4000
    iq = np.stack([synth.make_channel_block(25 * n, 2, seed=c)[0] for c in range(min(C, 8))])
    iq = np.tile(iq, ((C + 7) // 8, 1))[:C]
    with capi.Recc(n_channels=C, sps=10, max_samples=n, max_bursts=max(64, 4 * C)) as r:
        lat, nrec = [], 0
        for rep in range(3):
            for k in range(25):
                blk = np.ascontiguousarray(iq[:, k * n:(k + 1) * n])
                t0 = time.perf_counter()
                r.push_iq(blk)
                recs = r.drain(copy=False)
                lat.append(time.perf_counter() - t0)
                nrec += len(recs)
        lat = np.array(lat[25:]) * 1e6
    print("%4d channels x 20 ms blocks: median %.0f us, p99 %.0f us per block (%.1f %% of real time), %d bursts" %
          (C, np.median(lat), np.percentile(lat, 99), np.median(lat) / 20000 * 100, nrec))
