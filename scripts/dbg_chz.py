import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import channelizer as cz
from gr_amps_amd import capi, synth_wideband as sw
D = 512
rng = np.random.default_rng(1)
n = 200 * D
t = np.arange(n)
x = 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
for k, a in ((3, 1.0), (100, 0.5), (511, 0.7), (900, 0.3)):
    x += a * np.exp(2j * np.pi * (sw.bin_freq(k) + 5e3) * t / sw.FS_WIDE)
x = x.astype(np.complex64)
with capi.Recc(n_channels=1024, sps=3, max_samples=n // D + 8, max_bursts=256, wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 0}) as r:
    got = r.debug_channelize(x)
want = cz.channelize(x, P=8)
scale = np.abs(want).max()
err = np.abs(got - want) / scale
print("shape", got.shape, "max err", err.max())
bad_frames = np.nonzero(err.max(0) > 1e-4)[0]
bad_bins = np.nonzero(err.max(1) > 1e-4)[0]
print("bad frames", bad_frames[:40], len(bad_frames))
print("bad bins", bad_bins[:40], len(bad_bins))
