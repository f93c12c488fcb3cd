import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle
from oracle import channelizer as cz
from gr_amps_amd import capi, synth_wideband as sw
oracle.build()
D = 768
def H(C, first, max_frames, **kw):
    return capi.Recc(n_channels=C, sps=2, max_samples=max_frames, max_bursts=256,
                     wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}, **kw)
# 1. filter bank vs numpy
rng = np.random.default_rng(1)
n = 200 * D
t = np.arange(n)
x = 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
for k, a in ((3, 1.0), (100, 0.5), (511, 0.7), (900, 0.3)):
    x += a * np.exp(2j * np.pi * (sw.bin_freq(k) + 5e3) * t / sw.FS_WIDE)
x = x.astype(np.complex64)
with H(1024, 0, n // D + 8) as r:
    got = r.debug_channelize(x)
want = cz.channelize(x, P=8, D=D)
print("shape", got.shape, want.shape)
err = np.abs(got - want).max() / np.abs(want).max()
print("numpy model err", err)
# 2. streaming == one shot
rng = np.random.default_rng(2)
n = 96 * D + 77
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
with H(832, 96, 200) as r:
    one = r.debug_channelize(x)
with H(832, 96, 200) as r:
    parts, off = [], 0
    for m in (1, 511, 512, 513, 5000, 12345, n):
        m = min(m, n - off)
        if m <= 0: break
        parts.append(r.debug_channelize(x[off:off + m])); off += m
    many = np.concatenate(parts, axis=1)
print("stream", one.shape, many.shape, np.array_equal(one.view(np.uint32), many.view(np.uint32)))
# 3. bursts decode, fused == model on own output
first, C = 96, 832
n = int(0.2 * sw.FS_WIDE) // D * D
bursts = [(first + 4, 200000), (first + 5, 250000), (first + 6, 300000), (first + 700, 100000), (first + 831, 400000), (first + 0, 50000)]
x, truth = sw.make_wideband(n, bursts, seed=3)
with H(C, first, n // D + 8) as r:
    chan = r.debug_channelize(x)
for slicer in ("exact", "atan", "product", "sine"):
    with H(C, first, n // D + 8, slicer=slicer) as r:
        half = (n // 2) // D * D + 100
        r.push_wideband(x[:half]); r.push_wideband(x[half:])
        got = r.drain()
    by_chan = {int(g["channel"]): g for g in got}
    okw = 0
    for (k, off), (kind, min10, esn, dialed, words) in truth.items():
        g = by_chan.get(k - first)
        if g is not None and g["min"].decode() == min10 and g["valid"].all() and all(list(g["word_raw"][w][:36]) == list(b) for w, b in enumerate(words)): okw += 1
    active = sorted(set(k - first for k, _ in bursts))
    want = oracle.fused_push_all(chan[active], sps=2, slicer={"atan":0,"product":1,"sine":2,"exact":3}[slicer])
    want["channel"] = np.array(active, np.uint32)[want["channel"]]
    print(slicer, "records", len(got), "truth ok", okw, "== model", got.tobytes() == want.tobytes(), len(want))
    if got.tobytes() != want.tobytes() and len(got) == len(want):
        for a, b in zip(got, want):
            for f in a.dtype.names:
                if not np.array_equal(a[f], b[f]): print("  diff", int(a["channel"]), f, a[f] if a[f].size < 10 else "...", b[f] if b[f].size < 10 else "...")
