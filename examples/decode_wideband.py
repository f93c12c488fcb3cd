"""decode_wideband.py -- the headline path in twenty lines: one 30.72 Msps complex stream in, decoded RECC seizure bursts of the whole
AMPS band out (BASELINE configs[3]; replaces 832 x [freq_xlating_fir_filter_ccc -> quadrature_demod -> clock_recovery_mm -> binary_slicer
-> amps.recc -> amps.recc_decode] of grc/recctest.grc).  Needs an MI355X: the library has no CPU path.

    python examples/decode_wideband.py            # 12 mobiles on random channels, the stream pushed in ragged blocks
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_amps_amd import capi, synth_wideband as sw

FIRST_BIN, CHANNELS = 96, 832                     # the band selection: FFT bins 96 .. 927 of the 1024-branch filter bank
NSAMP = 12 * (1 << 20)                            # 0.41 s of signal
rng = np.random.default_rng(7)
plan = [(FIRST_BIN + int(c), int(rng.integers(4000, NSAMP - 3456 * 1536 - 4000))) for c in rng.choice(CHANNELS, 12, replace=False)]
x, truth = sw.make_wideband(NSAMP, plan, seed=7, snr_db=20.0, sym_ppm=50.0)   # every mobile 50 ppm off the nominal bit clock

try:
    rx = capi.Recc(n_channels=CHANNELS, max_samples=NSAMP // 512 + 72, max_bursts=256,
                   wideband={"channels": 1024, "first_channel": FIRST_BIN})     # decimation: the library default (768: 40 ksps per channel)
except capi.AmpsError as e:
    sys.exit("no MI355X here (%s): the library has no CPU fallback" % e)
with rx:
    pos = 0
    while pos < NSAMP:                            # blocks of any size: what is left of a frame waits in the handle for the next push
        n = min(int(rng.integers(100_000, 3_000_000)), NSAMP - pos)
        rx.push_wideband(x[pos:pos + n])
        pos += n
    recs = rx.drain()
sent = {k - FIRST_BIN: v[1] for (k, _), v in truth.items()}
print("%d bursts sent, %d decoded" % (len(sent), len(recs)))
for r in recs:
    ch, got = int(r["channel"]), r["min"].decode()
    print("  channel %3d  MIN %s  class %d  words valid %s  %s" % (ch, got, int(r["msg_class"]), "".join(str(int(v)) for v in r["valid"]),
                                                                 "ok" if sent.get(ch) == got else "MISMATCH"))
sys.exit(0 if len(recs) == len(sent) and all(sent.get(int(r["channel"])) == r["min"].decode() for r in recs) else 1)
