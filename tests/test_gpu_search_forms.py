"""-m gpu: the three launch structures of the wideband seam's trigger search give the same records, byte for byte.

  default                          the search stage INSIDE the resolve kernel (round 6: recc_resolve_kernel<..., SEARCH>, one launch per push
                                   behind the filter bank; a quarter of the channel's push per wave, hits in LDS)
  AMPS_RECC_BITS_KERNEL=separate   recc_bits_kernel as its own launch in front of the resolve kernel (rounds 2-5), hits in HBM lists
  AMPS_RECC_BITS_KERNEL=front      the streaming kernel's bit-domain mode (D = 512 only): an independent implementation of the search

The knob is read once per process, so every form runs in its own interpreter on the same stream: ragged pushes, exact and tolerant sync,
both decimations, bursts at the edges of the quarters included."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, hashlib
import numpy as np
sys.path.insert(0, %(root)r)
from gr_amps_amd import capi, synth_wideband as sw
D, tol = %(decim)d, %(tol)d
first, C = 96, 832
n = int(0.42 * sw.FS_WIDE) // 1536 * 1536
rng = np.random.default_rng(17)
# bursts spread over the push so that triggers fall into every quarter of it, a few right behind one another in time
offs = [30000 + int(rng.integers(0, n - 3456 * 1536 - 60000)) for _ in range(14)]
planted = [((first + 59 * i + (i %% 7)) %% 1024, offs[i]) for i in range(14)]
x, truth = sw.make_wideband(n, planted, seed=23, snr_db=22.0)
with capi.Recc(n_channels=C, sps=1536 // D, max_samples=n // D + 72, max_bursts=256, sync_tolerance=tol,
               wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}) as r:
    recs = []
    for lo, hi in ((0, 3000000), (3000000, 3000001), (3000001, 9000000), (9000000, n)):
        r.push_wideband(x[lo:hi])
        recs.append(r.drain())
    r.push_wideband(np.zeros(64 * D, np.complex64))
    recs.append(r.drain())
got = np.concatenate(recs)
got = got[np.lexsort((got["position"], got["channel"]))]
print("RESULT", len(got), hashlib.sha256(got.tobytes()).hexdigest())
'''


def _run(decim, tol, knob):
    env = dict(os.environ)
    env.pop("AMPS_RECC_BITS_KERNEL", None)
    if knob:
        env["AMPS_RECC_BITS_KERNEL"] = knob
    p = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT, "decim": decim, "tol": tol}], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    return int(line[1]), line[2]


@pytest.mark.parametrize("decim,tol", [(768, 0), (768, 3), (512, 0), (512, 3)])
def test_search_inside_the_resolve_kernel_equals_the_separate_launch(gpu, decim, tol):
    n0, h0 = _run(decim, tol, None)
    n1, h1 = _run(decim, tol, "separate")
    assert n0 == n1 == 14 and h0 == h1
    if decim == 512:                                         # the streaming kernel's bit-domain mode exists at three samples per symbol
        n2, h2 = _run(decim, tol, "front")
        assert (n2, h2) == (n0, h0)
