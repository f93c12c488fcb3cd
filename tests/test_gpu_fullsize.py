"""-m gpu, BASELINE size: the bench's own 2^27-sample wideband block (1 GiB, 416 planted seizure
bursts, built on the GPU exactly as bench.py builds it) through the headline path; every planted burst must come back with
the transmitted words, and a 16-channel slice of the filter bank's output at full size must give the SAME records through
the CPU model, bit for bit.  At both decimations of the filter bank (conftest.py: `decim`)."""
import os
import sys

import numpy as np
import pytest

import oracle
from gr_amps_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.mark.parametrize("spec,sid", [("sine", 2), ("atan", 0), ("exact", 3)])
def test_bench_block_full_size_words_and_cpu_model_slice(gpu, spec, sid, decim):
    import torch
    import bench
    NW, first, C, D = 1 << 27, 96, 832, decim
    sps = 1536 // D
    x, planted = bench.make_wideband_batch(torch, gpu, NW, first, C, 2, seed=1)
    torch.cuda.synchronize()
    wb = {"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}
    flush = torch.zeros(64 * D, dtype=torch.complex64, device=gpu)
    with capi.Recc(n_channels=C, sps=sps, max_samples=NW // D + 72, max_bursts=4096, wideband=wb, slicer=spec) as r:
        r.push_wideband(x)
        r.push_wideband(flush)
        recs = r.drain()
    by_ch = {}
    for g in recs:
        by_ch.setdefault(int(g["channel"]), []).append(g)
    # every planted burst: transmitted MIN and every transmitted word (raw repeat 0 and corrected bits), all words valid
    for c, (min10, words) in planted.items():
        assert c in by_ch, "planted burst in channel %d not found" % c
        g = by_ch[c][0]
        assert g["min"].decode() == min10 and g["valid"].all() and g["manch_bad"].sum() == 0
        for w, bits in enumerate(words):
            assert list(g["word_raw"][w][:36]) == list(bits) == list(g["word_dec"][w])
    assert len(recs) == len(planted) == 416                   # and nothing else was "found" in 832 channels x 262144 samples of noise
    # a 16-channel slice at full size: the filter bank's own output through the CPU model gives the same records
    lo = 400
    with capi.Recc(n_channels=16, sps=sps, max_samples=NW // D + 72, max_bursts=64,
                   wideband=dict(wb, first_channel=first + lo)) as r:
        chan = r.debug_channelize(x)
    assert chan.shape == (16, NW // D // 4 * 4)                  # the unfused form consumes whole groups of four frames at D = 768
    want = oracle.fused_push_all(chan, sps=sps, slicer=sid)
    got = np.array([g for g in recs if lo <= int(g["channel"]) < lo + 16], dtype=capi.BURST_DTYPE)
    got["channel"] -= lo
    assert len(want) == 8 and got.tobytes() == want.tobytes()
