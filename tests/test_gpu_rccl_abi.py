"""-m gpu: RCCL inside the C ABI (amps_recc_rccl_unique_id / _rccl_init / _push_wideband_bcast / _drain_gather, include/amps_recc.h) -- the entry a
flow graph uses to run one band over the GPUs of a node.  A box has one GPU, so the communicator has ONE rank here (RCCL refuses two
ranks on a device): ncclCommInitRank, ncclBroadcast into the two receive buffers, the event ordering against the handle's
stream and the push behind it all run; the records must be those of a plain amps_recc_push_wideband of the same stream."""
import os

import numpy as np
import pytest

# a one-rank communicator uses a device-resident block in place since round 6 (no 1 GiB copy per push); this file is about the real
# librccl's data collectives, so they are forced on (read once by the library, at its first distributed push)
os.environ["AMPS_RECC_RCCL_FORCE_COLLECTIVE"] = "1"

from gr_amps_amd import capi, synth_wideband as sw

pytestmark = pytest.mark.gpu
D, FIRST, CW = 512, 96, 832
WB = {"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": FIRST}


def test_broadcast_push_equals_plain_push(gpu):
    import torch
    n = int(0.25 * sw.FS_WIDE) // D * D
    planted = [(FIRST + 7, 90000), (FIRST + 300, 150000), (FIRST + 831, 120000)]
    x, truth = sw.make_wideband(n, planted, seed=12)
    parts = [x[:2000000], x[2000000:2000000 + 3000064], x[5000064:]]
    with capi.Recc(n_channels=CW, sps=3, max_samples=n // D + 72, max_bursts=64, wideband=WB) as r:
        for p in parts:
            r.push_wideband(p)
        r.push_wideband(np.zeros(64 * D, np.complex64))
        want = r.drain()
    assert len(want) == 3
    with capi.Recc(n_channels=CW, sps=3, max_samples=n // D + 72, max_bursts=64, wideband=WB) as r:
        with pytest.raises(capi.AmpsError):                      # no communicator yet
            r.push_wideband_bcast(torch.from_numpy(parts[0]).to(gpu), len(parts[0]))
        uid = capi.Recc.rccl_unique_id()
        assert len(uid) == 128
        r.rccl_init(uid, 1, 0)
        with pytest.raises(capi.AmpsError):                      # one communicator per handle
            r.rccl_init(uid, 1, 0)
        dev = [torch.from_numpy(p).to(gpu) for p in parts] + [torch.zeros(64 * D, dtype=torch.complex64, device=gpu)]
        for t in dev:                                            # four pushes: both receive buffers are reused (grown once)
            r.push_wideband_bcast(t, t.shape[0], root=0)
        got = r.drain()
    assert got.tobytes() == want.tobytes()
    # the same with HOST blocks at the root (what gr::amps::recc_wideband::set_rccl feeds it): staged by the library
    with capi.Recc(n_channels=CW, sps=3, max_samples=n // D + 72, max_bursts=64, wideband=WB) as r:
        with pytest.raises(capi.AmpsError):                      # the collective drain needs the communicator too
            r.drain_gather()
        r.rccl_init(capi.Recc.rccl_unique_id(), 1, 0)
        assert len(r.drain_gather(root=0)) == 0                  # nothing yet: the count exchange alone
        for p in parts + [np.zeros(64 * D, np.complex64)]:
            r.push_wideband_bcast(p, len(p), root=0)
        # amps_recc_drain_gather: the ranks' records merged at the root (here: the one rank's own list through both all-gathers)
        assert r.drain_gather(root=0).tobytes() == want.tobytes()
        assert len(r.drain_gather(root=0)) == 0
        with pytest.raises(capi.AmpsError):
            r.drain_gather(root=1)                               # not a rank of this communicator
        for p in parts + [np.zeros(64 * D, np.complex64)]:       # a list longer than the caller's buffer: what fits, and -ENOSPC
            r.push_wideband_bcast(p, len(p), root=0)
        with pytest.raises(capi.AmpsError) as ei:
            r.drain_gather(root=0, cap=2)
        assert ei.value.code == -28
    for (k, off), (kind, min10, esn, dialed, words) in truth.items():
        assert any(g["min"].decode() == min10 and g["valid"].all() for g in got)
