"""-m gpu: the two forms of the capture stage give the same records.  A handle with fewer than 64 channels queues accepted captures
and decodes them in a second kernel (one workgroup per burst: one channel x 2^26 samples holds 745 bursts); with 64 or more the
resolve kernel's own workgroup decodes its channel's bursts (recc_resolve.hip.h).  The same eight channels are pushed through a
handle of each kind -- the second one padded with idle channels -- in ragged pieces, so captures also wait for their tails
across pushes."""
import numpy as np
import pytest

from gr_amps_amd import capi, synth

pytestmark = pytest.mark.gpu
FIELDS = ("channel", "position", "dcc", "valid", "first_valid_rep", "manch_bad", "word_raw", "word_dec", "a_MIN1", "b_MIN2", "msg_class",
          "min", "dialed", "esn", "n_called_words", "flags")


def _records(iq, n_channels, pieces, **kw):
    c, n = iq.shape
    x = np.zeros((n_channels, n), np.complex64)
    x[:c] = iq
    out, syms = [], []
    with capi.Recc(n_channels=n_channels, sps=10, max_samples=n, max_bursts=1024, **kw) as r:
        o = 0
        for p in pieces:
            r.push_iq(np.ascontiguousarray(x[:, o:o + p]))
            o += p
            if kw.get("keep_bursts"):
                g, s = r.drain_bursts()
                out.append(g); syms.append(s)
            else:
                out.append(r.drain())
        assert o == n
    recs = np.concatenate(out)
    order = np.lexsort((recs["position"], recs["channel"]))
    order = order[recs["channel"][order] < c]
    return recs[order], (np.concatenate(syms)[order] if syms else None)


@pytest.mark.parametrize("kw", [{}, {"majority": True}, {"keep_bursts": True}])
def test_queue_form_and_fused_form_agree(gpu, kw):
    n = 200000
    iq = np.stack([synth.make_channel_block(n, 4, seed=600 + c, sps=10, snr_db=14.0)[0] for c in range(8)])
    pieces = [70016, 3, 40000, 64, 89917]
    assert sum(pieces) == n
    a, sa = _records(iq, 8, pieces, **kw)        # queue form
    b, sb = _records(iq, 72, pieces, **kw)       # fused form
    assert len(a) == len(b) and len(a) >= 24, (len(a), len(b))
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), f
    if sa is not None:
        assert sa.shape == sb.shape and np.array_equal(sa, sb)
        assert set(np.unique(sa).tolist()) <= {0, 1} and sa.any()
