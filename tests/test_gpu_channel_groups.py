"""-m gpu: channel groups (cfg.wideband_groups): G handles fed the same wideband stream, each decoding one interleaved group of the
band's channels -- the one-band multi-GPU split of BASELINE configs[4], where a rank skips the last FFT pass and the slicer for
everybody else's bins.  The union of the groups' records must be the whole-band handle's records, byte for byte."""
import numpy as np
import pytest

from gr_amps_amd import capi, synth_wideband as sw

pytestmark = pytest.mark.gpu
FIRST, CW = 96, 832
D = 512


def _records(x, n, D=512, **wb):
    with capi.Recc(n_channels=CW, sps=1536 // D, max_samples=n // D + 72, max_bursts=1024,
                   wideband=dict({"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": FIRST}, **wb)) as r:
        for part in np.array_split(x, 3):                         # streaming pushes: the carry and the pre-roll are exercised too
            r.push_wideband(part)
        r.push_wideband(np.zeros(64 * D, np.complex64))
        return r.drain()


@pytest.mark.parametrize("G", [2, 4, 8])
def test_groups_partition_the_band_and_reproduce_its_records(gpu, G, decim):
    D = decim
    rng = np.random.default_rng(60 + G)
    n = int(0.3 * sw.FS_WIDE) // 1536 * 1536
    chans = sorted(set(int(c) for c in rng.integers(0, CW, 40)) | {0, 1, 7, 8, 63, 64, 415, 416, 831})
    planted = [((FIRST + c) % 1024, int(rng.integers(20000, n - 3456 * 1536 - 20000))) for c in chans]
    x, truth = sw.make_wideband(n, planted, seed=900 + G, snr_db=24.0)
    whole = _records(x, n, D)
    assert len(whole) == len(chans)
    parts, seen = [], set()
    for r in range(G):
        got = _records(x, n, D, groups=G, group=r)
        for g in got:                                             # every record belongs to this group: (bin mod 64) in the group's window
            k = (FIRST + int(g["channel"])) % 1024
            assert (k % 64) // (64 // G) == r
        seen |= {int(g["channel"]) for g in got}
        parts.append(got)
    union = np.concatenate(parts)
    union = union[np.lexsort((union["position"], union["channel"]))]
    assert seen == set(chans)
    assert union.tobytes() == whole.tobytes()


def test_group_argument_errors(gpu):
    wb = {"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": FIRST}
    for bad in ({"groups": 3, "group": 0}, {"groups": 8, "group": 8}, {"groups": 16, "group": 0}):
        with pytest.raises(capi.AmpsError):
            capi.Recc(n_channels=CW, sps=3, max_samples=4096, max_bursts=16, wideband=dict(wb, **bad))
    with pytest.raises(capi.AmpsError):                            # channel groups exist in the fused form only
        capi.Recc(n_channels=CW, sps=3, max_samples=4096, max_bursts=16, unfused_wideband=True, wideband=dict(wb, groups=2, group=1))


def test_a_group_handle_serves_the_wideband_seam_only(gpu):
    """a channel-group handle owns its group's rows; the channel-major seams would take group rows for band channels and number their
    records through the group's row map, so they refuse: -ENOSYS, and the handle keeps working"""
    import errno
    n = int(0.1 * sw.FS_WIDE) // D * D
    with capi.Recc(n_channels=CW, sps=3, max_samples=n // D + 72, max_bursts=64,
                   wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": FIRST, "groups": 2, "group": 1}) as r:
        with pytest.raises(capi.AmpsError) as e:
            r.push_iq(np.zeros((CW, 640), np.complex64))
        assert e.value.code == -errno.ENOSYS
        with pytest.raises(capi.AmpsError) as e:
            r.push_symbols(np.zeros((CW, 100), np.uint8))
        assert e.value.code == -errno.ENOSYS
        r.push_wideband(np.zeros(n, np.complex64))
        assert len(r.drain()) == 0
