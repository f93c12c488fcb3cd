"""-m gpu: the HIP-event timing modes of the C ABI (include/amps_recc.h, amps_recc_set_timing) count what they say they bracket, and
the records do not depend on the mode.  ALL: every launch; DOMINANT: the filter bank (wideband seam) / the streaming kernel (IQ seam)
of every push; DOMINANT_SAMPLED: of every AMPS_RECC_TIMING_SAMPLE_PERIOD-th push (bench.py's long timed regions: the two event
records per push cost a wideband step 1.9 %, profiles/r06/event_cost.txt); OFF: none."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PERIOD = 8


def test_wideband_modes_count_the_filter_bank_and_leave_the_records_alone(gpu, decim):
    from gr_amps_amd import capi, synth_wideband as sw
    D, first, C = decim, 96, 832
    n = 64 * 1536 * 90                                          # whole 64-frame groups at either decimation; one burst per push
    x, _ = sw.make_wideband(n, [((first + 100) % 1024, 200000)], seed=5, snr_db=25.0)
    got = {}
    for mode in ("off", "all", "dominant", "sampled"):
        with capi.Recc(n_channels=C, sps=1536 // D, max_samples=n // D + 72, max_bursts=64, time_kernels=True,
                       wideband={"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}) as r:
            r.set_timing(mode)
            r.timing(reset=True)
            recs = []
            for _ in range(2 * PERIOD + 3):
                r.push_wideband(x)
                recs.append(r.drain())
            t = r.timing()
            got[mode] = (int(t["launches_channelizer"]), float(t["ms_channelizer"]), float(t["ms_resolve"]), np.concatenate(recs).tobytes())
    pushes = 2 * PERIOD + 3
    assert got["off"][0] == 0 and got["off"][1] == 0.0
    assert got["all"][0] == pushes and got["all"][2] > 0.0                  # every launch, the kernels behind the filter bank too
    assert got["dominant"][0] == pushes and got["dominant"][2] == 0.0       # the filter bank of every push, nothing else
    assert got["sampled"][0] == 3 and got["sampled"][2] == 0.0              # pushes 0, 8, 16
    per = {m: got[m][1] / got[m][0] for m in ("all", "dominant", "sampled")}
    assert max(per.values()) < 3.0 * min(per.values())                       # the same kernel, whoever brackets it
    assert got["off"][3] == got["all"][3] == got["dominant"][3] == got["sampled"][3] and len(got["off"][3]) > 0


def test_iq_seam_sampled_mode_counts_the_streaming_kernel(gpu):
    from gr_amps_amd import capi
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((4, 8192)) + 1j * rng.standard_normal((4, 8192))).astype(np.complex64)
    with capi.Recc(n_channels=4, sps=10, max_samples=8192, max_bursts=16, time_kernels=True) as r:
        r.set_timing("sampled")
        r.timing(reset=True)
        for _ in range(PERIOD + 1):
            r.push_iq(x)
            r.drain()
        t = r.timing()
        assert int(t["launches_front"]) == 2 and t["ms_front"] > 0.0 and t["ms_resolve"] == 0.0
        assert capi.load().amps_recc_set_timing(r._h, 4) == -22 and capi.load().amps_recc_set_timing(r._h, -1) == -22
