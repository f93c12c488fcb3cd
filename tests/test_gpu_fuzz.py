"""-m gpu: a slice of the randomised differential campaign (tests/fuzzlib.py; scripts/fuzz_parity.py runs it at length)."""
import pytest

import fuzzlib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,resident", [(11, False), (12, False), (13, True)])
def test_random_streams_match_the_cpu_model(gpu, seed, resident):
    bad = []
    for case in range(40):
        ok, info = fuzzlib.run_case(case, seed, resident)
        if not ok:
            bad.append((case, info))
    assert not bad, bad


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wideband_random_schedules_give_the_one_shot_records(gpu, seed, decim):
    """Wideband seam: whatever the push schedule (ragged sizes, host or device blocks, sync / split / no drains in between,
    fused or two-kernel form, exact or tolerant sync) the records equal those of one push with the same tolerance."""
    import numpy as np
    import torch
    from gr_amps_amd import capi, synth_wideband as sw
    D = decim
    rng = np.random.default_rng(seed)
    first, C = int(rng.integers(0, 1024)), 832
    n = int(0.26 * sw.FS_WIDE) // 1536 * 1536
    chans = rng.choice(C, size=6, replace=False)
    bursts = [((first + int(c)) % 1024, int(rng.integers(20000, 2200000))) for c in chans]
    x, truth = sw.make_wideband(n, bursts, seed=100 + seed)
    wb = {"channels": 1024, "decim": D, "taps_per_branch": 8, "first_channel": first}

    spec = ("exact", "atan", "sine", "product")[seed % 4]      # the slicer spec under test rotates with the seed

    def run(schedule, unfused, tol, resident, mode):
        with capi.Recc(n_channels=C, sps=1536 // D, max_samples=n // D + 72, max_bursts=64, unfused_wideband=unfused,
                       sync_tolerance=tol, wideband=wb, slicer=spec) as r:
            off, recs, open_, keep = 0, [], False, []
            for m in schedule + [64 * D]:                      # the last block is silence: flushes the held-back frames
                blk = x[off:off + m] if off < n else np.zeros(m, np.complex64)
                off += m
                if resident:
                    blk = torch.from_numpy(np.ascontiguousarray(blk)).to("cuda:0")
                    torch.cuda.synchronize()
                    keep.append(blk)
                r.push_wideband(blk)
                if mode == "sync":
                    recs.append(r.drain())
                elif mode == "split":
                    if open_:
                        recs.append(r.drain_end())
                    r.drain_begin()
                    open_ = True
            recs.append(r.drain_end() if open_ else r.drain())
            got = np.concatenate(recs)
        return got[np.lexsort((got["position"], got["channel"]))]

    # one reference per tolerance: a tolerant trigger can match one sample phase more on one side, which moves the run
    # centre (the record's position) by a sample -- exact and tolerant runs agree on the words, not on every position
    ref = {0: run([n], False, 0, False, "sync"), 3: run([n], False, 3, False, "sync")}
    assert len(ref[0]) == len(ref[3]) == len(bursts)
    assert sorted(g["min"].decode() for g in ref[0]) == sorted(v[1] for v in truth.values())
    for a, b in zip(ref[0], ref[3]):
        assert a["channel"] == b["channel"] and abs(int(a["position"]) - int(b["position"])) <= 1
        assert np.array_equal(a["word_raw"], b["word_raw"]) and np.array_equal(a["word_dec"], b["word_dec"])
    for _ in range(5):
        cuts = np.sort(rng.integers(1, n, size=int(rng.integers(1, 6))))
        schedule = [int(b - a) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n]) if b > a]
        tol = int(rng.choice([0, 3]))
        got = run(schedule, bool(rng.integers(0, 2)), tol, bool(rng.integers(0, 2)), str(rng.choice(["sync", "split", "none"])))
        assert got.tobytes() == ref[tol].tobytes(), schedule


@pytest.mark.parametrize("seed", [1, 2])
def test_translate_seam_random_schedules(gpu, seed):
    """push_raw with ragged host/device blocks and no drains in between == one push == the restated reference chain's words."""
    import numpy as np
    import torch
    import oracle
    from gr_amps_amd import capi, synth
    rng = np.random.default_rng(seed)
    fc = float(rng.choice([160e3, -160e3, 120e3]))
    iq400, truth = synth.make_channel_block(2 * 400000, 8, seed=900 + seed, sps=20, spacing=(3456 + 74 + 4096 + 600) * 20)
    k = np.arange(iq400.size)
    iq400 = (iq400 * np.exp(2j * np.pi * fc * k / 400e3)).astype(np.complex64)
    n = iq400.size

    def run(schedule, resident):
        with capi.Recc(n_channels=1, sps=10, max_samples=n // 2 + 8, max_bursts=64) as r:
            r.set_xlate(rate_hz=400e3, center_hz=fc, decim=2)
            off, keep = 0, []
            for m in schedule:
                blk = np.ascontiguousarray(iq400[None, off:off + m])
                off += m
                if resident:
                    blk = torch.from_numpy(blk).to("cuda:0")
                    torch.cuda.synchronize()
                    keep.append(blk)
                r.push_raw(blk)
            return r.drain()

    ref = run([n], False)
    assert len(ref) == len(truth)
    for _ in range(4):
        cuts = np.sort(rng.integers(1, n, size=int(rng.integers(1, 7))))
        schedule = [int(b - a) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n]) if b > a]
        assert run(schedule, bool(rng.integers(0, 2))).tobytes() == ref.tobytes(), schedule
    chain = oracle.chain_iq400(iq400, fc, chunk=4096)
    by_min = {g["min"]: g for g in ref}
    assert len(chain) >= len(ref) // 2
    for rr in chain:
        g = by_min[rr["min"]]
        assert np.array_equal(rr["word_raw"], g["word_raw"]) and np.array_equal(rr["word_dec"], g["word_dec"])
