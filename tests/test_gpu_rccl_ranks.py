"""-m gpu: the library's OWN collective path with 2, 4 and 8 ranks -- amps_recc_rccl_init (group-split check, common capacities),
amps_recc_push_wideband_dist in both modes (flat broadcast; scatter + all-gather), amps_recc_drain_gather, the status word that
travels through every collective and the bounded waits -- on the one GPU a test box has.  RCCL refuses two ranks on one device, so the
ranks meet through tests/loopccl (a loop-back stand-in for the dozen librccl entry points the library binds, selected with
AMPS_RECC_RCCL_LIB); everything above that transport is the code an 8-GPU node runs.  The real librccl is exercised with a one-rank
communicator in tests/test_gpu_rccl_abi.py.

What must hold: the records gathered at the root are, byte for byte, the records ONE whole-band handle drains from the same stream
(channel independence: the reference keeps per-instance state only, lib/recc_impl.h:31-43)."""
import errno
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gr_amps_amd import capi, synth_wideband as sw

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, FIRST, CW = 512, 96, 832
WB = {"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": FIRST}


@pytest.fixture(scope="module")
def stream(gpu, tmp_path_factory):
    import loopccl
    lib = loopccl.build()
    d = tmp_path_factory.mktemp("ranks")
    n = int(0.25 * sw.FS_WIDE) // D * D
    # bursts in channels of every group of every split (bin mod 64 windows of 32 / 16 / 8), two of them in adjacent channels
    chans = [0, 7, 8, 31, 32, 63, 64, 300, 415, 416, 500, 831]
    rng = np.random.default_rng(5)
    planted = [((FIRST + c) % 1024, int(rng.integers(20000, n - 3456 * 1536 - 20000))) for c in chans]
    x, truth = sw.make_wideband(n, planted, seed=21, snr_db=24.0)
    np.save(d / "x.npy", x)
    np.save(d / "n.npy", np.int64(n))
    with capi.Recc(n_channels=CW, sps=3, max_samples=n // D + 72, max_bursts=256, wideband=WB) as r:
        for lo, hi in ((0, 2000000), (2000000, 5000064), (5000064, n)):
            r.push_wideband(x[lo:hi])
        r.push_wideband(np.zeros(64 * D, np.complex64))
        whole = r.drain()
    assert len(whole) == len(chans)
    return {"dir": d, "lib": lib, "n": n, "whole": whole}


def _run(stream, tmp_path, nranks, scenario, mode="broadcast", root=0, env_extra=None, expect_fail=()):
    for f in ("x.npy", "n.npy"):
        os.symlink(stream["dir"] / f, tmp_path / f)
    env = dict(os.environ, AMPS_RECC_RCCL_LIB=stream["lib"], LOOPCCL_DIR=str(tmp_path), LOOPCCL_TIMEOUT_MS="60000")
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), "--rank", str(r), "--nranks", str(nranks), "--dir", str(tmp_path),
                               "--scenario", scenario, "--mode", mode, "--root", str(root)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(nranks)]
    outs = []
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("rank %d of %s hung (the very thing this file is about)" % (r, scenario))
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0 or r in expect_fail, (r, outs[r][-3000:])
    return [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(nranks)]


@pytest.mark.parametrize("nranks,mode,root,transport", [(2, "broadcast", 0, "sync"), (2, "scatter_allgather", 1, "sync"), (4, "scatter_allgather", 0, "sync"),
                                                        (8, "broadcast", 0, "sync"), (8, "scatter_allgather", 3, "sync"),
                                                        (2, "broadcast", 1, "async"), (4, "scatter_allgather", 0, "async"), (8, "scatter_allgather", 3, "async")])
def test_ranks_reproduce_the_whole_band_records(stream, tmp_path, nranks, mode, root, transport):
    """transport "async" (LOOPCCL_ASYNC=1, round 6 -- ADVICE r05): the stand-in returns from a collective at once and the data lands
    later, in stream order, as with the real library; a consumer that did not wait for the library's `filled` event, or a collective
    that overwrote a receive buffer without waiting for `freed` / `root_ready`, would decode the wrong block here"""
    res = _run(stream, tmp_path, nranks, "dist", mode, root, env_extra={"LOOPCCL_ASYNC": "1"} if transport == "async" else None)
    n = stream["n"]
    want_pushed = [2000000, 3000064, n - 5000064, 64 * D, 100]
    for r, o in enumerate(res):
        assert o["init"] == 0
        assert o["pushed"] == want_pushed                      # the ROOT's block sizes arrived on every rank (the others passed none)
        i = o["info"]
        assert i["alive"] == 1 and i["nranks"] == i["comm_nranks"] == nranks and i["rank"] == i["comm_rank"] == r
        assert i["max_samples_per_push"] == (n // D + 72) * D and i["max_bursts_per_gather"] == 96      # min / max over the ranks
        assert "loopccl" in i["library"] and len(i["device_uuid"]) == 32
        ia = o["info_after"]
        assert ia["last_mode"] == mode and ia["collectives_timed"] == 5 and ia["collective_bytes"] == 8 * sum(want_pushed) and ia["collective_ms"] > 0
        assert o["gathered"] == (len(stream["whole"]) if r == root else 0) and o["second_gather"] == 0
    got = np.load(tmp_path / "gathered.npy")
    assert got.tobytes() == stream["whole"].tobytes()


def test_the_smallest_rank_bounds_the_push_and_an_oversize_block_is_everybodys_error(stream, tmp_path):
    """one rank's handle takes a third of the others' block: rccl_init agrees on the minimum; a block beyond it is refused at the root
    (-E2BIG) and on every other rank (-EREMOTEIO) BEFORE any data moves; the thirds then go through and the records are the band's"""
    res = _run(stream, tmp_path, 4, "small_rank", "scatter_allgather", 0)
    n = stream["n"]
    for r, o in enumerate(res):
        assert o["info"]["max_samples_per_push"] == ((n // 3) // D + 72) * D
        assert o["events"][0] == ["oversize", -errno.E2BIG if r == 0 else -errno.EREMOTEIO]
        assert o["pushed"] == [n // 3, n // 3, n - 2 * (n // 3), 64 * D, 100]
    got = np.load(tmp_path / "gathered.npy")
    assert got.tobytes() == stream["whole"].tobytes()           # other cuts of the same stream: the same records


@pytest.mark.parametrize("mode", ["broadcast", "scatter_allgather"])
def test_a_local_error_is_everybodys_verdict_and_the_communicator_survives(stream, tmp_path, mode):
    res = _run(stream, tmp_path, 4, "root_error", mode, 0)
    for r, o in enumerate(res):
        ev = dict((e[0], e[1]) for e in o["events"])
        assert ev["good"] == ev["good_again"] == ev["good_after_mismatch"] == 1000000
        assert ev["root_without_samples"] == (-errno.EINVAL if r == 0 else -errno.EREMOTEIO)
        assert ev["end_of_stream"] == -errno.ENODATA            # the root's (NULL, 0): every rank learns the stream has ended, nobody times out
        assert ev["mode_mismatch"] == -errno.EINVAL             # every rank sees all the headers: the same verdict everywhere


def test_a_handle_built_for_another_group_fails_the_init_on_every_rank(stream, tmp_path):
    """a 4-rank communicator on handles built for another split silently decoded part of the band"""
    res = _run(stream, tmp_path, 4, "bad_groups")
    assert [o["init"] for o in res] == [-errno.EREMOTEIO] * 3 + [-errno.EINVAL]


def test_a_rank_that_leaves_does_not_hang_its_peers(stream, tmp_path):
    """the last rank aborts its communicator after the first push; the others' next collective never completes (the stand-in leaves
    a blocked operation on the stream, as a collective kernel without its peer would be): the bounded wait ends it with -ETIMEDOUT
    after ~1.5 s, later calls answer -ENOTCONN at once; what the handle found behind the aborted collective is void (-ESTALE from the
    drains and the data seams: ADVICE r05) until amps_recc_reset, after which it decodes on its own"""
    res = _run(stream, tmp_path, 4, "peer_leaves", "broadcast", 0, env_extra={"LOOPCCL_ASYNC_HANG_MS": "20000", "LOOPCCL_TIMEOUT_MS": "300"})
    for r, o in enumerate(res):
        ev = {e[0]: e[1:] for e in o["events"]}
        assert ev["good"] == [1000000]
        if r == 3:
            assert ev["after_own_abort"] == [-errno.ENOTCONN]
        else:
            rc, secs = ev["peer_gone"]
            assert rc == -errno.ETIMEDOUT and 1.0 < secs < 10.0, ev
            assert ev["after_timeout"] == [-errno.ENOTCONN] and ev["gather_after_timeout"] == [-errno.ENOTCONN]
        assert ev["drain_before_reset"] == [-errno.ESTALE] and ev["push_before_reset"] == [-errno.ESTALE]
        assert o["info_after"]["alive"] == 0
        assert o["plain_drain_after"] == 0
