"""-m gpu: config-0 plumbing through the host-side C++ blocks (gr::amps::recc -> gr::amps::recc_decode,
and gr::amps::recc_fused -> recc_decode) -- the reference's block API above the C ABI.  The program
gr_amps_amd/recctest prints every message recc_decode publishes; the expected lines are rebuilt from
the oracle (recc work() replica + burst decode + reply words)."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from gr_amps_amd import synth
from gr_amps_amd.host import build_host

pytestmark = pytest.mark.gpu


def _bits(a):
    return "".join(str(int(b)) for b in a)


def _expected_lines(records):
    lines = []
    for rec in records:
        r = oracle.reply_words(rec)
        if r.has_focc:
            lines.append(f"MSG focc_words stream={r.focc_stream} n={r.focc_nwords} w1={_bits(r.focc_word1)} w2={_bits(r.focc_word2)}")
        if r.has_fvc:
            lines.append(f"MSG fvc_words n={r.fvc_count} w1={_bits(r.fvc_word1)} repeat={r.fvc_repeat}")
        if r.has_mutes:
            lines.append(f"MSG fvc_mute {r.fvc_mute}")
            lines.append(f"MSG audio_mute {r.audio_mute}")
        if r.has_command:
            lines.append("MSG command_out " + r.command.decode())
    return lines


def _run(mode, path, chunk):
    _, exe = build_host()
    out = subprocess.run([exe, mode, path, str(chunk)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return [l for l in out.stdout.splitlines() if l.startswith("MSG ")]


@pytest.mark.parametrize("chunk", [4096, 1000, 8191])
def test_recctest_symbol_file_through_recc_and_recc_decode(gpu, tmp_path, chunk):
    rng = np.random.default_rng(31)
    bursts, off = [], 3000
    for _ in range(5):
        _, _, _, _, words = synth.random_message(rng)
        bursts.append((off, synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)))
        off += 3456 + 74 + 8191 + int(rng.integers(100, 900))
    s = synth.symbol_stream(off + 9000, bursts, rng)
    p = tmp_path / "recc.syms"
    s.tofile(p)
    pubs = oracle.Recc().run(s, chunk)
    assert len(pubs) == 5
    want = _expected_lines(oracle.decode_bursts(np.stack([b for _, b in pubs])))
    assert _run("syms", str(p), chunk) == want and len(want) >= 5


def test_recctest_iq_file_through_recc_fused(gpu, tmp_path):
    iq, truth = synth.make_channel_block(4 * 40000, 4, seed=32)
    p = tmp_path / "recc200k.fc32"
    iq.tofile(p)
    want = _expected_lines(oracle.fused_push_all(iq[None, :], block=10000))
    assert _run("iq", str(p), 10000) == want and len(want) >= 4


def test_recctest_raw_capture_through_channel_filter_and_recc_fused(gpu, tmp_path):
    """The flow graph's own capture format (grc/recctest.grc:591: fc32 at 400 ksps, channel at +160 kHz): the GPU
    channel filter + fused chain must publish what the restated reference chain (G1..G4 + R2..R8) decodes."""
    iq400, truth = synth.make_channel_block(2 * 400000, 6, seed=33, sps=20, spacing=(3456 + 74 + 4096 + 600) * 20)
    k = np.arange(iq400.size)
    iq400 = (iq400 * np.exp(2j * np.pi * 160e3 * k / 400e3)).astype(np.complex64)
    p = tmp_path / "recc.raw"
    iq400.tofile(p)
    ref = oracle.chain_iq400(iq400, 160e3, chunk=4096)
    assert len(ref) >= len(truth) // 2
    got = _run("raw", str(p), 50001)
    want = _expected_lines(ref)
    # the reference chain may miss bursts while its M&M loop is still pulling in; everything it does decode must be
    # published identically by the GPU chain, in order
    it = iter(got)
    assert all(line in it for line in want), (want, got)
    assert len(got) >= len(want)


def test_recc_fused_bursts_port_carries_what_amps_recc_publishes(gpu, tmp_path):
    """recc_fused's "bursts" port (the header's contract, matching lib/recc_impl.cc:126): the 3374-byte symbol blob, decoded
    by recc_decode, must produce exactly what the already decoded "records" port produces"""
    iq, truth = synth.make_channel_block(4 * 40000, 4, seed=34)
    p = tmp_path / "recc200k.fc32"
    iq.tofile(p)
    via_records = _run("iq", str(p), 10000)
    via_bursts = _run("iqb", str(p), 10000)
    assert via_bursts == via_records and len(via_records) >= 4


def test_drain_bursts_symbols_decode_back_to_the_records(gpu):
    from gr_amps_amd import capi
    iq, truth = synth.make_channel_block(3 * 40000, 3, seed=35)
    with capi.Recc(n_channels=1, sps=10, max_samples=iq.size, max_bursts=16, keep_bursts=True) as r:
        r.push_iq(iq[None, :])
        recs, syms = r.drain_bursts()
        again = r.decode_bursts(syms)
    assert len(recs) == len(truth) == 3 and syms.shape == (3, 3374) and set(np.unique(syms)) <= {0, 1}
    for f in ("dcc", "valid", "word_raw", "word_dec", "msg_class", "min", "esn", "dialed", "manch_bad", "first_valid_rep"):
        assert np.array_equal(recs[f], again[f]), f
    want = oracle.fused_push_all(iq[None, :])
    assert recs.tobytes() == want.tobytes()


@pytest.mark.parametrize("chunk", [4096, 1000])
def test_recc_bank_block_equals_one_recc_block_per_channel(gpu, tmp_path, chunk):
    """gr::amps::recc_bank: C channels in one block, one launch per work(); every channel must publish what a lone
    gr::amps::recc (= the reference's work() replica) publishes on that channel's stream with the same chunk schedule"""
    rng = np.random.default_rng(36)
    bursts, off = [], 3000
    for _ in range(3):
        _, _, _, _, words = synth.random_message(rng)
        bursts.append((off, synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)))
        off += 3456 + 74 + 8191 + int(rng.integers(100, 900))
    s = synth.symbol_stream(off + 9000, bursts, rng)
    p = tmp_path / "recc.syms"
    s.tofile(p)
    C = 6
    got = _run("bank", str(p), chunk) if False else None
    _, exe = build_host()
    out = subprocess.run([exe, "bank", str(p), str(chunk), str(C)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [l for l in out.stdout.splitlines() if l.startswith("MSG ")]
    # expected: per channel c the stream is the file delayed by 37 c symbols; calls interleave channel-ordered per work() call
    total = s.size + 37 * C
    chans = [np.concatenate([np.zeros(37 * c, np.uint8), s, np.zeros(37 * (C - c), np.uint8)]) for c in range(C)]
    refs = [oracle.Recc() for _ in range(C)]
    want = []
    for o in range(0, total, chunk):
        for c in range(C):
            b = refs[c].work(chans[c][o:o + chunk])
            if b is not None:
                want.append("MSG channel %d" % c)
                want += _expected_lines(oracle.decode_bursts(b[None, :]))
    assert got == want and sum(1 for l in got if l.startswith("MSG channel")) == 3 * C


def test_recc_wideband_block_decodes_the_band(gpu, tmp_path):
    """gr::amps::recc_wideband: one 30.72 Msps capture file in, (channel, burst) pairs out, through recc_decode: the lines of
    every planted burst, under its channel, as the C ABI's own records give them (ragged work() sizes)"""
    from gr_amps_amd import capi, synth_wideband as sw
    n = int(0.26 * sw.FS_WIDE) // 512 * 512
    planted = [(96 + 7, 150000), (96 + 500, 90000), (96 + 831, 230000)]
    x, truth = sw.make_wideband(n, planted, seed=41)
    p = tmp_path / "band.fc32"
    x.tofile(p)
    with capi.Recc(n_channels=832, sps=3, max_samples=n // 512 + 72, max_bursts=64,
                   wideband={"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": 96}) as r:
        r.push_wideband(x)
        r.push_wideband(np.zeros(64 * 512, np.complex64))
        recs = r.drain()
    assert sorted(int(g["channel"]) for g in recs) == [7, 500, 831]
    want = []
    for g in recs:                                   # the block publishes in (channel, position) order per work() call
        want.append("MSG channel %d" % int(g["channel"]))
        want += _expected_lines(recs[recs["channel"] == g["channel"]])
    _, exe = build_host()
    out = subprocess.run([exe, "wide", str(p), "777777"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [l for l in out.stdout.splitlines() if l.startswith("MSG ")]
    assert sorted(got) == sorted(want) and [l for l in got if l.startswith("MSG channel")] != []


@pytest.mark.parametrize("mode", [0, 1])
def test_recc_wideband_blocks_share_one_band_over_four_processes(gpu, tmp_path, mode):
    """gr::amps::recc_wideband::set_rccl (BASELINE configs[4] on the flow-graph side): four processes, one block each, channel groups 0..3;
    rank 0 owns the capture and the library distributes it (mode 0: broadcast, 1: scatter + all-gather).  The schedulers of separate flow
    graphs do not agree on item counts: the ranks here are fed in DIFFERENT chunk sizes, the non-root ranks' items are
    ignored, and the ranks stay in step by stream position.  The union of what the four processes print is what ONE whole-band block
    prints.  (Transport between the processes: the loop-back stand-in of tests/loopccl -- RCCL refuses four ranks on one GPU.)"""
    import loopccl
    from gr_amps_amd import synth_wideband as sw
    n = int(0.26 * sw.FS_WIDE) // 512 * 512
    planted = [(96 + c, off) for c, off in ((7, 150000), (16, 400000), (40, 90000), (63, 230000), (500, 60000), (831, 300000))]   # all four groups
    x, truth = sw.make_wideband(n, planted, seed=43)
    p = tmp_path / "band.fc32"
    x.tofile(p)
    _, exe = build_host()
    one = subprocess.run([exe, "wide", str(p), "777777"], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr
    want = [l for l in one.stdout.splitlines() if l.startswith("MSG ")]
    assert sum(1 for l in want if l.startswith("MSG channel")) == len(planted)
    env = dict(os.environ, AMPS_RECC_RCCL_LIB=loopccl.build(), LOOPCCL_DIR=str(tmp_path))
    chunks = ["600000", "77777", "2000001", "333333"]            # rank 0's cut of the stream, and three others that share nothing with it
    procs = [subprocess.Popen([exe, "widerank", str(p), chunks[r], str(tmp_path / "id.bin"), "4", str(r), str(mode)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(4)]
    got = []
    for r, q in enumerate(procs):
        try:
            o, e = q.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for k in procs:
                k.kill()
            raise AssertionError("rank %d hung" % r)
        assert q.returncode == 0, (r, e[-2000:])
        lines = [l for l in o.splitlines() if l.startswith("MSG ")]
        for l in lines:
            if l.startswith("MSG channel"):                       # a rank publishes its own group only: (bin mod 64) in its window of 16
                assert ((96 + int(l.split()[2])) % 64) // 16 == r
        got += lines
    assert sorted(got) == sorted(want)
