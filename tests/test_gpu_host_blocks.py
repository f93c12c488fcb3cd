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
