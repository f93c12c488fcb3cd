"""Regenerates tests/golden/recc_golden_r02.npz -- round-2 regression fixtures produced by the CPU oracle (same status as
make_golden.py: they pin the ORACLE over time and travel to the GPU box; they are not reference-derived).

Contents: the records of recc_golden.npz's IQ block under slicer specs B and C; a noisy two-channel block (10 dB, int8 samples
x/48) with the slicer bit streams (sha256) and the records of all three specs, at an SNR where the specs differ.   Run from the repo root:  python tests/golden/make_golden_r02.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from gr_amps_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def q8(x, scale):
    return np.clip(np.round(np.ascontiguousarray(x).view(np.float32) * scale), -127, 127).astype(np.int8)


def main():
    out = {}
    base = np.load(os.path.join(HERE, "recc_golden.npz"))
    x = (base["iq_i16"].astype(np.float32) / 8192.0).view(np.complex64)
    for code, name in ((1, "product"), (2, "sine")):
        out[f"iq_records_{name}"] = oracle.fused_push_all(x[None, :], slicer=code).view(np.uint8)
    # noisy block: two channels at 10 dB, one burst each
    rng = np.random.default_rng(2024)
    sps, n = 10, 3600 * 2 * 10 + 8000
    chans = []
    for c in range(2):
        off, bursts = int(rng.integers(500, 3000)), []
        for _ in range(1):
            _, _, _, _, words = synth.random_message(rng)
            bursts.append((off, synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)))
            off += 3600 * 2 * sps
        chans.append(synth.fsk_modulate(n, bursts, sps=sps, fs=200e3, snr_db=10.0, rng=rng))
    q = np.stack([q8(ch, 48.0) for ch in chans])
    out["noisy_i8"] = q
    xn = (q.astype(np.float32) / 48.0).view(np.complex64).reshape(2, -1)
    for code, name in ((0, "atan"), (1, "product"), (2, "sine")):
        out[f"noisy_records_{name}"] = oracle.fused_push_all(xn, sps=sps, slicer=code).view(np.uint8)
        sha = []
        for c in range(2):
            f = oracle.Fused(c, sps, 0, False, code)
            f.push(xn[c])
            sha.append(hashlib.sha256(f.taps()[2].tobytes()).hexdigest())
        out[f"noisy_bits_sha_{name}"] = np.array(sha)
    np.savez_compressed(os.path.join(HERE, "recc_golden_r02.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})
    print({k: len(v) // 728 for k, v in out.items() if "records" in k})


if __name__ == "__main__":
    main()
