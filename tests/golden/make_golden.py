"""Regenerates tests/golden/recc_golden.npz -- regression fixtures produced by the CPU oracle.

The reference cannot be built in this image (DESIGN.md "Oracle"), so these vectors pin the ORACLE's
behaviour over time and let the GPU box check the HIP path against committed data; the only
reference-derived vectors are in survey_kats.json.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from gr_amps_amd import synth  # noqa: E402


def stream(seed, n, offs):
    rng = np.random.default_rng(seed)
    b = []
    for o in offs:
        _, _, _, _, w = synth.random_message(rng)
        b.append((o, synth.burst_bits(w, dcc=int(rng.integers(0, 4)), rng=rng)))
    return synth.symbol_stream(n, b, rng)


def main():
    out = {}
    # (a) symbol streams + chunk schedules -> (publishing call index, payload sha256)
    cases = {"q2": (stream(1, 40000, [4743, 4743 + 9456, 4743 + 2 * 9456]), [1000, 4096, 333, 8191]),
             "q4": (stream(2, 80000, [63000]), [4096]),
             "q4ok": (stream(2, 80000, [60000]), [4096])}
    for name, (s, chunks) in cases.items():
        out[f"sym_{name}"] = np.packbits(s)
        out[f"sym_{name}_len"] = np.array([s.size])
        for ch in chunks:
            res = oracle.Recc().run(s, ch)
            out[f"sym_{name}_chunk{ch}_calls"] = np.array([c for c, _ in res], np.int64)
            out[f"sym_{name}_chunk{ch}_sha"] = np.array([hashlib.sha256(b.tobytes()).hexdigest() for _, b in res])
    # (b) bursts -> records
    rng = np.random.default_rng(5)
    bursts = []
    for i in range(12):
        _, _, _, _, words = synth.random_message(rng)
        syms = synth.manchester(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng))[82:82 + 3374].copy()
        idx = rng.choice(syms.size, int(rng.integers(0, 80)), replace=False)
        syms[idx] ^= 1
        bursts.append(syms)
    bursts.append(rng.integers(0, 2, 3374).astype(np.uint8))
    bursts = np.stack(bursts)
    out["bursts"] = np.packbits(bursts, axis=1)
    out["burst_records"] = oracle.decode_bursts(bursts, np.arange(len(bursts))).view(np.uint8)
    # (c) IQ (int16, x/8192) -> fused records
    iq, truth = synth.make_channel_block(48000, 1, seed=77)
    q = np.clip(np.round(iq.view(np.float32) * 8192.0), -32768, 32767).astype(np.int16)
    out["iq_i16"] = q
    x = (q.astype(np.float32) / 8192.0).view(np.complex64)
    out["iq_records"] = oracle.fused_push_all(x[None, :]).view(np.uint8)
    out["iq_truth_min"] = np.array([t[2] for t in truth])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "recc_golden.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()
