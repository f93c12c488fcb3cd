"""Regenerates tests/golden/recc_golden_r04.npz -- round-4 regression fixtures produced by the CPU oracle (same status as
make_golden.py / make_golden_r02.py: they pin the ORACLE over time and travel to the GPU box; they are not reference-derived).

Contents: the records of recc_golden.npz's IQ block under slicer spec D; the records of recc_golden_r02.npz's 10 dB block under
all four slicer specs WITH the capture's timing tracking (round 4's default; the round-2 file holds the fixed-timing ones), and
the spec-D slicer bit streams (sha256); two impaired bursts at 30 dB (int8 samples x/48) -- symbol clock +800 ppm, carrier
+1.5 kHz, at 10 and at 3 samples per symbol -- with their records, which only the tracking capture produces.
Run from the repo root:  python tests/golden/make_golden_r04.py"""
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
    r02 = np.load(os.path.join(HERE, "recc_golden_r02.npz"))
    x = (base["iq_i16"].astype(np.float32) / 8192.0).view(np.complex64)
    out["iq_records_exact"] = oracle.fused_push_all(x[None, :], slicer=3).view(np.uint8)
    xn = (r02["noisy_i8"].astype(np.float32) / 48.0).view(np.complex64).reshape(2, -1)
    for code, name in ((0, "atan"), (1, "product"), (2, "sine"), (3, "exact")):
        out[f"noisy_tracked_records_{name}"] = oracle.fused_push_all(xn, sps=10, slicer=code).view(np.uint8)
    sha = []
    for c in range(2):
        f = oracle.Fused(c, 10, slicer=3)
        f.push(xn[c])
        sha.append(hashlib.sha256(f.taps()[2].tobytes()).hexdigest())
    out["noisy_bits_sha_exact"] = np.array(sha)
    # impaired bursts: one row per samples-per-symbol, zero padded to a common length
    rows, lens, mins = [], [], []
    for sps in (10, 3):
        n = 3456 * sps + 6000
        xi, truth = synth.make_channel_block(n, 1, seed=404 + sps, sps=sps, snr_db=30.0, first=1500, sym_ppm=800.0, cfo_hz=1500.0)
        rows.append(q8(xi, 48.0))
        lens.append(n)
        mins.append(truth[0][2])
    width = max(len(r) for r in rows)
    out["impaired_i8"] = np.stack([np.pad(r, (0, width - len(r))) for r in rows])
    out["impaired_len"] = np.array(lens)
    out["impaired_min"] = np.array(mins)
    xi = (out["impaired_i8"].astype(np.float32) / 48.0).view(np.complex64).reshape(2, -1)
    for row, sps in enumerate((10, 3)):
        rec = oracle.fused_push_all(xi[row:row + 1, :lens[row]], sps=sps)
        assert len(rec) == 1 and rec[0]["valid"].all() and rec[0]["min"].decode() == mins[row], (sps, len(rec))
        out[f"impaired_records_sps{sps}"] = rec.view(np.uint8)
    np.savez_compressed(os.path.join(HERE, "recc_golden_r04.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()
