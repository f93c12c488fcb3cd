"""Randomised differential cases for the fused IQ seam: GPU records vs the CPU model (oracle/fused_model.c), byte for byte.
Random samples-per-symbol, channel count, SNR (8..30 dB), burst placement (incl. truncated bursts, damaged preambles and
triggers inside a hold-off window), push schedule (sync drains, split drains, several pushes per drain), sync tolerance,
slicer spec (A / B / C / D of include/amps_recc_numerics.h), symbol-clock and carrier offsets of the mobile, tracked or fixed capture timing
(round 4), host- or device-resident blocks.  Used by tests/test_gpu_fuzz.py and scripts/fuzz_parity.py.

History: the first campaign (150 cases) failed 48 times -- host-resident blocks pushed back to back without a drain in
between could be overwritten in the device staging buffer while the previous push's kernels were still reading it (a
staged copy from pageable memory is not ordered after earlier kernels of a non-blocking stream): fixed with StageFence.
A second campaign failed 2 of 1500 host-block cases: hipMemcpy2DAsync from pageable memory had not read its source
when it returned and the test freed the block: fixed with a synchronous staging copy (keep_host=True keeps the blocks
alive, which is how the cause was isolated)."""
import numpy as np

import oracle
from gr_amps_amd import capi, synth

FULL = 41 + 7 + 7 * 240


def build_case(case, seed0, sps=None):
    """sps: force the sample rate (2 = the wideband seam at D = 768, which the IQ seam's kernels do not offer) -- every other draw stays
    what it is for that (case, seed)"""
    rng = np.random.default_rng(seed0 * 100000 + case)
    drawn = int(rng.choice([3, 4, 5, 6, 8, 10, 12]))
    sps = drawn if sps is None else int(sps)
    C = int(rng.integers(1, 5))
    tol = int(rng.choice([0, 0, 0, 1, 2, 4, 8]))
    majority = bool(rng.integers(0, 4) == 0)
    slicer = int(rng.choice([0, 3, 3, 1, 2]))              # numeric spec of the slicer: A, D (default since round 4), B, C
    fixed = bool(rng.integers(0, 4) == 0)                   # AMPS_RECC_FLAG_FIXED_TIMING: one case in four
    ppm = float(rng.choice([0.0, 0.0, 100.0, -100.0, 400.0, -700.0, 1500.0]))
    cfo = float(rng.choice([0.0, 0.0, 1000.0, -2000.0, 3000.0]))
    specs, N = [], 0
    for _ in range(C):
        off, bursts = int(rng.integers(200, 5000)), []
        for _ in range(int(rng.integers(0, 5))):
            _, _, _, _, words = synth.random_message(rng)
            bits = synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)
            keep = FULL if rng.random() < 0.6 else int(rng.integers(45, FULL))
            bits = bits[:keep]
            for p in rng.choice(np.arange(6, 39), size=int(rng.choice([0, 0, 0, 1, 2])), replace=False):
                bits[int(p)] ^= 1
            bursts.append((off, bits))
            off += int(keep * 2 * sps + rng.integers(50, 3000) * sps)
        specs.append(bursts)
        N = max(N, off + 3000)
    N = int(N)
    snr = float(rng.uniform(8, 30))
    iq = np.stack([synth.fsk_modulate(N, b, sps=sps, fs=20e3 * sps, snr_db=snr, rng=rng, sym_ppm=ppm, cfo_hz=cfo) for b in specs])
    # one case in three runs on a handle padded with idle (all-zero) channels to 64 and more: there the resolve kernel's own
    # workgroup decodes its channel's captures; below 64 channels they go through the capture queue (recc_resolve.hip.h)
    pad = int(rng.integers(64, 80)) if rng.integers(0, 3) == 0 else 0
    return rng, dict(sps=sps, C=C, tol=tol, majority=majority, slicer=slicer, snr=round(snr, 1), N=N, pad=pad, fixed=fixed, ppm=ppm, cfo=cfo), iq


def run_case(case, seed0, resident=False, keep_host=False):
    """returns (ok, info)"""
    rng, info, iq = build_case(case, seed0)
    sps, C, tol, N = info["sps"], info["C"], info["tol"], info["N"]
    want = oracle.fused_push_all(iq, sps=sps, tolerance=tol, majority=info["majority"], slicer=info["slicer"], tracking=not info["fixed"])
    if resident:
        import torch
    Cg = C + info["pad"]
    if info["pad"]:
        iq = np.concatenate([iq, np.zeros((info["pad"], N), iq.dtype)])
    with capi.Recc(n_channels=Cg, sps=sps, max_samples=N, max_bursts=256, sync_tolerance=tol, majority=info["majority"],
                   slicer=("atan", "product", "sine", "exact")[info["slicer"]], fixed_timing=info["fixed"]) as r:
        off, recs, pipelined, open_, keep = 0, [], bool(rng.integers(0, 2)), False, []
        while off < N:
            b = int(min(N - off, rng.integers(1, max(2, N // 2))))
            blk = np.ascontiguousarray(iq[:, off:off + b])
            if resident:           # the caller owns device blocks until the results are drained
                blk = torch.from_numpy(blk).to("cuda:0")
                torch.cuda.synchronize()
                keep.append(blk)
            if keep_host:
                keep.append(blk)
            r.push_iq(blk)
            off += b
            if pipelined:
                if open_:
                    recs.append(r.drain_end())
                r.drain_begin()
                open_ = True
            elif rng.integers(0, 2):
                recs.append(r.drain())
        recs.append(r.drain_end() if open_ else r.drain())
        recs.append(r.drain())
    got = np.concatenate(recs)
    got = got[np.lexsort((got["position"], got["channel"]))]
    info.update(got=len(got), want=len(want), pipelined=pipelined)
    return got.tobytes() == want.tobytes(), info
