#!/usr/bin/env python3
"""Step 1 of the pin recipe (scripts/pin_with_reference.sh): writes tests/golden/pin_inputs.npz -- the seeded inputs on which the
REFERENCE (unsynchronized/gr-amps built against GNU Radio 3.7 + IT++, elsewhere) and the oracle (oracle/ref_chain.c, here) are to be
compared.  Plain arrays only (no pickles), so that the Python 2 interpreter GNU Radio 3.7 comes with can read the file.

What the file holds, and which rows of SURVEY.md 8(a) each entry pins once scripts/pin/run_reference.py has produced
tests/golden/reference_pins.npz from it:
  sym_*      u8 symbol streams (0/1) for amps.recc -> R1 (trigger), R2 (recc_impl::work, lib/recc_impl.cc:93-145).  Bursts are spaced so
             that the scheduler's chunking cannot change what is published, plus one stream per quirk Q1 / Q3 /
             Q4 whose outcome is chunk-independent by construction.
  bursts     3374-byte bursts for amps.recc_decode -> R3..R8 (lib/recc_decode_impl.cc:53-169, lib/utils.cc:27-59, lib/amps_packet.h)
             and the reply generation (:181-272) as seen on its output ports; clean ones of every message class, ones with bit errors
             (first-valid-of-five), non-Manchester pairs, weight-3 error patterns that IT++'s "#roots == deg" rule accepts.
  bch48      48-bit words for a direct itpp::BCH(63,2,true) decode -> R4 (only if the runner's helper is built; see run_reference.py)
  iq400      fc32 at 400 ksps, the channel at +160 kHz (grc/recctest.grc:591) -> G1 freq_xlating_fir_filter_ccc + firdes.low_pass
  iq200      fc32 at 200 ksps -> G2 quadrature_demod_cf (fast_atan2f), G3 clock_recovery_mm_ff, G4 binary_slicer_fb
Run from the repo root:  python scripts/pin/make_pin_inputs.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
    # ---- R2: bursts 20 000 symbols apart (>> 3374 + 74 + any scheduler chunk), then the quirks
    out["sym_spaced"] = stream(101, 130000, [5000 + 20000 * i for i in range(6)])
    out["sym_q1_short_tail"] = stream(102, 4743 + 82 + 3374, [4743])       # exactly 3374 symbols follow the trigger: NOT published (strict >)
    out["sym_q1_one_more"] = stream(102, 4743 + 82 + 3375, [4743])         # one more: published
    out["sym_q4_lost"] = stream(2, 80000, [63000])                          # first-fill wrap forgets the pending trigger
    out["sym_q4_kept"] = stream(2, 80000, [60000])
    # ---- R3..R8 + replies
    rng = np.random.default_rng(7)
    bursts = []
    for kind, esn, dialed in (("page_response", 0, ""), ("registration", 0x82345678, ""), ("origination", 0x1234abcd, "5551212"),
                              ("origination", 0xdeadbeef, "18005551212*#"), ("origination", 1, "0"), ("registration", 0, "")) * 3:
        words = synth.make_message(kind, "".join(str(int(d)) for d in rng.integers(0, 10, 10)), esn, dialed)
        syms = synth.manchester(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng))[82:82 + 3374].copy()
        bursts.append(syms)
    for i in range(24):                                                      # random messages, 0..60 symbol errors
        _, _, _, _, words = synth.random_message(rng)
        syms = synth.manchester(synth.burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng))[82:82 + 3374].copy()
        idx = rng.choice(syms.size, int(rng.integers(0, 60)), replace=False)
        syms[idx] ^= 1
        bursts.append(syms)
    bursts.append(rng.integers(0, 2, 3374).astype(np.uint8))                 # noise
    out["bursts"] = np.stack(bursts).astype(np.uint8)
    # ---- R4 directly: valid code words with 0..4 errors, incl. the weight-3 class S1 = 0 (IT++ "corrects" those whose S3 is a cube)
    words48 = []
    for i in range(400):
        cw = np.array(synth.bch_encode(rng.integers(0, 2, 36)), np.uint8)
        e = rng.choice(48, int(rng.integers(0, 5)), replace=False)
        cw[e] ^= 1
        words48.append(cw)
    out["bch48"] = np.stack(words48)
    # ---- G1..G4
    n4 = 1 << 17
    iq400, _ = synth.make_channel_block(n4, 1, seed=77, sps=20)
    out["iq400"] = (iq400 * np.exp(2j * np.pi * 0.4 * np.arange(n4))).astype(np.complex64)     # the channel at +160 kHz of 400 ksps
    iq200, _ = synth.make_channel_block(1 << 16, 1, seed=78, sps=10)
    out["iq200"] = iq200.astype(np.complex64)
    path = os.path.join(ROOT, "tests", "golden", "pin_inputs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
