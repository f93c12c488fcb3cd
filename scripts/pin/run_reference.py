#!/usr/bin/env python
"""Step 2 of the pin recipe (scripts/pin_with_reference.sh) -- runs where GNU Radio 3.7 + IT++ + the BUILT reference
(unsynchronized/gr-amps, `import amps`) are installed; NOT in the build image of this repository, where none of them exists.
Python 2 and 3 compatible on purpose (GNU Radio 3.7 is a Python 2 framework); needs numpy only.

Reads  tests/golden/pin_inputs.npz   (scripts/pin/make_pin_inputs.py)
Writes tests/golden/reference_pins.npz: what the REFERENCE's own blocks produce on those inputs --
  R2   amps.recc                      : the 3374-byte blobs published on "bursts" per symbol stream          (lib/recc_impl.cc:93-145)
  R3-8 amps.recc_decode               : every message on focc_words / fvc_words / fvc_mute / audio_mute / command_out per burst, as
                                        the text lines gr_amps_amd/recctest prints                          (lib/recc_decode_impl.cc:81-272)
  G1   filter.freq_xlating_fir_filter_ccc(2, firdes.low_pass(3, 400e3, 10e3, 4.5e3), 160e3, 400e3) + its taps   (grc/recctest.grc:889-937, 115-155)
  G2   analog.quadrature_demod_cf(1)                                                                          (:458)
  G3   digital.clock_recovery_mm_ff(10, .25*.175*.175*3, 0, .05, .005)  [as grc/recctest.grc:846-874 sets it]
  G4   digital.binary_slicer_fb                                                                               (:807)
tests/test_cpu_reference_pins.py then holds oracle/ref_chain.c to that file (and is skipped while it does not exist).
Nothing here is imported by the product, the tests or bench.py."""
from __future__ import print_function

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from gnuradio import analog, blocks, digital, filter, gr   # noqa: A004
    from gnuradio.filter import firdes
    import pmt
    import amps

    inp = np.load(os.path.join(ROOT, "tests", "golden", "pin_inputs.npz"))
    out = {"gnuradio_version": np.array(gr.version()), "generated": np.array(time.strftime("%Y-%m-%d %H:%M:%S"))}

    class msg_sink(gr.basic_block):
        """collects (port, message) pairs"""

        def __init__(self, ports):
            gr.basic_block.__init__(self, name="pin_msg_sink", in_sig=None, out_sig=None)
            self.got = []
            for p in ports:
                self.message_port_register_in(pmt.intern(p))
                self.set_msg_handler(pmt.intern(p), self._handler(p))

        def _handler(self, port):
            def h(m):
                self.got.append((port, m))
            return h

    def blob_bytes(m):
        return np.array(pmt.u8vector_elements(m), np.uint8) if pmt.is_u8vector(m) else np.frombuffer(bytearray(pmt.blob_data(m)), np.uint8)

    def bits(m):
        return "".join(str(int(b)) for b in blob_bytes(m))

    # ---- R2: symbol streams through amps.recc
    for key in [k for k in inp.files if k.startswith("sym_")]:
        syms = inp[key].astype(np.uint8)
        tb = gr.top_block()
        src = blocks.vector_source_b(syms.tolist(), False)
        recc = amps.recc()
        sink = msg_sink(["bursts"])
        tb.connect(src, recc)
        tb.msg_connect((recc, "bursts"), (sink, "bursts"))
        tb.run()
        time.sleep(0.2)
        pubs = [blob_bytes(m) for _, m in sink.got]
        out["ref_" + key + "_count"] = np.array([len(pubs)])
        out["ref_" + key + "_bursts"] = np.stack(pubs) if pubs else np.zeros((0, 3374), np.uint8)

    # ---- R3..R8 + replies: bursts through amps.recc_decode, one at a time, the output ports as text lines
    ports = ["focc_words", "fvc_words", "audio_mute", "fvc_mute", "command_out"]
    lines_all, owner = [], []
    for i, b in enumerate(inp["bursts"]):
        tb = gr.top_block()
        dec = amps.recc_decode()
        sink = msg_sink(ports)
        for p in ports:
            tb.msg_connect((dec, p), (sink, p))
        tb.start()
        dec.to_basic_block()._post(pmt.intern("bursts"), pmt.init_u8vector(len(b), [int(v) for v in b]))
        time.sleep(0.3)
        tb.stop()
        tb.wait()
        for port, m in sink.got:
            if port == "focc_words":
                l = "MSG focc_words stream=%d n=%d w1=%s w2=%s" % (pmt.to_long(pmt.tuple_ref(m, 0)), pmt.to_long(pmt.tuple_ref(m, 1)), bits(pmt.tuple_ref(m, 2)), bits(pmt.tuple_ref(m, 3)))
            elif port == "fvc_words":
                l = "MSG fvc_words n=%d w1=%s repeat=%d" % (pmt.to_long(pmt.tuple_ref(m, 0)), bits(pmt.tuple_ref(m, 1)), pmt.to_uint64(pmt.tuple_ref(m, 2)))
            elif port == "command_out":
                l = "MSG command_out " + "".join(chr(c) for c in pmt.u8vector_elements(pmt.cdr(m)))
            else:
                l = "MSG %s %d" % (port, 1 if pmt.to_bool(m) else 0)
            lines_all.append(l)
            owner.append(i)
    out["ref_burst_lines"] = np.array(lines_all if lines_all else [""])
    out["ref_burst_line_owner"] = np.array(owner, np.int64)

    # ---- G1: the channel filter and its taps
    taps = firdes.low_pass(3.0, 400e3, 10e3, 4.5e3, firdes.WIN_BLACKMAN, 6.76)
    out["ref_g1_taps"] = np.array(taps, np.float32)

    def run_stream(src_block, chain, sink_block):
        tb = gr.top_block()
        tb.connect(*([src_block] + chain + [sink_block]))
        tb.run()
        return np.array(sink_block.data())

    x4 = inp["iq400"].astype(np.complex64)
    y = run_stream(blocks.vector_source_c(x4.tolist(), False), [filter.freq_xlating_fir_filter_ccc(2, taps, 160e3, 400e3)], blocks.vector_sink_c())
    out["ref_g1_out"] = y.astype(np.complex64)
    # ---- G2, G3, G4 on the 200 ksps block, stage by stage and chained
    x2 = inp["iq200"].astype(np.complex64)
    d = run_stream(blocks.vector_source_c(x2.tolist(), False), [analog.quadrature_demod_cf(1.0)], blocks.vector_sink_f())
    out["ref_g2_out"] = d.astype(np.float32)
    mm = run_stream(blocks.vector_source_f(d.tolist(), False), [digital.clock_recovery_mm_ff(10.0, 0.25 * 0.175 * 0.175 * 3, 0.0, 0.05, 0.005)], blocks.vector_sink_f())
    out["ref_g3_out"] = mm.astype(np.float32)
    sl = run_stream(blocks.vector_source_f(mm.tolist(), False), [digital.binary_slicer_fb()], blocks.vector_sink_b())
    out["ref_g4_out"] = sl.astype(np.uint8)

    path = os.path.join(ROOT, "tests", "golden", "reference_pins.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    for k in sorted(out):
        print("  %-28s %s" % (k, getattr(out[k], "shape", "")))


if __name__ == "__main__":
    sys.exit(main())
