"""tests/golden/pin_manifest.json: what a GOOD tests/golden/reference_pins.npz looks like -- every array scripts/pin/run_reference.py writes, with
its dtype kind and the shape the committed inputs (tests/golden/pin_inputs.npz) imply, and the GNU Radio version the reference is built
against (CMakeLists.txt:96: >= 3.7.2; the blocks' arithmetic this repository restates is 3.7's).  The shapes are those of the ORACLE's
side of the recipe: where the reference's output may legitimately differ in length (filter history, the clock-recovery loop's
last symbols) the manifest gives a tolerance.  tests/test_cpu_reference_pins.py holds the oracle's side to this file on every CPU run
(the recipe cannot rot) and checks a reference file against it before comparing values (a bad run is told from a real difference).
usage: python scripts/pin/make_manifest.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_cpu_reference_pins as T
    inp = np.load(T.INPUTS)
    orc = T.oracle_side(inp)
    arrays = {}
    for key in [k for k in inp.files if k.startswith("sym_")]:
        b = orc["orc_" + key + "_bursts"]
        arrays["ref_" + key + "_bursts"] = {"kind": "u", "shape": list(b.shape), "exact_shape": True, "what": "blobs amps.recc publishes on port bursts, in order"}
        arrays["ref_" + key + "_count"] = {"kind": "iu", "shape": [1], "exact_shape": True, "what": "number of those blobs"}
    nl = sum(len(l) for l in orc["orc_burst_lines"])
    arrays["ref_burst_lines"] = {"kind": "U", "shape": [nl], "exact_shape": False, "tolerance": 0, "what": "one line per message amps.recc_decode publishes, all bursts"}
    arrays["ref_burst_line_owner"] = {"kind": "i", "shape": [nl], "exact_shape": False, "tolerance": 0, "what": "index of the burst each line belongs to"}
    arrays["ref_g1_taps"] = {"kind": "f", "shape": [int(orc["orc_g1_taps"].size)], "exact_shape": True, "what": "firdes.low_pass(3, 400e3, 10e3, 4.5e3, WIN_BLACKMAN)"}
    arrays["ref_g1_out"] = {"kind": "c", "shape": [int(orc["orc_g1_out"].size)], "exact_shape": False, "tolerance": 4, "what": "freq_xlating_fir_filter_ccc output"}
    arrays["ref_g2_out"] = {"kind": "f", "shape": [int(orc["orc_g2_out"].size)], "exact_shape": False, "tolerance": 2, "what": "quadrature_demod_cf output"}
    arrays["ref_g3_out"] = {"kind": "f", "shape": [int(orc["orc_g4_out"].size)], "exact_shape": False, "tolerance": 2, "what": "clock_recovery_mm_ff output (soft symbols)"}
    arrays["ref_g4_out"] = {"kind": "u", "shape": [int(orc["orc_g4_out"].size)], "exact_shape": False, "tolerance": 2, "what": "binary_slicer_fb output: the symbols amps.recc is fed"}
    man = {"gnuradio_version_prefix": "3.7", "gnuradio_version_min": "3.7.2",
           "inputs": {k: {"dtype": str(inp[k].dtype), "shape": list(inp[k].shape)} for k in sorted(inp.files)},
           "arrays": arrays,
           "note": "kind = numpy dtype.kind(s) accepted; exact_shape false: the last axis may differ from `shape` by at most `tolerance`"}
    path = os.path.join(ROOT, "tests", "golden", "pin_manifest.json")
    with open(path, "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print("wrote", path, len(arrays), "arrays")


if __name__ == "__main__":
    main()
