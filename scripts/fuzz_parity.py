"""Randomised differential campaign of the fused IQ seam (tests/fuzzlib.py) at length.
usage (GPU box): python scripts/fuzz_parity.py [n_cases] [seed] [device]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzlib

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
resident = len(sys.argv) > 3 and sys.argv[3] == "device"
bad, t0 = 0, time.time()
for case in range(ncases):
    ok, info = fuzzlib.run_case(case, seed0, resident)
    if not ok:
        bad += 1
        print("MISMATCH case", case, info, flush=True)
print("%d cases, %d mismatches, %.1f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
