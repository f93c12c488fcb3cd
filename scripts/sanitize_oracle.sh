#!/bin/bash
# The CPU checker under AddressSanitizer + UndefinedBehaviorSanitizer: a copy of oracle/ is built with -fsanitize=address,undefined in a
# scratch directory and the oracle's own CPU tests plus 60 random fused-seam cases (tests/fuzzlib.py, three push block sizes each) run
# against it.  Nothing in the tree is touched.  Round 4: 78 tests + 180 runs, no report.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/amps_san}
rm -rf "$W"; mkdir -p "$W/t"
cp -r "$ROOT/oracle" "$W/oracle"; rm -f "$W/oracle/libamps_oracle.so"
cp "$ROOT"/tests/*.py "$W/t/"; cp -r "$ROOT/tests/golden" "$W/t/"
ln -s "$ROOT/gr_amps_amd" "$W/gr_amps_amd"; ln -s "$ROOT/include" "$W/include"
(cd "$W/oracle" && gcc -O1 -g -std=gnu11 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -mfma -fno-fast-math \
    -I"$ROOT/include" -I. ref_chain.c fused_model.c -lm -o libamps_oracle.so)
cat > "$W/fz.py" <<PY
import sys
sys.path.insert(0, "$W"); sys.path.insert(0, "$W/t")
import oracle, fuzzlib
assert oracle.__file__.startswith("$W")
n = 0
for case in range(60):
    rng, info, iq = fuzzlib.build_case(case, 77)
    for blk in (None, 777, 4096):
        n += len(oracle.fused_push_all(iq, sps=info["sps"], tolerance=info["tol"], majority=info["majority"], slicer=info["slicer"],
                                       tracking=not info["fixed"], block=blk))
print("fused-seam cases: 180 runs,", n, "records")
PY
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
cd "$W"
python -m pytest t/test_cpu_oracle.py t/test_cpu_oracle_pins.py t/test_cpu_exact_slicer.py t/test_second_restatement.py t/test_recc_work_restatement.py \
    -q -p no:cacheprovider 2>&1 | tee "$W/pytest.log" | tail -3
python fz.py 2>&1 | tee "$W/fz.log" | tail -2
if grep -q "runtime error\|AddressSanitizer" "$W/pytest.log" "$W/fz.log"; then echo "SANITIZER REPORTS:"; grep -n "runtime error\|AddressSanitizer" "$W/pytest.log" "$W/fz.log" | head; exit 1; fi
echo "no sanitizer report"
