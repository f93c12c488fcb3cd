#!/bin/bash
# EXPERIMENT (round 6): resolve + capture + decode of an IQ-seam push on the tail stream beside the next push's streaming kernel (few-channel
# handles).  Parity first (every push of the few-channel tests forced through it), then direct1 with and without, alternating, one call.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/tail_iq_ab.txt
: > $OUT
echo "== parity, AMPS_RECC_TAIL_STREAM=1 AMPS_RECC_TAIL_MIN_WORK=0" >> $OUT
AMPS_RECC_TAIL_STREAM=1 AMPS_RECC_TAIL_MIN_WORK=0 timeout 900 python -m pytest tests -m gpu -x -q -k "not wideband and not channelizer and not rccl and not ranks and not fullsize" 2>&1 | tail -5 >> $OUT
COMMON="--workload direct1 --no-cpu-baseline --no-other-specs --secondary none --steps 2000 --no-power-sample --no-latency"
run() {
    label=$1; shift
    line=$(env "$@" timeout 300 python bench.py $COMMON 2>>gpurun_out/tail_iq_ab.err | grep '^{' | tail -1)
    python - "$label" <<PY >> $OUT
import json, sys
d = json.loads('''$line''') if '''$line'''.strip() else None
if d is None:
    print(sys.argv[1], "FAILED")
else:
    r = d["roofline"]
    print("%-12s value %9.1f  ms/step %.4f  kernel_ms %.4f  frac %.4f  e2e %.4f  checked %s" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_end_to_end"], d["config"].get("checked")))
PY
}
for rep in 1 2 3; do
run "one stream" AMPS_RECC_TAIL_STREAM=0
run "tail stream" AMPS_RECC_TAIL_STREAM=1
done
cat $OUT
