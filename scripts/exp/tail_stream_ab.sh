#!/bin/bash
# EXPERIMENT (round 6): resolve / capture / decode of step i on a second stream beside the filter bank of step i + 1
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/tail_stream_ab.txt
: > $OUT
COMMON="--no-cpu-baseline --no-other-decim --no-other-specs --secondary none --steps 3000 --no-power-sample --no-latency"
run() {  # label, env..., -- args
    label=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" timeout 300 python bench.py $COMMON "$@" 2>>gpurun_out/tail_stream_ab.err | grep '^{' | tail -1)
    python - "$label" <<PY >> $OUT
import json, sys
d = json.loads('''$line''') if '''$line'''.strip() else None
if d is None:
    print(sys.argv[1], "FAILED")
else:
    r = d["roofline"]
    print("%-34s value %9.1f  ms/step %.4f  kernel_ms %.4f  frac %.4f  e2e %.4f  samples %d" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_end_to_end"], d["config"]["wideband_samples_per_step"]))
PY
}
for rep in 1 2; do
run "base 256"            A=1 --
run "tail 256"            AMPS_RECC_TAIL_STREAM=1 --
run "tail 248"            AMPS_RECC_TAIL_STREAM=1 AMPS_RECC_CHZ_WGS=248 -- --samples $((248*11*64*768))
run "tail 244"            AMPS_RECC_TAIL_STREAM=1 AMPS_RECC_CHZ_WGS=244 -- --samples $((244*11*64*768))
run "tail 240"            AMPS_RECC_TAIL_STREAM=1 AMPS_RECC_CHZ_WGS=240 -- --samples $((240*11*64*768))
run "tail 232"            AMPS_RECC_TAIL_STREAM=1 AMPS_RECC_CHZ_WGS=232 -- --samples $((232*11*64*768))
run "notail 240"          AMPS_RECC_CHZ_WGS=240 -- --samples $((240*11*64*768))
done
cat $OUT
