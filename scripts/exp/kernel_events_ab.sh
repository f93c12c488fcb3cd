#!/bin/bash
# EXPERIMENT (round 6): the dominant kernel's span from events bound to its own dispatch (hipExtLaunchKernelGGL) against a pair of
# hipEventRecord around the launch (AMPS_RECC_TIMING_EVENTS=markers), in the driver's 20-step form and in the sustained form
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/kernel_events_ab.txt; : > $OUT
timeout 600 python -m pytest tests/test_gpu_timing_modes.py tests/test_gpu_errors.py -x -q 2>&1 | tail -2 >> $OUT
C="--no-cpu-baseline --no-other-specs --no-latency --no-other-decim"
line() { python - "$1" "$2" <<'PY' >> gpurun_out/kernel_events_ab.txt
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")][-1]
    r = d["roofline"]; s = d.get("secondary")
    print("%-22s value %9.1f  ms/step %.4f  kernel_ms %.4f  frac %.4f  e2e %.4f  timed %d  %s | direct832 %s" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["frac_end_to_end"], r["launches_timed"], r["events_vs_step"]["consistent"],
          ("%.1f ms/step %.4f kernel %.4f frac %.4f" % (s["value"], s["ms_per_step"], s["roofline"]["kernel_ms"], s["roofline"]["frac"])) if s else "-"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2 3; do
  AMPS_RECC_TIMING_EVENTS=markers python bench.py $C --steps 20 --warmup 5 > /tmp/l.json 2>/dev/null; line "markers steps20" /tmp/l.json
  python bench.py $C --steps 20 --warmup 5 > /tmp/l.json 2>/dev/null; line "kernel-bound steps20" /tmp/l.json
done
AMPS_RECC_TIMING_EVENTS=markers python bench.py $C > /tmp/l.json 2>/dev/null; line "markers sustained" /tmp/l.json
python bench.py $C > /tmp/l.json 2>/dev/null; line "kernel-bound sustained" /tmp/l.json
AMPS_BENCH_TIMING=dominant python bench.py $C --secondary none > /tmp/l.json 2>/dev/null; line "kernel-bound every push" /tmp/l.json
cat $OUT
