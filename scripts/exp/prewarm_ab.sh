#!/bin/bash
# EXPERIMENT (round 6): how long the untimed prewarm of the driver's 20-step form has to be for the timed region to sit in the settled power state
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prewarm_ab.txt; : > $OUT
C="--no-cpu-baseline --no-other-specs --no-latency --no-other-decim --secondary none --steps 20 --warmup 5"
for rep in 1 2 3 4; do
for pw in 400 1000 2000 4000; do
  python bench.py $C --prewarm-ms $pw > /tmp/l.json 2>/dev/null
  python - $pw <<'PY' >> gpurun_out/prewarm_ab.txt
import json, sys
d = [json.loads(l) for l in open("/tmp/l.json") if l.startswith("{")][-1]
r = d["roofline"]; p = d.get("power") or {}
print("prewarm %5s ms  value %9.1f  ms/step %.4f  kernel_ms %.4f  frac %.4f  W %s MHz %s" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], p.get("package_w_mean"), p.get("sclk_mhz_mean")))
PY
done
done
sort -k2,2n -s $OUT
