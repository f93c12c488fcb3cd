#!/bin/bash
# EXPERIMENT helper: bench.py (wideband headline only) with each scripts/variants/<name>.so in turn, alternating, one gpurun call
# usage: scripts/exp/ab_bench_libs.sh <rounds> <steps> name1 name2 ...
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
N=${1:-2}; STEPS=${2:-3000}; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    line=$(AMPS_RECC_LIB=$PWD/scripts/variants/$v.so timeout 300 python bench.py --no-cpu-baseline --no-other-decim --no-other-specs --secondary none --steps $STEPS --no-latency --no-power-sample 2>/dev/null | grep '^{' | tail -1)
    echo "$v: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms/step %.4f  kernel_ms %.4f  Gsym/s %.1f  resolve %.4f" % (d["ms_per_step"], r["kernel_ms"], d["value"]/1e3, r["other_kernels_ms_per_step"]["ms_resolve"]))')"
  done
done
