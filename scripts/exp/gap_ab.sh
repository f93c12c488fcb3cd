#!/bin/bash
# EXPERIMENT: what the per-step HIP event records cost the step (timed region brackets the dominant kernel of every push)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/gap_ab.txt
: > $OUT
COMMON="--no-cpu-baseline --no-other-decim --no-other-specs --secondary none --steps 4000 --no-power-sample --no-latency"
for rep in 1 2 3; do
for mode in dominant off; do
  line=$(AMPS_BENCH_TIMING=$mode timeout 300 python bench.py $COMMON "$@" 2>>gpurun_out/gap_ab.err | grep '^{' | tail -1)
  echo "$mode $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  kernel_ms %.4f value %.1f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"]))')" >> $OUT
done
done
cat $OUT
