#!/bin/bash
# EXPERIMENT (round 6): the streaming kernel's wave-stream geometry against the HBM channel interleave -- 4096 waves x 104 tiles start 416 KB
# apart (13 x 32 KB); fewer waves give spans that are no such multiple
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/front_span_ab.txt; : > $OUT
for rep in 1 2; do
for w in 4096 4092 4064 4032 4000 3968 3904 3840 3584; do
  echo -n "waves $w: " >> $OUT
  AMPS_RECC_MAX_WAVES=$w timeout 200 python scripts/bench_front.py 40 2>&1 | grep -E "exact|product" | cut -c1-44 | tr '\n' ' ' >> $OUT; echo >> $OUT
done
done
cat $OUT
