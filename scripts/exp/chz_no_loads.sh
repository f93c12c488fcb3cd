#!/bin/bash
# EXPERIMENT (round 6): the filter bank with its input stream switched off (every fast load out of the descriptor's range: zeros, no memory
# access; WRONG results) -- an upper bound on what better prefetching could buy, power effect included
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/chz_no_loads.txt; : > $OUT
for i in 1 2 3; do
  for v in base noloads cachedloads; do
    echo -n "$v: " >> $OUT; CHZ_DECIM=768 AMPS_RECC_LIB=$PWD/scripts/variants/$v.so timeout 300 python scripts/bench_chz.py 200 exact 400 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
